#!/bin/bash
# round 3, run a: the new compute modes (bf16x3 split precision, f16) on hardware -- smoke, operator / ping-pong parity, bench with the
# parity + per-kernel roofline records, throughput per mode, kernel trace of the bf16x3 step
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
OUT=gpurun_out
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo "smoke rc=$?"; tail -5 $OUT/smoke.log
timeout 900 python -m pytest tests/test_kernels_parity.py tests/test_conv_pingpong.py -m gpu -x -q > $OUT/pytest_a.log 2>&1; echo "tests rc=$?"; tail -4 $OUT/pytest_a.log
timeout 600 python bench.py --steps 10 --warmup 3 > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"; tail -3 $OUT/bench.err
python - <<'PY'
import json
d=json.load(open('gpurun_out/bench.json'))
print('bench', round(d['value']), round(d['ms_per_step'],2), 'roofline', d['roofline']['kernel'], round(d['roofline']['frac'],3), 'family', round(d['roofline']['family']['frac'],3))
for k,v in d['parity']['modes'].items(): print(' ', k, {a:(round(b,7) if isinstance(b,float) else b) for a,b in v.items()})
print(' oracle', d['parity']['oracle_spot_check'])
for c in d['roofline']['classes'][:14]: print('  ', c)
PY
for m in bf16x3 f16; do
  timeout 300 python bench.py --steps 10 --warmup 3 --skip-extras --dtype $m > $OUT/b_$m.json 2>/dev/null
  python -c "import json;d=json.load(open('$OUT/b_$m.json'));print('bench $m', round(d['value']), round(d['ms_per_step'],3))"
done
NOPE_CONV_PP=7 timeout 300 python bench.py --steps 10 --warmup 3 --skip-extras --dtype bf16x3 2>/dev/null | python -c "import json,sys;d=json.loads(sys.stdin.read());print('bench bf16x3 posmajor-pp', round(d['value']), round(d['ms_per_step'],3))"
export TMPDIR=/tmp
( cd /tmp && timeout 300 rocprofv3 --kernel-trace -d $OLDPWD/$OUT/prof_x3 -o p -- python $OLDPWD/bench.py --steps 5 --warmup 2 --skip-extras --dtype bf16x3 > /dev/null 2>&1 )
python tools/rocpd_stats.py $(find $OUT/prof_x3 -name "*.db" | head -1) > $OUT/bench_bf16x3_kernel_stats.csv 2>&1; head -14 $OUT/bench_bf16x3_kernel_stats.csv
rm -rf $OUT/prof_x3
