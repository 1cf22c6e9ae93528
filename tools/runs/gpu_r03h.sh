#!/bin/bash
# round 3, run h: what the driver runs at round end, on the final tree -- smoke, the default bench line; plus the bf16x3 kernel trace
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
OUT=gpurun_out
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo "smoke rc=$?"; tail -2 $OUT/smoke.log
( time timeout 600 python bench.py --gpus 1 > $OUT/bench.json 2> $OUT/bench.err ) 2>&1 | grep real; echo "bench rc=$?"
python - <<'PY'
import json
d=json.load(open('gpurun_out/bench.json'))
r=d['roofline']
print('bench', d['dtype'], round(d['value']), round(d['ms_per_step'],2), r['kernel'], round(r['frac'],3), 'family', round(r['family']['frac'],3), 'traffic', r['traffic'])
for k,v in d['parity']['modes'].items(): print(' ', k, {a:(round(b,6) if isinstance(b,float) else b) for a,b in v.items()})
PY
export TMPDIR=/tmp
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $OLDPWD/$OUT/prof_x3 -o p -- python $OLDPWD/bench.py --steps 5 --warmup 2 --skip-extras --dtype bf16x3 > /dev/null 2>&1 )
python tools/rocpd_stats.py $(find $OUT/prof_x3 -name "*.db" | head -1) > $OUT/bench_bf16x3_kernel_stats.csv 2>&1; head -12 $OUT/bench_bf16x3_kernel_stats.csv
rm -rf $OUT/prof_x3
