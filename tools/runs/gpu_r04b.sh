#!/bin/bash
# round 4, run b: first GPU contact of the small-tile kernel, the geodesic kernel, graph opt-in; then the launch-policy sweep on
# reference-sized banks, the per-launch table at 64 hypotheses and the one-image encoder.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
OUT=$PWD/gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_conv_small.py tests/test_gpu_configs.py -m gpu -x -q -k "small or geodesic or graph or two_stream" > $OUT/pytest_r04b.log 2>&1; echo "tests rc=$?"; tail -8 $OUT/pytest_r04b.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo "smoke rc=$?"; tail -2 $OUT/smoke.log
timeout 900 python tools/small_bank_sweep.py --dtype f16 > $OUT/small_bank_sweep.txt 2>$OUT/sweep.err; echo "sweep rc=$?"; cat $OUT/small_bank_sweep.txt; tail -3 $OUT/sweep.err
timeout 300 python bench.py --dtype f16 --templates 64 --steps 10 --warmup 3 --extras roofline > $OUT/bench_n64_classes.json 2>$OUT/bench_n64.err
python - <<'PY'
import json
d=json.load(open('gpurun_out/bench_n64_classes.json'))
print('n64', round(d['ms_per_step'],3),'ms')
tot=0
for c in d['roofline']['classes']:
    tot+=c['avg_ms']*c['launches']
    print(f"{c['kernel'][:22]:>22} mode {c['mode']} taps {c['taps']} {c['Cin']:>4}->{c['Cout']:<4} @{c['H']}x{c['W']} n={c['n']:<3} x{c['launches']:<2} {c['avg_ms']*1e3:8.1f} us {c['tflops']:7.1f} TF {c['frac']:.3f}")
print('conv total ms', tot)
PY
timeout 300 python tools/encoder_bench.py > $OUT/encoder_bench.txt 2>&1; cat $OUT/encoder_bench.txt
NOPE_CONV_SMALL=0 timeout 300 python tools/encoder_bench.py > $OUT/encoder_bench_old.txt 2>&1; cat $OUT/encoder_bench_old.txt
echo done
