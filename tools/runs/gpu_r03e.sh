#!/bin/bash
# round 3, run e: the round's evidence on the final tree -- smoke, full GPU suite, bench (default f16) + kernel trace of the same
# command, the other modes, PMC passes (U-Net step in the bench dtype, scoring kernel at 32 x 2048 bf16), scoring-only, harness
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
OUT=gpurun_out
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo "smoke rc=$?"; tail -5 $OUT/smoke.log
timeout 1500 python -m pytest tests -m gpu -x -q -s --durations=25 > $OUT/pytest_gpu.log 2>&1; echo "tests rc=$?"; tail -32 $OUT/pytest_gpu.log
timeout 600 python bench.py > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"
python - <<'PY'
import json
d=json.load(open('gpurun_out/bench.json'))
print('bench', d['dtype'], round(d['value']), round(d['ms_per_step'],2), 'roofline', d['roofline']['kernel'], round(d['roofline']['frac'],3), 'family', round(d['roofline']['family']['frac'],3), 'traffic', d['roofline']['traffic'])
for k,v in d['parity']['modes'].items(): print(' ', k, {a:(round(b,7) if isinstance(b,float) else b) for a,b in v.items()})
print(' oracle', d['parity']['oracle_spot_check'])
for s in d['scoring_roofline']: print(' scoring', s['bank_dtype'], s['N'], round(s['frac'],3))
print(' cpu', d['cpu_baseline']['value'], d['cpu_baseline']['cores'], 'speedup', round(d['speedup_vs_cpu']))
PY
export TMPDIR=/tmp
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $OLDPWD/$OUT/prof_bench -o p -- python $OLDPWD/bench.py --steps 5 --warmup 2 --skip-extras > /dev/null 2>&1 )
python tools/rocpd_stats.py $(find $OUT/prof_bench -name "*.db" | head -1) > $OUT/bench_f16_kernel_stats.csv 2>&1; head -16 $OUT/bench_f16_kernel_stats.csv
rm -rf $OUT/prof_bench
for m in bf16 bf16x3 f32; do
  timeout 300 python bench.py --steps 5 --warmup 2 --skip-extras --dtype $m > $OUT/b_$m.json 2>/dev/null
  python -c "import json;d=json.load(open('$OUT/b_$m.json'));print('bench $m', round(d['value']), round(d['ms_per_step'],3))"
done
timeout 300 python bench.py --scoring-only --steps 50 --warmup 10 > $OUT/scoring_only.json 2>/dev/null; python -c "import json;d=json.load(open('$OUT/scoring_only.json'));print('scoring-only', d['value'], d['roofline']['frac'])"
timeout 300 python -m nope_amd.harness --pose-level 2 --size 256 --dtype f16 --bank-dtype f16 > $OUT/harness_level2.txt 2>&1; tail -1 $OUT/harness_level2.txt | cut -c1-300
bash tools/gpu_pmc.sh unet tools/unet_step.py --dtype f16 > /dev/null 2>&1; echo "pmc unet done"; head -14 $OUT/pmc_unet.txt
bash tools/gpu_pmc.sh sim tools/sim_step.py > /dev/null 2>&1; echo "pmc sim done"; head -14 $OUT/pmc_sim.txt
python tools/pmc_to_traffic.py $OUT/pmc_unet.txt $OUT/pmc_traffic.json --dtype f16 --sim $OUT/pmc_sim.txt | cut -c1-1500
