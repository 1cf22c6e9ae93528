#!/bin/bash
# Round-2 record run: smoke, full GPU suite, bench (with extras), scoring-only bench, rocprofv3 kernel stats, PMC traffic.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
OUT=gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
export TMPDIR=/tmp
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo "smoke rc=$?" | tee -a $OUT/smoke.log
timeout 1800 python -m pytest tests -m gpu -q -s > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/pytest_gpu.log
grep -a "passed\|failed" $OUT/pytest_gpu.log | tail -3
timeout 600 python bench.py --steps 10 --warmup 3 > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"
timeout 300 python bench.py --scoring-only --steps 50 --warmup 10 > $OUT/bench_scoring.json 2> $OUT/bench_scoring.err; echo "scoring rc=$?"
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d "$OLDPWD/$OUT/prof" -o bench -- python "$OLDPWD/bench.py" --gpus 1 --steps 4 --warmup 1 --skip-extras > "$OLDPWD/$OUT/prof.log" 2>&1 ); echo "rocprof rc=$?"
python tools/rocpd_stats.py $(find $OUT/prof -name "*.db" | head -1) > $OUT/kernel_stats.csv; rm -rf $OUT/prof
head -14 $OUT/kernel_stats.csv
bash tools/gpu_pmc.sh unet tools/unet_step.py > $OUT/pmc_run.log 2>&1; echo "pmc rc=$?"
python tools/pmc_to_traffic.py $OUT/pmc_unet.txt $OUT/pmc_traffic.json; echo "traffic rc=$?"
python -c "import json;d=json.load(open('$OUT/bench.json'));print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['kernel_ms_per_step'], [(s['bank_dtype'], round(s['frac'],3)) for s in d['scoring_roofline']], d['cpu_baseline']['value'])"
cat $OUT/bench_scoring.json | head -c 600
