#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
OUT=gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 1500 python -m pytest tests -m gpu -q -s > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/pytest_gpu.log
grep -v "amdgpu.ids\|Gloo\|^$" $OUT/pytest_gpu.log | tail -40
timeout 600 python bench.py --steps 10 --warmup 3 > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"
python -c "import json;d=json.load(open('$OUT/bench.json'));print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['kernel_ms_per_step'], [ (s['bank_dtype'], round(s['frac'],3)) for s in d['scoring_roofline']], d['cpu_baseline']['value'])"
timeout 300 python bench.py --scoring-only --steps 50 --warmup 5 > $OUT/bench_scoring.json 2> $OUT/bench_scoring.err; echo "scoring rc=$?"; cat $OUT/bench_scoring.json
