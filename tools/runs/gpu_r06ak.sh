#!/bin/bash
# round 6, run ak: what the driver does at round end, on one more box of the pool: smoke(), the GPU suite with -x, and `python bench.py` with NO flags (wall time of the default run)
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out; mkdir -p $OUT
export HSA_ENABLE_IPC_MODE_LEGACY=0 TMPDIR=/tmp
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/r06ak_smoke.log 2>&1; echo "smoke rc=$?"; tail -1 $OUT/r06ak_smoke.log
timeout 2400 python -m pytest tests/ -x -q -m gpu > $OUT/r06ak_pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -2 $OUT/r06ak_pytest_gpu.log
s=$(date +%s); timeout 900 python bench.py > $OUT/r06ak_bench_default.json 2> $OUT/r06ak_bench.err; echo "bench rc=$? wall $(( $(date +%s) - s )) s"
python -c "
import json; r=json.load(open('gpurun_out/r06ak_bench_default.json')); print({k:r[k] for k in ('value','ms_per_step','steps','warmup','tolerance_met','n_gpus','scaling','dtype')}, r['roofline']['frac'], r['roofline']['traffic'], r['cpu_baseline']['value'])"
