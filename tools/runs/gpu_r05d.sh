#!/bin/bash
# round 5, run d: the f16x2 mode end to end -- smoke, the GPU suite with the tightened per-mode regression bounds (-s: the observed errors go to the log),
# the bench line with f16x2 as the timed mode, rocprofv3 kernel stats of the same command
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
OUT=gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/r05d_smoke.log 2>&1; echo "smoke rc=$?" | tee -a $OUT/r05d_smoke.log; tail -7 $OUT/r05d_smoke.log
timeout 900 python bench.py --gpus 1 --steps 5 --warmup 2 > $OUT/r05d_bench.json 2> $OUT/r05d_bench.err; echo "bench rc=$?"; cut -c1-1500 $OUT/r05d_bench.json; tail -5 $OUT/r05d_bench.err
timeout 1500 python -m pytest tests -m gpu -q -s -x > $OUT/r05d_pytest_gpu.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/r05d_pytest_gpu.log; grep -v "amdgpu.ids" $OUT/r05d_pytest_gpu.log | tail -40
export TMPDIR=/tmp
( cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats -d "$OLDPWD/$OUT/prof_f16x2" -o bench -- python "$OLDPWD/bench.py" --gpus 1 --steps 3 --warmup 1 --skip-extras > "$OLDPWD/$OUT/r05d_bench_prof.json" 2> "$OLDPWD/$OUT/prof.log" ); echo "rocprof rc=$?"
python tools/rocpd_stats.py $OUT/prof_f16x2/bench_results.db > $OUT/r05d_bench_f16x2_kernel_stats.csv 2>/dev/null; head -16 $OUT/r05d_bench_f16x2_kernel_stats.csv | cut -c1-180
