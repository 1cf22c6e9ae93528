#!/bin/bash
# One GPU-box session: smoke, GPU parity tests, bench line, rocprofv3 kernel stats.
# Usage (from the build container):  gpurun --timeout 1500 -- 'bash tools/gpu_round.sh'
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
OUT=gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
echo "== smoke" | tee $OUT/smoke.log
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" >> $OUT/smoke.log 2>&1; echo "smoke rc=$?" | tee -a $OUT/smoke.log
tail -5 $OUT/smoke.log
echo "== pytest -m gpu"
timeout 900 python -m pytest tests -m gpu -x -q -s > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/pytest_gpu.log
tail -25 $OUT/pytest_gpu.log
echo "== bench"
timeout 900 python bench.py --gpus 1 --steps 5 --warmup 2 > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"
cat $OUT/bench.json; tail -5 $OUT/bench.err
if [ "${SKIP_PROF:-0}" != "1" ]; then
  echo "== rocprofv3 kernel stats"
  export TMPDIR=/tmp
  ( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d "$OLDPWD/$OUT/prof" -o bench -- python "$OLDPWD/bench.py" --gpus 1 --steps 3 --warmup 1 --skip-extras > "$OLDPWD/$OUT/prof.log" 2>&1 ); echo "rocprof rc=$?"
  find $OUT/prof -name "*stats*" | head; 
  f=$(find $OUT/prof -name "*kernel_stats.csv" | head -1); if [ -n "$f" ]; then head -30 "$f"; fi; true
fi
