#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
OUT=gpurun_out/pp_ablation6.txt
: > $OUT
timeout 600 python -m pytest tests/test_conv_pingpong.py -m gpu -x -q > gpurun_out/pytest_pp.log 2>&1; echo "pp rc=$?" >> $OUT; tail -3 gpurun_out/pytest_pp.log >> $OUT
for cfg in "3 0" "3 64" "3 16" "3 32" "3 48" "3 0"; do
  set -- $cfg
  echo "## NOPE_CONV_PP=$1 NOPE_PP_VARIANT=$2" >> $OUT
  NOPE_PP_VARIANT=$2 timeout 300 python tools/conv_bench.py --pp $1 --rounds 2 --reps 5 --only 0,1,2,4,6 >> $OUT 2>&1
done
timeout 600 python tools/conv_bench.py --pp 0,1,3 --rounds 3 --reps 5 >> $OUT 2>&1
for pp in 0 1 0 1; do
  NOPE_CONV_PP=$pp timeout 300 python bench.py --steps 10 --warmup 3 --skip-extras > gpurun_out/bench_pp$pp.json 2> gpurun_out/bench_pp$pp.err; echo "bench pp=$pp rc=$?" >> $OUT
  python -c "import json;d=json.load(open('gpurun_out/bench_pp$pp.json'));print('pp=$pp', d['value'], d['ms_per_step'], d['config']['top5'])" >> $OUT
done
cat $OUT
