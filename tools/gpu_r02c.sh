#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
OUT=gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 600 python -m pytest tests/test_conv_pingpong.py -m gpu -x -q -s > $OUT/pytest_pp.log 2>&1; echo "pp rc=$?" | tee -a $OUT/pytest_pp.log
tail -8 $OUT/pytest_pp.log
timeout 600 python tools/conv_bench.py --pp 0,17,1,3 --rounds 3 --reps 5 > $OUT/conv_bench2.txt 2>&1; echo "conv_bench rc=$?"
cat $OUT/conv_bench2.txt
for pp in 0 17 1 3 0 1 3; do
  NOPE_CONV_PP=$pp timeout 300 python bench.py --steps 10 --warmup 3 --skip-extras > $OUT/bench_pp$pp.json 2> $OUT/bench_pp$pp.err; echo "bench pp=$pp rc=$?"
  python -c "import json;d=json.load(open('$OUT/bench_pp$pp.json'));print('pp=$pp', d['value'], d['ms_per_step'], d['config']['top5'])"
done
