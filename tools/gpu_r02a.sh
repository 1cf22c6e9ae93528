#!/bin/bash
# Round 2, first GPU session: parity tests, conv kernel A/B, bench under the kernel policies.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
OUT=gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
echo "== pingpong tests first"
timeout 600 python -m pytest tests/test_conv_pingpong.py -m gpu -x -q -s > $OUT/pytest_pp.log 2>&1; echo "pp rc=$?" | tee -a $OUT/pytest_pp.log
tail -15 $OUT/pytest_pp.log
echo "== conv bench A/B"
timeout 600 python tools/conv_bench.py --pp 0,1,3,5 --rounds 3 --reps 5 > $OUT/conv_bench.txt 2>&1; echo "conv_bench rc=$?"
cat $OUT/conv_bench.txt
echo "== bench under policies"
for pp in 0 1 3 5; do
  NOPE_CONV_PP=$pp timeout 300 python bench.py --steps 10 --warmup 3 --skip-extras > $OUT/bench_pp$pp.json 2> $OUT/bench_pp$pp.err; echo "bench pp=$pp rc=$?"
  python -c "import json;d=json.load(open('$OUT/bench_pp$pp.json'));print('pp=$pp', d['value'], d['ms_per_step'], d['config']['top5'])"
done
echo "== full pytest -m gpu"
timeout 1200 python -m pytest tests -m gpu -x -q -s > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/pytest_gpu.log
tail -12 $OUT/pytest_gpu.log
