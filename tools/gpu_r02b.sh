#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
OUT=gpurun_out/pp_ablation8.txt
: > $OUT
timeout 600 python -m pytest tests/test_conv_pingpong.py -m gpu -x -q > gpurun_out/pytest_pp.log 2>&1; echo "pp rc=$?" >> $OUT; tail -3 gpurun_out/pytest_pp.log >> $OUT
for lib in prev new prev new; do
  echo "## lib=$lib" >> $OUT
  if [ $lib = prev ]; then export NOPE_HIP_LIB=$PWD/build/ab/libnope_hip_prev.so; else unset NOPE_HIP_LIB; fi
  timeout 300 python tools/conv_bench.py --pp 3 --rounds 2 --reps 5 --only 0,1,2,4,6 >> $OUT 2>&1
done
unset NOPE_HIP_LIB
for v in 16 32; do echo "## new variant $v" >> $OUT; NOPE_PP_VARIANT=$v timeout 300 python tools/conv_bench.py --pp 3 --rounds 2 --reps 5 --only 2,6 >> $OUT 2>&1; done
for lib in prev new prev new; do
  if [ $lib = prev ]; then export NOPE_HIP_LIB=$PWD/build/ab/libnope_hip_prev.so; else unset NOPE_HIP_LIB; fi
  timeout 300 python bench.py --steps 10 --warmup 3 --skip-extras > gpurun_out/bench_$lib.json 2> gpurun_out/bench_$lib.err; echo "bench $lib rc=$?" >> $OUT
  python -c "import json;d=json.load(open('gpurun_out/bench_$lib.json'));print('$lib', d['value'], d['ms_per_step'], d['config']['top5'])" >> $OUT
done
cat $OUT
