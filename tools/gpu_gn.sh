#!/bin/bash
# GroupNorm apply A/B: working tree vs build/ab/libnope_hip_prev.so, per shape under rocprofv3 (kernel trace)
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
OUT=$PWD/gpurun_out
export TMPDIR=/tmp
: > $OUT/gn_bench.txt
for lib in prev new prev new; do
  unset NOPE_HIP_LIB
  if [ $lib = prev ]; then export NOPE_HIP_LIB=$PWD/build/ab/libnope_hip_prev.so; fi
  for args in "--act 1 --emb 1" "--act 1 --resid 1"; do
    echo "## lib=$lib $args" >> $OUT/gn_bench.txt
    timeout 200 python tools/gn_bench.py $args 2>/dev/null | grep "^gn " >> $OUT/gn_bench.txt
  done
done
unset NOPE_HIP_LIB
for lib in prev new; do
  if [ $lib = prev ]; then export NOPE_HIP_LIB=$PWD/build/ab/libnope_hip_prev.so; else unset NOPE_HIP_LIB; fi
  timeout 300 python bench.py --steps 10 --warmup 3 --skip-extras > $OUT/b.json 2>/dev/null
  python -c "import json;d=json.load(open('$OUT/b.json'));print('bench lib=$lib', round(d['value']), round(d['ms_per_step'],3))" >> $OUT/gn_bench.txt
done
cat $OUT/gn_bench.txt
