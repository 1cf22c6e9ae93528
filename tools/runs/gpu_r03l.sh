#!/bin/bash
# round 3, run l: linattn_kernel with its global loads issued ahead (all three sweeps) against the build before
# (build/ab/libnope_hip_prev.so), per level and on the whole step
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
OUT=$PWD/gpurun_out
export TMPDIR=/tmp
: > $OUT/attn_bench.txt
for lib in prev new prev new; do
  echo "## $lib" >> $OUT/attn_bench.txt
  if [ $lib = prev ]; then export NOPE_HIP_LIB=$PWD/build/ab/libnope_hip_prev.so; else unset NOPE_HIP_LIB; fi
  timeout 200 python tools/attn_bench.py 2>/dev/null | grep "^linattn" >> $OUT/attn_bench.txt
done
for lib in prev new prev new; do
  if [ $lib = prev ]; then export NOPE_HIP_LIB=$PWD/build/ab/libnope_hip_prev.so; else unset NOPE_HIP_LIB; fi
  timeout 300 python bench.py --steps 10 --warmup 3 --skip-extras > $OUT/b.json 2>/dev/null
  python -c "import json;d=json.load(open('$OUT/b.json'));print('bench $lib', round(d['value']), round(d['ms_per_step'],3))" >> $OUT/attn_bench.txt
done
unset NOPE_HIP_LIB
cat $OUT/attn_bench.txt
