#!/bin/bash
# round 3, run n: packed epilogue with the row offsets taken once per tile, pair arithmetic (bias, column statistics) and a real
# branch on `act` -- against the build before (build/ab/libnope_hip_prev.so): per conv shape and on the whole step
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
OUT=$PWD/gpurun_out
export TMPDIR=/tmp
: > $OUT/epilogue_ab.txt
for lib in prev new; do
  echo "## $lib" >> $OUT/epilogue_ab.txt
  if [ $lib = prev ]; then export NOPE_HIP_LIB=$PWD/build/ab/libnope_hip_prev.so; else unset NOPE_HIP_LIB; fi
  timeout 300 python tools/conv_bench.py --dtype f16 --pp 3 --rounds 3 2>/dev/null | grep -v amdgpu.ids >> $OUT/epilogue_ab.txt
done
for lib in prev new prev new; do
  if [ $lib = prev ]; then export NOPE_HIP_LIB=$PWD/build/ab/libnope_hip_prev.so; else unset NOPE_HIP_LIB; fi
  timeout 300 python bench.py --steps 10 --warmup 3 --skip-extras > $OUT/b.json 2>/dev/null
  python -c "import json;d=json.load(open('$OUT/b.json'));print('bench $lib', round(d['value']), round(d['ms_per_step'],3))" >> $OUT/epilogue_ab.txt
done
unset NOPE_HIP_LIB
cat $OUT/epilogue_ab.txt
