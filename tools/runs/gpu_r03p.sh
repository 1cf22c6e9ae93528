#!/bin/bash
# round 3, run p: fused-PreNorm 1x1 convs (qkv projections) through the packed 16-bit epilogue panel when a wave's 64 rows are one
# sample -- against the build before (build/ab/libnope_hip_prev.so), whole step, alternating; then the per-launch-shape table
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
OUT=$PWD/gpurun_out
export TMPDIR=/tmp
: > $OUT/pn_epilogue_ab.txt
for lib in prev new prev new; do
  if [ $lib = prev ]; then export NOPE_HIP_LIB=$PWD/build/ab/libnope_hip_prev.so; else unset NOPE_HIP_LIB; fi
  timeout 300 python bench.py --steps 10 --warmup 3 --skip-extras > $OUT/b.json 2>/dev/null
  python -c "import json;d=json.load(open('$OUT/b.json'));print('bench $lib', round(d['value']), round(d['ms_per_step'],3))" >> $OUT/pn_epilogue_ab.txt
done
unset NOPE_HIP_LIB
timeout 120 python -m pytest tests/test_conv_pingpong.py -m gpu -q -x -k small_shapes -s 2>&1 | grep -i "U-Net\|passed\|failed" >> $OUT/pn_epilogue_ab.txt
cat $OUT/pn_epilogue_ab.txt
