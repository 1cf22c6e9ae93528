#!/bin/bash
# round 3, run m: 4x4 / 8x8-level 3x3 convs -- tap-resident kernel in (sample, pixel) order (NOPE_CONV_PP=3, default: executes the
# padding taps) against the per-tap ping-pong kernel in position-major order (NOPE_CONV_PP=7: skips 31 % / 16 % of the MACs)
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
OUT=$PWD/gpurun_out
export TMPDIR=/tmp
: > $OUT/posmajor_pp.txt
timeout 300 python tools/conv_bench.py --dtype f16 --pp 3,7 --only 4,5,6,7 --rounds 3 2>/dev/null | grep -v amdgpu.ids >> $OUT/posmajor_pp.txt
for pp in 3 7 3 7; do
  NOPE_CONV_PP=$pp timeout 300 python bench.py --steps 10 --warmup 3 --skip-extras > $OUT/b.json 2>/dev/null
  python -c "import json;d=json.load(open('$OUT/b.json'));print('bench NOPE_CONV_PP=$pp', round(d['value']), round(d['ms_per_step'],3))" >> $OUT/posmajor_pp.txt
done
NOPE_POSMAJOR_HW=64 NOPE_CONV_PP=7 timeout 300 python bench.py --steps 10 --warmup 3 --skip-extras > $OUT/b.json 2>/dev/null
python -c "import json;d=json.load(open('$OUT/b.json'));print('bench NOPE_CONV_PP=7 NOPE_POSMAJOR_HW=64', round(d['value']), round(d['ms_per_step'],3))" >> $OUT/posmajor_pp.txt
cat $OUT/posmajor_pp.txt
