#!/bin/bash
# round 4, run h: two K groups per tile (tile 3) for few-tile / long-K launches: GPU tests, the sweep with it on / off, the encoder.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
OUT=$PWD/gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_conv_small.py -m gpu -x -q > $OUT/pytest_r04h.log 2>&1; echo "tests rc=$?"; tail -5 $OUT/pytest_r04h.log
timeout 900 python tools/small_bank_sweep.py --dtype f16 --banks 26,64,91,341 --settings ";NOPE_SMALL_KG2_MAX=0;NOPE_SMALL_KG2_MAX=512;NOPE_SMALL_KG2_MAX=128" > $OUT/small_bank_sweep.txt 2>$OUT/sweep.err; cat $OUT/small_bank_sweep.txt
timeout 300 python tools/encoder_bench.py > $OUT/encoder_bench.txt 2>&1; cat $OUT/encoder_bench.txt
NOPE_SMALL_KG2_MAX=0 timeout 300 python tools/encoder_bench.py > $OUT/encoder_bench_kg1.txt 2>&1; cat $OUT/encoder_bench_kg1.txt
echo done
