#!/bin/bash
# round 4, run l: two half batches on two streams with the round-4 kernels; the LDM variant's step.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
OUT=$PWD/gpurun_out
timeout 600 python tools/small_bank_sweep.py --dtype f16 --banks 26,64,91,341 --settings ";NOPE_TWO_STREAM_BELOW=400" > $OUT/small_bank_sweep_two_streams.txt 2>$OUT/sweep.err; cat $OUT/small_bank_sweep_two_streams.txt
timeout 300 python tools/ldm_step.py 128 > $OUT/ldm_step.txt 2>&1; timeout 300 python tools/ldm_step.py 26 >> $OUT/ldm_step.txt 2>&1; NOPE_CONV_SMALL=0 NOPE_HALO_SPLIT=0 timeout 300 python tools/ldm_step.py 26 >> $OUT/ldm_step.txt 2>&1; cat $OUT/ldm_step.txt
echo done
