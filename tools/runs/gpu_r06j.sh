#!/bin/bash
# round 6, run j: start skew of the persistent 1x1 launches (NOPE_DMA_SKEW = n x 8 k cycles for the second half of the grid; + 256: every other
# workgroup of an XCD instead) -- do the read phase and the write phase of an HBM-bound launch overlap once the workgroups are out of step?
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out; mkdir -p $OUT
export HSA_ENABLE_IPC_MODE_LEGACY=0 TMPDIR=/tmp
S="NOPE_CONV_STREAM=0;NOPE_CONV_STREAM=0,NOPE_DMA_SKEW=1;NOPE_CONV_STREAM=0,NOPE_DMA_SKEW=2;NOPE_CONV_STREAM=0,NOPE_DMA_SKEW=3;NOPE_CONV_STREAM=0,NOPE_DMA_SKEW=5;NOPE_CONV_STREAM=0,NOPE_DMA_SKEW=258;NOPE_CONV_STREAM=0,NOPE_DMA_SKEW=260;NOPE_CONV_STREAM=1,NOPE_DMA_SKEW=2;NOPE_CONV_STREAM=1,NOPE_DMA_SKEW=4"
timeout 600 python tools/stream_bench.py --dtype bf16x3 --settings "$S" > $OUT/r06j_skew_bf16x3.txt 2>&1; cut -c1-400 $OUT/r06j_skew_bf16x3.txt
