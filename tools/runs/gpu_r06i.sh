#!/bin/bash
# round 6, run i: ablations of the 128 x 192 LDS-DMA kernel on the HBM-bound 1x1 shapes (NOPE_CONV_VARIANT bits: 32 = no MFMA, 64 = no epilogue,
# 16 = no loads after the first stage, 1 = no raised priority) -- what bounds them?
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out; mkdir -p $OUT
export HSA_ENABLE_IPC_MODE_LEGACY=0 TMPDIR=/tmp
S="NOPE_CONV_STREAM=0;NOPE_CONV_STREAM=0,NOPE_CONV_VARIANT=32;NOPE_CONV_STREAM=0,NOPE_CONV_VARIANT=64;NOPE_CONV_STREAM=0,NOPE_CONV_VARIANT=96;NOPE_CONV_STREAM=0,NOPE_CONV_VARIANT=16;NOPE_CONV_STREAM=0,NOPE_CONV_VARIANT=48;NOPE_CONV_STREAM=0,NOPE_CONV_VARIANT=112;NOPE_CONV_STREAM=0,NOPE_CONV_PERSIST=0"
timeout 600 python tools/stream_bench.py --dtype bf16x3 --settings "$S" > $OUT/r06i_dma_ablations_bf16x3.txt 2>&1; cut -c1-400 $OUT/r06i_dma_ablations_bf16x3.txt
