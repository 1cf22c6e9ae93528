#!/bin/bash
# round 4, run ab: fused GEGLU everywhere vs only where the 128 x 192 kernel would run the projection anyway (level 2: ping-pong kernel + geglu_kernel)
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
OUT=$PWD/gpurun_out
for i in 1 2; do
for n in 128 26 8; do
timeout 300 python tools/ldm_step.py $n 2>/dev/null | tail -1 | sed 's/^/fused=1  /'
NOPE_GEGLU_FUSED=2 timeout 300 python tools/ldm_step.py $n 2>/dev/null | tail -1 | sed 's/^/fused=2  /'
done; done | tee $OUT/ldm_geglu_fused_mode_ab.txt
