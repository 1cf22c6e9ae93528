#!/bin/bash
# round 4, run aa: GEGLU in the feed-forward projection's epilogue (LDM variant): same-box A/B + the LDM tests
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
OUT=$PWD/gpurun_out
timeout 900 python -m pytest tests/test_kernels_parity.py tests/test_gpu_fullsize.py -m gpu -x -q -k "geglu or ldm" 2>&1 | tail -3
for i in 1 2; do
for n in 128 26; do
timeout 300 python tools/ldm_step.py $n 2>/dev/null | tail -1 | sed 's/^/fused    /'
NOPE_GEGLU_FUSED=0 timeout 300 python tools/ldm_step.py $n 2>/dev/null | tail -1 | sed 's/^/unfused  /'
done; done | tee $OUT/ldm_geglu_fused_ab.txt
