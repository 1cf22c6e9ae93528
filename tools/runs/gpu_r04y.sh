#!/bin/bash
# round 4, run y: XCD map for panel counts outside {1,2,4,8} (LDM linears): same-box A/B on the LDM forward + its tests
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
OUT=$PWD/gpurun_out
for i in 1 2; do
timeout 300 python tools/ldm_step.py 128 2>/dev/null | tail -1 | sed 's/^/map4    /'
NOPE_XCD_ANY=0 timeout 300 python tools/ldm_step.py 128 2>/dev/null | tail -1 | sed 's/^/xcd_any=0 /'
done | tee $OUT/ldm_xcd_any_ab.txt
timeout 900 python -m pytest tests/test_kernels_parity.py -m gpu -x -q -k "any_panel_count or ldm" 2>&1 | tail -3
