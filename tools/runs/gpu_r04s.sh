#!/bin/bash
# round 4, run s: PMC passes over a 64-hypothesis U-Net batch: what the small-tile kernel and the split-K tap-resident kernel do with their cycles
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
bash tools/gpu_pmc.sh unet64 tools/unet_step.py --templates 64 --dtype f16 > /dev/null 2>&1; echo "pmc done"
grep -A12 "small_kernel\|halo_kernel\|reduce_stats\|gn_apply" gpurun_out/pmc_unet64.txt | head -150
