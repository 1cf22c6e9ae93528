#!/bin/bash
# round 5, run ac: comment-only edits of the kernel sources after run aa changed their hash: the PMC passes again (same machine code), so that
# profiles/pmc_traffic.json names the tree the driver benches
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
OUT=gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0 TMPDIR=/tmp
bash tools/gpu_pmc.sh unet tools/unet_step.py --dtype f16x2 > /dev/null 2>&1; echo "pmc unet done"; head -3 $OUT/pmc_unet.txt
bash tools/gpu_pmc.sh sim tools/sim_step.py > /dev/null 2>&1; echo "pmc sim done"
python tools/pmc_to_traffic.py $OUT/pmc_unet.txt $OUT/pmc_traffic.json --dtype f16x2 --sim $OUT/pmc_sim.txt | cut -c1-300
cp $OUT/pmc_unet.txt $OUT/r05ac_pmc_unet_f16x2.txt; cp $OUT/pmc_sim.txt $OUT/r05ac_pmc_sim_bf16.txt
