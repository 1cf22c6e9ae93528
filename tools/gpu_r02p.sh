#!/bin/bash
# evidence extras for the final tree: LDM variant per-kernel split (MFMA token attention), phase timeline of the tap-resident kernel
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
for v in 0 1; do NOPE_LDM_ATTN=$v python tools/ldm_step.py 128 2>/dev/null | grep LDM | sed "s/^/NOPE_LDM_ATTN=$v: /"; done > gpurun_out/ldm_step.txt
( cd /tmp && rocprofv3 --kernel-trace --stats -d /tmp/ldmprof -o l -- python $GRAFT_REPO_ROOT/tools/ldm_step.py 128 > /dev/null 2>&1 )
python tools/rocpd_stats.py $(find /tmp/ldmprof -name "*.db" | head -1) > gpurun_out/ldm_kernel_stats.csv
cat gpurun_out/ldm_step.txt; head -12 gpurun_out/ldm_kernel_stats.csv
timeout 200 python tools/pp_timeline.py 2>/dev/null | grep -v amdgpu.ids > gpurun_out/timeline.txt; head -12 gpurun_out/timeline.txt
