#!/bin/bash
# round 3, run s: cycle stamps of the tap-resident kernel (NOPE_PP_VARIANT=256) on the final tree -- the epilogue after the trimming
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
timeout 200 python tools/pp_timeline.py 2>/dev/null | grep -v amdgpu.ids > gpurun_out/halo_phase_timeline.txt
cat gpurun_out/halo_phase_timeline.txt | cut -c1-260
