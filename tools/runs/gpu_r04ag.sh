#!/bin/bash
# round 4, run ag: what a read-one / write-one kernel reaches on this part (hand-written copy kernels, not the runtime's blit)
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
timeout 300 tools/probes/copy_probe | tee gpurun_out/copy_probe.txt
