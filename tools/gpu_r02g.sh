#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
: > gpurun_out/sim_bench.txt
for v in 0 1 2 3 0 3; do NOPE_SIM_VARIANT=$v timeout 200 python tools/sim_bench.py >> gpurun_out/sim_bench.txt 2>&1; done
grep -v amdgpu gpurun_out/sim_bench.txt
