#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_kernels_parity.py tests/test_gpu_configs.py -m gpu -q -s -k "ldm" > gpurun_out/pytest_ldm.log 2>&1; echo "rc=$?"
grep -av "amdgpu.ids" gpurun_out/pytest_ldm.log | tail -30
