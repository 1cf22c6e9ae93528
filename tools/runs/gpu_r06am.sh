#!/bin/bash
# round 6, run am: the branch-free GELU (Abramowitz & Stegun 7.1.26 on v_rcp_f32 / v_exp_f32) in the GEGLU epilogue / kernel of every mode but f32:
# LDM tests on the device, the 128-hypothesis forward per mode against a build with the exact erff everywhere (-DNOPE_GELU_FAST=0, loaded through NOPE_HIP_LIB), interleaved.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out; mkdir -p $OUT
export HSA_ENABLE_IPC_MODE_LEGACY=0 TMPDIR=/tmp
timeout 900 python -m pytest tests -q -m gpu -s -k "ldm or geglu or gelu or token" > $OUT/r06am_pytest_ldm.log 2>&1; echo "pytest rc=$?"; grep -E "LDM variant|passed|failed" $OUT/r06am_pytest_ldm.log | tail -8
for dt in bf16 f16 f16x2 bf16x3; do
  for v in gelu0 fast gelu0 fast; do
    if [ $v = gelu0 ]; then export NOPE_HIP_LIB=$PWD/nope_amd/csrc/libnope_hip_gelu0.so; else unset NOPE_HIP_LIB; fi
    echo -n "$v: "; timeout 300 python tools/ldm_step.py 128 --dtype $dt 2>&1 | grep LDM
  done
done | tee $OUT/r06am_ldm_fast_gelu_ab.txt
