#!/bin/bash
# round 6, run ac: GEGLU in the feed-forward projection's epilogue for the f32-storage modes (generic row loop of epilogue_wide): LDM tests on the device,
# the 128-hypothesis forward per mode with NOPE_GEGLU_FUSED_F32=0/1, same box.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out; mkdir -p $OUT
export HSA_ENABLE_IPC_MODE_LEGACY=0 TMPDIR=/tmp
timeout 900 python -m pytest tests -q -m gpu -k "ldm or token or geglu" > $OUT/r06ac_pytest_ldm.log 2>&1; echo "pytest rc=$?"; tail -3 $OUT/r06ac_pytest_ldm.log
for dt in bf16x3 f16x2 f32; do
  for a in 0 1 0 1; do
    echo -n "NOPE_GEGLU_FUSED_F32=$a: "; NOPE_GEGLU_FUSED_F32=$a timeout 300 python tools/ldm_step.py 128 --dtype $dt 2>&1 | grep LDM
  done
done | tee $OUT/r06ac_ldm_geglu_ab.txt
