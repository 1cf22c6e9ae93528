#!/bin/bash
# round 6, run ad: the LDM variant's 1x1 convs / linears with a second (NOPE_F16X2) weight pack -- taken by the launches the per-tap ping-pong kernel runs,
# input maxima from absmax passes where the producer records none: LDM tests on the device, 128-hypothesis forward with NOPE_LDM_X2_1X1=0/1, same box.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out; mkdir -p $OUT
export HSA_ENABLE_IPC_MODE_LEGACY=0 TMPDIR=/tmp
timeout 900 python -m pytest tests -q -m gpu -s -k "ldm" > $OUT/r06ad_pytest_ldm.log 2>&1; echo "pytest rc=$?"; grep -E "LDM|passed|failed|Error" $OUT/r06ad_pytest_ldm.log | tail -12
for a in 0 1 0 1; do
  echo -n "NOPE_LDM_X2_1X1=$a: "; NOPE_LDM_X2_1X1=$a timeout 300 python tools/ldm_step.py 128 --dtype f16x2 2>&1 | grep LDM
done | tee $OUT/r06ad_ldm_x2_1x1_ab.txt
NOPE_CONV_TRACE=1 timeout 300 python tools/ldm_step.py 128 --dtype f16x2 2>&1 | grep "^conv" | sort | uniq -c | sort -rn | head -60 > $OUT/r06ad_ldm_conv_trace_f16x2.txt
