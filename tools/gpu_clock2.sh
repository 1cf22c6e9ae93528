#!/bin/bash
# Duration and implied shader clock of one conv shape under tuning variants:  gpurun -- 'bash tools/gpu_clock2.sh <shape> <variants...>'
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
IDX=$1; shift
for v in "$@"; do
  D=/tmp/clk2_$v
  ( cd /tmp && NOPE_CONV_VARIANT=$v timeout 300 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES -d $D -o p -- python $OLDPWD/tools/conv_bench.py --reps 5 --only $IDX > /dev/null 2>&1 )
  echo "variant $v"; python tools/rocpd_clock.py $(find $D -name "*.db" | head -1) | grep conv_gemm
  python tools/rocpd_pmc.py $(find $D -name "*.db" | head -1) | grep -A3 conv_gemm | grep -v GRBM | cut -c1-90
done 2>&1 | tee gpurun_out/clock2_$IDX.txt
