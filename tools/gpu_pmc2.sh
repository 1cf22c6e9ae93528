#!/bin/bash
# SQ issue/stall breakdown of one conv shape:  gpurun -- 'bash tools/gpu_pmc2.sh <shape-index>'
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
IDX=$1
OUT=/tmp/pmc2_$IDX
mkdir -p gpurun_out $OUT
export TMPDIR=/tmp
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY -d $OUT/a -o p -- python $OLDPWD/tools/conv_bench.py --reps 3 --only $IDX > /dev/null 2>&1 )
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_LDS SQ_INSTS_VALU SQ_INST_LEVEL_VMEM SQ_LEVEL_WAVES SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT -d $OUT/b -o p -- python $OLDPWD/tools/conv_bench.py --reps 3 --only $IDX > /dev/null 2>&1 )
python tools/rocpd_pmc.py $(find $OUT -name "*.db") 2>&1 | grep -A18 "conv_gemm" | head -20 > gpurun_out/pmc2_$IDX.txt
cat gpurun_out/pmc2_$IDX.txt
