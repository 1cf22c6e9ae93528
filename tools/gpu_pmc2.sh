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
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_INSTS_MFMA SQ_INST_CYCLES_VMEM SQ_INST_LEVEL_LDS SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU -d $OUT/c -o p -- python $OLDPWD/tools/conv_bench.py --reps 3 --only $IDX > /dev/null 2>&1 )
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc TA_TA_BUSY_sum TA_BUSY_max TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TD_TCP_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum -d $OUT/d -o p -- python $OLDPWD/tools/conv_bench.py --reps 3 --only $IDX > $OLDPWD/gpurun_out/pmc2_d.log 2>&1 )
python tools/rocpd_pmc.py $(find $OUT -name "*.db") 2>&1 | grep -A32 "conv_gemm" | head -34 > gpurun_out/pmc2_$IDX.txt
cat gpurun_out/pmc2_$IDX.txt
