#!/bin/bash
# PMC passes over one conv shape under given kernel settings:  bash tools/gpu_pmc2.sh <tag> <NOPE_CONV_PP> <NOPE_PP_VARIANT> <shape index>
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
TAG=$1; PP=$2; VAR=$3; SH=$4
OUT=$PWD/gpurun_out/pmc/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
export NOPE_PP_VARIANT=$VAR
CMD="$PWD/tools/conv_bench.py --pp $PP --rounds 1 --reps 3 --only $SH"
run() { name=$1; shift; ( cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc "$@" -d $OUT/$name -o p -- python $CMD > $OUT/$name.log 2>&1 ); echo "$name rc=$?"; }
run sq SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE
run sq2 SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL
run grbm GRBM_GUI_ACTIVE
python tools/rocpd_pmc.py $(find $OUT -name "*.db") > gpurun_out/pmc_$TAG.txt 2>&1
rm -rf $OUT
grep -A30 "halo\|pp_kernel" gpurun_out/pmc_$TAG.txt | head -40
