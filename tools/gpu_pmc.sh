#!/bin/bash
# PMC passes (counters only, --kernel-trace; one pass per counter set) over a command.
#   gpurun -- 'bash tools/gpu_pmc.sh <tag> <python script + args>'
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
TAG=$1; shift
OUT=$PWD/gpurun_out/pmc/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
run() { name=$1; shift; ( cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc "$@" -d $OUT/$name -o p -- python $CMD > $OUT/$name.log 2>&1 ); echo "$name rc=$?"; }
CMD="$*"
CMD="${CMD/#tools/$PWD/tools}"; CMD="${CMD/#bench.py/$PWD/bench.py}"
run sq SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE
run fetch FETCH_SIZE
run write WRITE_SIZE
run l2 TCC_HIT_sum TCC_MISS_sum
python tools/rocpd_pmc.py $(find $OUT -name "*.db") > gpurun_out/pmc_$TAG.txt 2>&1
rm -rf $OUT
head -60 gpurun_out/pmc_$TAG.txt
