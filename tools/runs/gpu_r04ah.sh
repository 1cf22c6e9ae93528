#!/bin/bash
# round 4, run ah: non-temporal loads / stores in gn_apply (NOPE_GN_NT bits: 1 x loads, 2 y stores, 4 residual loads): whole step and the kernel alone
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
OUT=$PWD/gpurun_out
export TMPDIR=/tmp
timeout 900 python tools/small_bank_sweep.py --dtype f16 --banks 64,512 --steps 30 --settings ";NOPE_GN_NT=1;NOPE_GN_NT=2;NOPE_GN_NT=3;NOPE_GN_NT=5;NOPE_GN_NT=7;;NOPE_GN_NT=3;NOPE_GN_NT=7" > $OUT/gn_nt_step_sweep.txt 2>$OUT/sweep.err; cat $OUT/gn_nt_step_sweep.txt
for nt in 0 3 7; do
for r in 0 1; do
( cd /tmp && rm -rf /tmp/prof_gn && NOPE_GN_NT=$nt timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_gn -o g -- python $OLDPWD/tools/gn_bench.py --resid $r --emb $r > /dev/null 2>&1 )
echo "NOPE_GN_NT=$nt resid=$r"; python tools/rocpd_stats.py $(find /tmp/prof_gn -name "*.db" | head -1) | grep gn_apply | cut -c1-60,150-230
done; done 2>&1 | tee $OUT/gn_nt_kernel.txt
