#!/bin/bash
# round 3, run r: planning traces on the final tree -- rocprofv3 kernel stats of the LDM variant's forward and of a 64-template step
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
OUT=gpurun_out
export TMPDIR=/tmp
( cd /tmp && timeout 200 rocprofv3 --kernel-trace --stats -d $OLDPWD/$OUT/prof_ldm -o p -- python $OLDPWD/tools/ldm_step.py > /dev/null 2>&1 )
python tools/rocpd_stats.py $(find $OUT/prof_ldm -name "*.db" | head -1) > $OUT/ldm_bf16_kernel_stats.csv 2>&1; head -14 $OUT/ldm_bf16_kernel_stats.csv | cut -c1-150
rm -rf $OUT/prof_ldm
( cd /tmp && timeout 200 rocprofv3 --kernel-trace --stats -d $OLDPWD/$OUT/prof_n64 -o p -- python $OLDPWD/bench.py --templates 64 --steps 20 --warmup 5 --skip-extras --dtype bf16 > /dev/null 2>&1 )
python tools/rocpd_stats.py $(find $OUT/prof_n64 -name "*.db" | head -1) > $OUT/bench_n64_kernel_stats.csv 2>&1; head -14 $OUT/bench_n64_kernel_stats.csv | cut -c1-150
rm -rf $OUT/prof_n64
