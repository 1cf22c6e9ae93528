#!/bin/bash
# round 4, run x: where the LDM variant's forward goes, per (kernel, grid): 128 hypotheses at a 32 x 32 latent, bf16, four forwards
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
OUT=$PWD/gpurun_out
export TMPDIR=/tmp
( cd /tmp && rm -rf /tmp/prof_l && NOPE_CONV_TRACE=0 timeout 600 rocprofv3 --kernel-trace -d /tmp/prof_l -o b -- python $OLDPWD/tools/ldm_step.py 128 > $OUT/prof_ldm.log 2>&1 ); tail -2 $OUT/prof_ldm.log
python tools/rocpd_timeline.py $(find /tmp/prof_l -name "*.db" | head -1) > $OUT/timeline_ldm.csv
python tools/timeline_summary.py $OUT/timeline_ldm.csv 4 | head -90 | tee $OUT/timeline_ldm_summary.txt
rm -f $OUT/timeline_ldm.csv
