#!/bin/bash
# round 6, run t: where does the LDM variant's forward go today?  rocprofv3 kernel traces of tools/ldm_step.py (128 hypotheses, shipped size)
# in bf16 and f16x2, summarised per launch shape (the last such profile is round 4's r04x).
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out; mkdir -p $OUT
export HSA_ENABLE_IPC_MODE_LEGACY=0 TMPDIR=/tmp
for d in bf16 f16x2; do
  ( cd /tmp && rm -rf /tmp/prof_l_$d && timeout 600 rocprofv3 --kernel-trace -d /tmp/prof_l_$d -o b -- python $OLDPWD/tools/ldm_step.py 128 --dtype $d > $OLDPWD/$OUT/r06t_prof_ldm_$d.log 2>&1 ); tail -1 $OUT/r06t_prof_ldm_$d.log
  python tools/rocpd_timeline.py $(find /tmp/prof_l_$d -name "*.db" | head -1) > /tmp/timeline_ldm_$d.csv
  python tools/timeline_summary.py /tmp/timeline_ldm_$d.csv 4 72 | head -70 > $OUT/r06t_timeline_ldm_128_${d}_summary.txt; head -45 $OUT/r06t_timeline_ldm_128_${d}_summary.txt | cut -c1-190
done
NOPE_CONV_TRACE=1 timeout 300 python tools/ldm_step.py 128 --dtype bf16 2>&1 | grep "^conv" | sort | uniq -c | sort -rn | head -60 > $OUT/r06t_ldm_conv_trace_bf16.txt
