#!/bin/bash
# round 6, run al: where the LDM variant's forward goes on the final tree: rocprofv3 kernel traces of tools/ldm_step.py (128 hypotheses, shipped size) in bf16 and f16x2,
# summarised per launch shape (as r06t, before this round's LDM changes; create-time kernels are folded into the per-step column: the tool builds the model in the traced process)
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out; mkdir -p $OUT
export HSA_ENABLE_IPC_MODE_LEGACY=0 TMPDIR=/tmp
for d in bf16 f16x2; do
  ( cd /tmp && rm -rf /tmp/prof_l_$d && timeout 600 rocprofv3 --kernel-trace -d /tmp/prof_l_$d -o b -- python $OLDPWD/tools/ldm_step.py 128 --dtype $d > $OLDPWD/$OUT/r06al_prof_ldm_$d.log 2>&1 ); tail -1 $OUT/r06al_prof_ldm_$d.log
  python tools/rocpd_timeline.py $(find /tmp/prof_l_$d -name "*.db" | head -1) > /tmp/timeline_ldm_$d.csv
  python tools/timeline_summary.py /tmp/timeline_ldm_$d.csv 4 72 | head -70 > $OUT/r06al_timeline_ldm_128_${d}_summary.txt; head -24 $OUT/r06al_timeline_ldm_128_${d}_summary.txt | cut -c1-170
done
