#!/bin/bash
# round 6, run ab: (1) where a 64-template f16x2 step goes (an 8-way shard of configs[1]; each query waiting for the previous one): rocprofv3 kernel
# trace, the last step dispatch by dispatch with the gaps; (2) the LDM variant's f16x2 / bf16x3 forward after the split-precision attention.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out; mkdir -p $OUT
export HSA_ENABLE_IPC_MODE_LEGACY=0 TMPDIR=/tmp
for n in 64 341; do
( cd /tmp && rm -rf /tmp/prof_b$n && timeout 600 rocprofv3 --kernel-trace -d /tmp/prof_b$n -o b -- python $OLDPWD/tools/small_bank_sweep.py --dtype f16x2 --banks $n --steps 6 --settings "NOPE_PIPELINE_ENCODERS=0" > $OLDPWD/$OUT/r06ab_prof_bank$n.log 2>&1 ); tail -1 $OUT/r06ab_prof_bank$n.log
python tools/rocpd_timeline.py $(find /tmp/prof_b$n -name "*.db" | head -1) --last 700 > $OUT/r06ab_timeline_bank${n}_f16x2_last700.csv
done
d=f16x2
( cd /tmp && rm -rf /tmp/prof_l_$d && timeout 600 rocprofv3 --kernel-trace -d /tmp/prof_l_$d -o b -- python $OLDPWD/tools/ldm_step.py 128 --dtype $d > $OLDPWD/$OUT/r06ab_prof_ldm_$d.log 2>&1 ); tail -1 $OUT/r06ab_prof_ldm_$d.log
python tools/rocpd_timeline.py $(find /tmp/prof_l_$d -name "*.db" | head -1) > /tmp/timeline_ldm_$d.csv
python tools/timeline_summary.py /tmp/timeline_ldm_$d.csv 4 72 | head -70 > $OUT/r06ab_timeline_ldm_128_${d}_summary.txt; head -30 $OUT/r06ab_timeline_ldm_128_${d}_summary.txt | cut -c1-190
