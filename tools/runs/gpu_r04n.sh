#!/bin/bash
# round 4, run n: same-box A/B of the 16-row statistics of the wide epilogue (NOPE_STATS16=0: the gn_stats passes of round 3)
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
OUT=$PWD/gpurun_out
timeout 900 python tools/small_bank_sweep.py --dtype f16 --banks 341,512 --steps 40 --settings ";NOPE_STATS16=0;;NOPE_STATS16=0" > $OUT/stats16_ab.txt 2>$OUT/sweep.err; cat $OUT/stats16_ab.txt
echo done
