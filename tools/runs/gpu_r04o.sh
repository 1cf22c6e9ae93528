#!/bin/bash
# round 4, run o: same-box A/B: statistics fold inside gn_apply at every batch size (NOPE_GN_FOLD_INLINE large) against the 16 MiB rule
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
OUT=$PWD/gpurun_out
timeout 900 python tools/small_bank_sweep.py --dtype f16 --banks 341,512 --steps 40 --settings ";NOPE_GN_FOLD_INLINE=4000000000;;NOPE_GN_FOLD_INLINE=4000000000;NOPE_GN_FOLD_INLINE=67108864" > $OUT/fold_inline_ab.txt 2>$OUT/sweep.err; cat $OUT/fold_inline_ab.txt
echo done
