#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
OUT=gpurun_out/pp_ablation5.txt
: > $OUT
for cfg in "3 48" "19 48" "19 49" "3 112" "19 113" "3 16" "19 17" "3 0" "19 1"; do
  set -- $cfg
  echo "## NOPE_CONV_PP=$1 NOPE_PP_VARIANT=$2" >> $OUT
  NOPE_PP_VARIANT=$2 timeout 300 python tools/conv_bench.py --pp $1 --rounds 2 --reps 5 --only 2,6 >> $OUT 2>&1
done
cat $OUT
