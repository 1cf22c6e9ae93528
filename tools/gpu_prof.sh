#!/bin/bash
# rocprofv3 kernel trace of a bench run, summarised on the box (the .db is too big to ship back).
#   gpurun -- 'bash tools/gpu_prof.sh <tag> [bench args]'
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
TAG=$1; shift
mkdir -p gpurun_out
export TMPDIR=/tmp
D=/tmp/prof_$TAG
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $D -o bench -- python $OLDPWD/bench.py --gpus 1 --steps 3 --warmup 1 --skip-extras "$@" > $OLDPWD/gpurun_out/prof_$TAG.log 2>&1 )
python tools/rocpd_stats.py $(find $D -name "*.db" | head -1) > gpurun_out/kernel_stats_$TAG.csv
head -14 gpurun_out/kernel_stats_$TAG.csv | cut -c1-130
