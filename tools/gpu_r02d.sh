#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
NOPE_CONV_TRACE=1 timeout 300 python tools/unet_step.py 2> gpurun_out/conv_trace.txt > /dev/null
sort gpurun_out/conv_trace.txt | uniq -c | sort -rn > gpurun_out/conv_trace_counts.txt
for pp in 1 3; do
( cd /tmp && NOPE_CONV_PP=$pp timeout 600 rocprofv3 --kernel-trace --stats -d "$OLDPWD/gpurun_out/prof_pp$pp" -o bench -- python "$OLDPWD/bench.py" --gpus 1 --steps 4 --warmup 1 --skip-extras > "$OLDPWD/gpurun_out/prof_pp$pp.log" 2>&1 ); echo "rocprof rc=$?"
python tools/rocpd_stats.py $(find gpurun_out/prof_pp$pp -name "*.db" | head -1) > gpurun_out/kernel_stats_pp$pp.csv
rm -rf gpurun_out/prof_pp$pp
head -16 gpurun_out/kernel_stats_pp$pp.csv
done
