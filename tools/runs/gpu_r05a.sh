#!/bin/bash
# round 5, run a: MX-scaled MFMA / fp8 conversion facts (tools/probes/mx_probe.hip) + the per-kernel profile of the bf16x3 step the f16x2 mode is measured against
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
OUT=gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 120 tools/probes/mx_probe > $OUT/r05a_mx_probe.txt 2>&1; echo "probe rc=$?" | tee -a $OUT/r05a_mx_probe.txt
cat $OUT/r05a_mx_probe.txt
export TMPDIR=/tmp
( cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats -d "$OLDPWD/$OUT/prof_bf16x3" -o bench -- python "$OLDPWD/bench.py" --gpus 1 --steps 3 --warmup 1 --dtype bf16x3 --skip-extras > "$OLDPWD/$OUT/r05a_bench_bf16x3.json" 2> "$OLDPWD/$OUT/prof.log" ); echo "rocprof rc=$?"
cat $OUT/r05a_bench_bf16x3.json | cut -c1-600
f=$(find $OUT/prof_bf16x3 -name "*kernel_stats.csv" | head -1); if [ -n "$f" ]; then cp "$f" $OUT/r05a_bench_bf16x3_kernel_stats.csv; head -24 "$f" | cut -c1-200; fi; true
