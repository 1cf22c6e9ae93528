#!/bin/bash
# round 6, run f: where does a 64-template (8-way shard) and a 341-template step go in f16x2?  rocprofv3 kernel traces.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out; mkdir -p $OUT
export HSA_ENABLE_IPC_MODE_LEGACY=0 TMPDIR=/tmp
for n in 64 341; do
  ( cd /tmp && timeout 600 rocprofv3 --kernel-trace -d "$OLDPWD/$OUT/prof_r06f_$n" -o bench -- python "$OLDPWD/bench.py" --gpus 1 --steps 20 --warmup 5 --templates $n --skip-extras > "$OLDPWD/$OUT/r06f_bench_$n.json" 2> "$OLDPWD/$OUT/r06f_prof_$n.log" ); echo "rocprof $n rc=$?"
  python tools/rocpd_stats.py $(find $OUT/prof_r06f_$n -name "*.db" | head -1) > $OUT/r06f_kernel_stats_$n.csv 2>&1 || true
  head -32 $OUT/r06f_kernel_stats_$n.csv | cut -c1-200; cat $OUT/r06f_bench_$n.json | python -c "import json,sys; r=json.load(sys.stdin); print(r['ms_per_step'])"
done
