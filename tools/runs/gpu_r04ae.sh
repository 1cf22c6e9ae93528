#!/bin/bash
# round 4, run ae: bench line with the per-launch median of three profiled steps + rocprofv3 kernel stats of the same command (same box)
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
OUT=gpurun_out
timeout 900 python bench.py > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"; tail -2 $OUT/bench.err
python - <<'PY'
import json
d=json.load(open('gpurun_out/bench.json'))
r=d['roofline']
print('bench', d['dtype'], round(d['value']), round(d['ms_per_step'],2), 'ms')
print(r['kernel'], round(r['frac'],3), 'traffic', r['traffic'], 'avg_launch_ms', r['avg_launch_ms'], 'kernel_ms_per_step', r['kernel_ms_per_step'], 'family frac', round(r['family']['frac'],3))
PY
export TMPDIR=/tmp
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_bench -o bench -- python "$OLDPWD/bench.py" --gpus 1 --steps 5 --warmup 2 --skip-extras > "$OLDPWD/$OUT/prof.log" 2>&1 ); echo "rocprof rc=$?"
python tools/rocpd_stats.py $(find /tmp/prof_bench -name "*.db" | head -1) > $OUT/bench_f16_kernel_stats.csv; head -4 $OUT/bench_f16_kernel_stats.csv | cut -c1-150; grep "halo_kernelIDF16_Lb0ELb1" $OUT/bench_f16_kernel_stats.csv | cut -c1-150
