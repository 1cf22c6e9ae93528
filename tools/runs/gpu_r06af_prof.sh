# round 6, run af (profile leg): duration of gn_apply_proj_kernel in the default bench step per pixels-per-workgroup setting
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export HSA_ENABLE_IPC_MODE_LEGACY=0 TMPDIR=/tmp
mkdir -p gpurun_out
for px in 64 128 256 512 1024; do
( cd /tmp && rm -rf /tmp/prof_t && NOPE_PROJ_PIXELS=$px timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_t -o b -- python $OLDPWD/bench.py --gpus 1 --steps 5 --warmup 2 --skip-extras > /tmp/prof_t.log 2>&1 )
python tools/rocpd_stats.py $(find /tmp/prof_t -name "*.db" | head -1) > /tmp/stats_$px.csv
echo -n "NOPE_PROJ_PIXELS=$px: "; grep -i "proj" /tmp/stats_$px.csv | cut -d, -f2- | cut -c1-120 | tail -1
done | tee gpurun_out/r06af_proj_pixels_sweep.txt
