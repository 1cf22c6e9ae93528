set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export HSA_ENABLE_IPC_MODE_LEGACY=0 TMPDIR=/tmp
mkdir -p gpurun_out
for dt in bf16 f16x2; do
echo -n "default: "; timeout 300 python tools/ldm_step.py 128 --dtype $dt 2>&1 | grep LDM
for mt in 700 1400 3000 100000; do
echo -n "NOPE_SMALL_TILE=1 NOPE_SMALL_MAX_TILES=$mt: "; NOPE_SMALL_TILE=1 NOPE_SMALL_MAX_TILES=$mt timeout 300 python tools/ldm_step.py 128 --dtype $dt 2>&1 | grep LDM
done
done | tee gpurun_out/r06aj_ldm_small_tile_128.txt
NOPE_CONV_TRACE=1 NOPE_SMALL_TILE=1 NOPE_SMALL_MAX_TILES=100000 timeout 300 python tools/ldm_step.py 128 --dtype bf16 2>&1 | grep "^conv" | sort | uniq -c | sort -rn | head -30 >> gpurun_out/r06aj_ldm_small_tile_128.txt
