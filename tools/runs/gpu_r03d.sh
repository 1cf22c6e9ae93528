#!/bin/bash
# round 3, run d: quick A/Bs that decide two defaults (tap-resident kernel for the 128-tile 3x3 launches; NCHW through LDS), 1x1 base cost
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
OUT=gpurun_out
for r in 1 2; do
for t in 256 128; do
  NOPE_HALO_MIN_TILES=$t timeout 200 python bench.py --steps 15 --warmup 4 --skip-extras 2>/dev/null | python -c "import json,sys;d=json.loads(sys.stdin.read());print('f16 halo_min_tiles=$t', round(d['value']), round(d['ms_per_step'],3))"
done
for n in 1 0; do
  NOPE_NCHW_STAGED=$n timeout 200 python bench.py --steps 15 --warmup 4 --skip-extras 2>/dev/null | python -c "import json,sys;d=json.loads(sys.stdin.read());print('f16 nchw_staged=$n', round(d['value']), round(d['ms_per_step'],3))"
done
done
NOPE_HALO_MIN_TILES=128 timeout 200 python bench.py --steps 15 --warmup 4 --skip-extras --dtype bf16x3 2>/dev/null | python -c "import json,sys;d=json.loads(sys.stdin.read());print('bf16x3 halo_min_tiles=128', round(d['value']), round(d['ms_per_step'],3))"
timeout 200 python bench.py --steps 15 --warmup 4 --skip-extras --dtype bf16x3 2>/dev/null | python -c "import json,sys;d=json.loads(sys.stdin.read());print('bf16x3 default', round(d['value']), round(d['ms_per_step'],3))"
for v in 112 128; do echo "variant $v"; NOPE_CONV_VARIANT=$v timeout 120 python tools/conv_bench.py --only 11,12 --pp 1 --rounds 2 2>&1 | grep -v "weighted\|amdgpu.ids"; done
timeout 600 python -m pytest tests/test_kernels_parity.py tests/test_gpu_sweeps.py -m gpu -x -q > $OUT/pytest_d.log 2>&1; echo "tests rc=$?"; tail -3 $OUT/pytest_d.log
