#!/bin/bash
# round 4, run u: same-box A/B of splitting 128-tile tap-resident launches in two (NOPE_HALO_SPLIT_MAX_TILES 127 = before), banks where it applies
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
timeout 900 python tools/small_bank_sweep.py --dtype f16 --banks 128,192,256,512 --steps 30 --settings ";NOPE_HALO_SPLIT_MAX_TILES=127;;NOPE_HALO_SPLIT_MAX_TILES=127" > gpurun_out/halo_split128_ab.txt 2>gpurun_out/sweep.err; cat gpurun_out/halo_split128_ab.txt
timeout 600 python -m pytest tests/test_conv_small.py tests/test_gpu_configs.py -m gpu -x -q 2>&1 | tail -3
