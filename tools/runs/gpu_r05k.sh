#!/bin/bash
# round 5, run k: launch-policy sweep for the f32-storage modes' step (the thresholds were tuned on the 16-bit modes): gn_apply bytes per workgroup, XCD map,
# small-tile threshold -- bench.py --dtype f16x2 --skip-extras, 10 steps each, one box
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
OUT=gpurun_out/r05k_f16x2_policy_sweep.txt
export HSA_ENABLE_IPC_MODE_LEGACY=0
: > $OUT
run() { echo -n "$* : " >> $OUT; env "$@" timeout 300 python bench.py --gpus 1 --steps 10 --warmup 3 --skip-extras --dtype f16x2 2>/dev/null | python -c "import sys,json; d=json.load(sys.stdin); print(round(d['ms_per_step'],3), 'ms', round(d['value']), 'hyp/s')" >> $OUT; }
run X=0
run NOPE_GN_BLOCK_KB=32
run NOPE_GN_BLOCK_KB=128
run NOPE_GN_BLOCK_KB=256
run X=0
run NOPE_GN_MIN_GRID=4096
run NOPE_CONV_PERSIST=0
run NOPE_HALO_PERSIST=512
run NOPE_HALO_PERSIST=0
run NOPE_XCD_MAP=1
run X=0
cat $OUT
