#!/bin/bash
# round 4, run v: launch-policy thresholds once more on the final kernels (same box): split-K chunk floor, small-tile ceiling, two-K-group ceiling
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
timeout 1200 python tools/small_bank_sweep.py --dtype f16 --banks 26,64,128,256,341 --steps 30 --settings ";NOPE_HALO_SPLIT_MIN_CHUNKS=6;NOPE_HALO_SPLIT_MIN_CHUNKS=9;NOPE_SMALL_MAX_TILES=160;NOPE_SMALL_MAX_TILES=480;NOPE_SMALL_KG2_MAX=128;NOPE_SMALL_KG2_MAX=512;NOPE_ENC_GRAPH=1;" > gpurun_out/policy_sweep.txt 2>gpurun_out/sweep.err; cat gpurun_out/policy_sweep.txt
