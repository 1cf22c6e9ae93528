#!/bin/bash
# round 4, run t: same-box A/B: launches of 128..255 tap-resident tiles split in two (341 templates: 176 tiles at the 4 x 4 level)
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
timeout 900 python tools/small_bank_sweep.py --dtype f16 --banks 256,341,400 --steps 30 --settings ";NOPE_HALO_SPLIT2=1;;NOPE_HALO_SPLIT2=1" > gpurun_out/halo_split2_ab.txt 2>gpurun_out/sweep.err; cat gpurun_out/halo_split2_ab.txt
