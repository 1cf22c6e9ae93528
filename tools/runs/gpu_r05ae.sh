#!/bin/bash
# round 5, run ae: (1) board power and clocks as rocm-smi reports them while the probe's MFMA streams run (constant vs toggling operands are
# separate phases of the probe: ~25 s in all), (2) the LDM variant's 128-hypothesis forward on the final tree
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
OUT=gpurun_out/r05ae_power_while_probing.txt
: > $OUT
rocm-smi --showpower --showclocks 2>/dev/null | grep -iE "power|sclk" | head -4 >> $OUT
( for i in 1 2 3 4 5 6; do tools/probes/overlap_probe --json; done > gpurun_out/r05ae_probe_loop.txt 2>&1 ) &
PID=$!
for i in $(seq 1 14); do
  sleep 0.5
  echo "--- sample $i" >> $OUT
  rocm-smi --showpower --showclocks 2>/dev/null | grep -iE "power|sclk" | head -4 >> $OUT
done
wait $PID
tail -1 gpurun_out/r05ae_probe_loop.txt >> $OUT
cat $OUT | cut -c1-200
timeout 200 python tools/ldm_step.py 2>&1 | grep -v amdgpu | tee gpurun_out/r05ae_ldm_step.txt | tail -6
