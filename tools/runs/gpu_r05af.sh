#!/bin/bash
# round 5, run af: rocm-smi's package power and sclk, sampled every ~0.4 s while the probe's full (text) run walks through its phases --
# constant-operand and toggling-operand MFMA streams last ~30-160 ms each at the end of the run (the --json form is too short to catch)
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
OUT=gpurun_out/r05af_power_while_probing.txt
: > $OUT
( for i in 1 2 3; do tools/probes/overlap_probe; done > gpurun_out/r05af_probe.txt 2>&1 ) &
PID=$!
for i in $(seq 1 60); do
  echo "sample $i $(rocm-smi --showpower --showclocks 2>/dev/null | grep -iE 'Package Power|sclk' | sed -e 's/.*: //' | tr '\n' ' ')" >> $OUT
  kill -0 $PID 2>/dev/null || break
done
wait $PID
cat $OUT | cut -c1-160
