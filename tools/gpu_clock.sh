#!/bin/bash
# Sustained shader clock per kernel during a bench run (GRBM_GUI_ACTIVE / duration).
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
D=/tmp/clk
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE -d $D -o p -- python $OLDPWD/bench.py --steps 3 --warmup 1 --skip-extras > $OLDPWD/gpurun_out/clock.log 2>&1 )
python tools/rocpd_clock.py $(find $D -name "*.db" | head -1) > gpurun_out/clock_by_kernel.csv 2>&1
cat gpurun_out/clock_by_kernel.csv | cut -c1-150
rocm-smi --showclocks --showpower 2>/dev/null | grep -iE "sclk|power|mclk" | head -5
