#!/bin/bash
# round 5, run c: first GPU contact of the f16x2 tile -- the emulator's op-level cases on the device (against the restated arithmetic and the
# f32 convolution), then the tap-resident 3x3 launches of the 512-hypothesis step per mode (f16x2 vs bf16x3 vs f16)
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
OUT=gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
tools/probes/mx_probe > $OUT/r05c_mx_probe.txt 2>&1; tail -1 $OUT/r05c_mx_probe.txt
timeout 300 python -c "
import sys; sys.path.insert(0, '.')
from nope_amd import hip
from tests import x2_emu_case
print('x2 op-level cases on the GPU: worst error / tolerance =', x2_emu_case.run(hip, 'cuda'))
print('x2 U-Net (dim 64, 5 hypotheses, 16 x 16) vs oracle:', x2_emu_case.run_unet(hip, 'cuda'))
" > $OUT/r05c_x2_cases.txt 2>&1; echo "cases rc=$?"; tail -5 $OUT/r05c_x2_cases.txt
for dt in f16x2 bf16x3 f16; do
  timeout 300 python tools/conv_bench.py --dtype $dt --only 0,1,2,3,4,5,6,7 --pp 1 --rounds 3 > $OUT/r05c_conv_bench_$dt.txt 2>&1; echo "bench $dt rc=$?"; cat $OUT/r05c_conv_bench_$dt.txt
done
