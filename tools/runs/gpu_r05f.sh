#!/bin/bash
# round 5, run f: f16x2 operand rewrite with v_cvt_scalef32_pk_fp8_f32 (pre-scale folded into the conversion, word_sel packing): probe fact 6, the
# tap-resident 3x3 launches of the 512-hypothesis step (compare r05c: 692.7 / 1206.8 / 582.7 / 848.1 us), the bench step
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
OUT=gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
tools/probes/mx_probe > $OUT/r05f_mx_probe.txt 2>&1; tail -1 $OUT/r05f_mx_probe.txt
timeout 300 python tools/conv_bench.py --dtype f16x2 --only 0,1,2,3,4,5,6,7 --pp 3 --rounds 3 > $OUT/r05f_conv_bench_f16x2.txt 2>&1; cat $OUT/r05f_conv_bench_f16x2.txt | grep -v amdgpu.ids
timeout 300 python -c "
import sys; sys.path.insert(0, '.')
from nope_amd import hip
from tests import x2_emu_case
print('x2 op-level cases on the GPU: worst error / tolerance =', x2_emu_case.run(hip, 'cuda'))
" 2>&1 | grep -v amdgpu.ids
timeout 600 python bench.py --gpus 1 --steps 10 --warmup 3 --skip-extras > $OUT/r05f_bench_step.json 2>/dev/null; cut -c1-260 $OUT/r05f_bench_step.json
