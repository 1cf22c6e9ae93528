#!/bin/bash
# round 5, run g: operand rewrite of the tap-resident kernel (f16x2, bf16x3) with its LDS read opening the LOAD phase and the DMA issue / address
# arithmetic underneath it (compare r05f: f16x2 647.7 / 1101.6 / 540.3 / 790.1 / 518.1 / 770.6 / 500.7 / 744.3 us; r05c bf16x3 930 / 1687 / 844 / 1243)
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
OUT=gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
for dt in f16x2 bf16x3; do
timeout 300 python tools/conv_bench.py --dtype $dt --only 0,1,2,3,4,5,6,7 --pp 3 --rounds 3 > $OUT/r05g_conv_bench_$dt.txt 2>&1; cat $OUT/r05g_conv_bench_$dt.txt | grep -v amdgpu.ids
done
timeout 300 python -c "
import sys; sys.path.insert(0, '.')
from nope_amd import hip
from tests import x2_emu_case, pp_emu_case
print('x2 op-level cases on the GPU: worst error / tolerance =', x2_emu_case.run(hip, 'cuda'))
print('pp cases bf16x3:', pp_emu_case.run(hip, 'cuda', dts=(3,)))
" 2>&1 | grep -v amdgpu.ids
timeout 600 python bench.py --gpus 1 --steps 10 --warmup 3 --skip-extras > $OUT/r05g_bench_step.json 2>/dev/null; cut -c1-260 $OUT/r05g_bench_step.json
