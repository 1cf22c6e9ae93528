#!/bin/bash
# round 5, run z: SAME-BOX A/B -- B (in-tree): the per-tap ping-pong kernel's K loop without the tuning tests (TUNE instantiation apart) and with its
# steady state peeled (no run-time test on k); A: previous commit's kernels_gemm_pp.hip
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
OUT=gpurun_out/r05z_pp_loop_ab.txt
export HSA_ENABLE_IPC_MODE_LEGACY=0
: > $OUT
python -c "
import sys; sys.path.insert(0, '.')
from nope_amd import hip
from tests import x2_emu_case, pp_emu_case
print('x2 cases on the GPU (worst error / bound):', x2_emu_case.run(hip, 'cuda'))
print('pp cases:', pp_emu_case.run(hip, 'cuda'))" 2>&1 | grep -v amdgpu | tee -a $OUT
for round in 1 2 3; do
 for v in A B; do
  if [ $v = A ]; then export NOPE_HIP_LIB=$PWD/build/ab/libnope_hip_a.so; else unset NOPE_HIP_LIB; fi
  for dt in f16x2 bf16x3 f16; do
   timeout 300 python bench.py --gpus 1 --steps 10 --warmup 3 --skip-extras --dtype $dt 2>/dev/null | python -c "import sys,json; d=json.load(sys.stdin); print('round $round variant $v step', d['dtype'], round(d['ms_per_step'],3), 'ms', round(d['value']), 'hyp/s')" >> $OUT
  done
 done
done
for v in A B; do
  if [ $v = A ]; then export NOPE_HIP_LIB=$PWD/build/ab/libnope_hip_a.so; else unset NOPE_HIP_LIB; fi
  timeout 300 python bench.py --gpus 1 --steps 5 --warmup 2 --extras roofline --dtype f16x2 2>/dev/null | python -c "
import sys,json; d=json.load(sys.stdin)
for c in d['roofline']['classes']:
    if c['kernel'] == 'conv_gemm_pp_kernel': print('variant $v', c['kernel'], 'mode', c['mode'], c['taps'], c['Cin'], c['Cout'], c['H'], 'x', c['launches'], round(c['avg_ms']*1e3,1), 'us', round(c['frac'],3))" >> $OUT
done
cat $OUT
