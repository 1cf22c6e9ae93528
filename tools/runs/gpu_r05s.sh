#!/bin/bash
# round 5, run s: SAME-BOX A/B -- B (in-tree): the per-tap ping-pong kernel's bf16x3 fragment split inside its COMPUTE phase; A (build/ab/libnope_hip_a.so): in the
# LOAD phase (previous commit).  Step of both f32-storage modes, three rounds, + the per-class table of the B build
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
OUT=gpurun_out/r05s_pp_prep_in_compute_ab.txt
export HSA_ENABLE_IPC_MODE_LEGACY=0
: > $OUT
for round in 1 2 3; do
 for v in A B; do
  if [ $v = A ]; then export NOPE_HIP_LIB=$PWD/build/ab/libnope_hip_a.so; else unset NOPE_HIP_LIB; fi
  for dt in f16x2 bf16x3; do
   timeout 300 python bench.py --gpus 1 --steps 10 --warmup 3 --skip-extras --dtype $dt 2>/dev/null | python -c "import sys,json; d=json.load(sys.stdin); print('round $round variant $v step', d['dtype'], round(d['ms_per_step'],3), 'ms', round(d['value']), 'hyp/s')" >> $OUT
  done
 done
done
for v in A B; do
  if [ $v = A ]; then export NOPE_HIP_LIB=$PWD/build/ab/libnope_hip_a.so; else unset NOPE_HIP_LIB; fi
  timeout 300 python bench.py --gpus 1 --steps 5 --warmup 2 --extras roofline --dtype f16x2 2>/dev/null | python -c "
import sys,json; d=json.load(sys.stdin)
for c in d['roofline']['classes']:
    if c['kernel'] != 'conv3x3_halo_kernel' and c['avg_ms'] * c['launches'] > 0.15: print('variant $v', c['kernel'], 'mode', c['mode'], c['taps'], c['Cin'], c['Cout'], c['H'], 'x', c['launches'], round(c['avg_ms']*1e3,1), 'us')" >> $OUT
done
cat $OUT
python -c "
import sys; sys.path.insert(0, '.')
from nope_amd import hip
from tests import pp_emu_case
print('bf16x3 pp cases on the GPU:', pp_emu_case.run(hip, 'cuda', dts=(3,)))" 2>&1 | grep -v amdgpu
