#!/bin/bash
# round 5, run v: SAME-BOX A/B -- B (in-tree): linattn on f32 storage with the hardware exp in the split-precision modes + (second block) the MFMA phase without raised priority (NOPE_PP_VARIANT=2, same library); A: previous commit

set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
OUT=gpurun_out/r05v_linattn_fast_exp_ab.txt
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
    if False: print('variant $v', c['kernel'], 'mode', c['mode'], c['taps'], c['Cin'], c['Cout'], c['H'], 'x', c['launches'], round(c['avg_ms']*1e3,1), 'us')" >> $OUT
done
cat $OUT
python -c "
import sys; sys.path.insert(0, '.')
from nope_amd import hip
from tests import pp_emu_case
print('x:', pp_emu_case.run(hip, 'cuda', dts=(3,)))" 2>&1 | grep -v amdgpu
python - <<'PY' 2>&1 | grep -v amdgpu
import torch, json, sys
sys.path.insert(0, '.')
from nope_amd.harness import build_model, synthetic_batch
b = synthetic_batch(1, 512, 256, seed=2022, device="cuda")
m32 = build_model(compute_dtype="f32", device="cuda")
s32, i32, _ = m32.generate_and_retrieve(b["query"], b["reference"], b["all_relativeR"])
for mode in ("f16x2", "bf16x3"):
    m = build_model(compute_dtype=mode, device="cuda")
    s, i, _ = m.generate_and_retrieve(b["query"], b["reference"], b["all_relativeR"])
    print(mode, "score_rel_err vs f32 with the fast SiLU:", float((s - s32).abs().max() / s32.abs().max()), "top5 equal", bool(torch.equal(i, i32)))
PY
