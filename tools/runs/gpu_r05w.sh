#!/bin/bash
# round 5, run w: the f16 + MX-fp8 tile on the per-tap ping-pong kernel (1x1, space-to-depth, phase convs; operands split in registers):
# same-box, same-library A/B through NOPE_X2_PP (0 = tap-resident launches only, as before), per-class times, score error, GPU form of the emulator cases

set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
OUT=gpurun_out/r05w_x2_per_tap_ab.txt
export HSA_ENABLE_IPC_MODE_LEGACY=0
: > $OUT
python -c "
import sys; sys.path.insert(0, '.')
from nope_amd import hip
from tests import x2_emu_case
print('x2 cases on the GPU (worst error / bound):', x2_emu_case.run(hip, 'cuda'))
print('x2 U-Net vs oracle:', x2_emu_case.run_unet(hip, 'cuda'))" 2>&1 | grep -v amdgpu | tee -a $OUT
for round in 1 2 3; do
 for v in 0 1; do
   NOPE_X2_PP=$v timeout 300 python bench.py --gpus 1 --steps 10 --warmup 3 --skip-extras --dtype f16x2 2>/dev/null | python -c "import sys,json; d=json.load(sys.stdin); print('round $round NOPE_X2_PP=$v step', d['dtype'], round(d['ms_per_step'],3), 'ms', round(d['value']), 'hyp/s')" >> $OUT
 done
done
for v in 0 1; do
  NOPE_X2_PP=$v timeout 300 python bench.py --gpus 1 --steps 5 --warmup 2 --extras roofline --dtype f16x2 2>/dev/null | python -c "
import sys,json; d=json.load(sys.stdin)
for c in d['roofline']['classes']:
    if c['kernel'] == 'conv_gemm_pp_kernel': print('NOPE_X2_PP=$v', c['kernel'], 'mode', c['mode'], c['taps'], c['Cin'], c['Cout'], c['H'], 'x', c['launches'], round(c['avg_ms']*1e3,1), 'us', round(c['frac'],3))" >> $OUT
done
cat $OUT
python - <<'PY' 2>&1 | grep -v amdgpu | tee -a $OUT
import torch, json, sys, os
sys.path.insert(0, '.')
from nope_amd.harness import build_model, synthetic_batch
b = synthetic_batch(1, 512, 256, seed=2022, device="cuda")
m32 = build_model(compute_dtype="f32", device="cuda")
s32, i32, _ = m32.generate_and_retrieve(b["query"], b["reference"], b["all_relativeR"])
for v in ("0", "1"):
    os.environ["NOPE_X2_PP"] = v
    m = build_model(compute_dtype="f16x2", device="cuda")
    s, i, _ = m.generate_and_retrieve(b["query"], b["reference"], b["all_relativeR"])
    print("NOPE_X2_PP=" + v, "f16x2 score_rel_err vs f32:", float((s - s32).abs().max() / s32.abs().max()), "top5 equal", bool(torch.equal(i, i32)))
PY
