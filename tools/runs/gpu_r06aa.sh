#!/bin/bash
# round 6, run aa: softmax self-attention of the LDM variant on three bf16 MFMA passes for the f32-storage compute modes (token_attn_mfma_x3_kernel):
# the LDM tests on the device, per-launch times at the three shipped attention shapes (VALU f32 / x3 / bf16 MFMA), and the 128-hypothesis step
# per mode with the old kernel (NOPE_LDM_ATTN=0) and the new one, same box.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out; mkdir -p $OUT
export HSA_ENABLE_IPC_MODE_LEGACY=0 TMPDIR=/tmp
timeout 900 python -m pytest tests -q -m gpu -k "ldm or token" > $OUT/r06aa_pytest_ldm.log 2>&1; echo "pytest rc=$?"; tail -3 $OUT/r06aa_pytest_ldm.log
timeout 600 python - > $OUT/r06aa_attn_shapes.txt 2>&1 <<'PY'
import os, sys, torch
sys.path.insert(0, ".")
from nope_amd import hip
g = torch.Generator(device="cuda").manual_seed(5)
for (n, N, C) in ((128, 1024, 256), (128, 256, 512), (128, 64, 1024)):
    qkv = torch.randn(n, N, 3 * C, device="cuda", generator=g)
    ref = None
    for dt, name in ((0, "f32 VALU"), (3, "x3 MFMA"), (1, "bf16 MFMA")):
        x = qkv.to(hip.torch_dtype(dt))
        y = hip.op_token_attention(dt, x)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ts = []
        for _ in range(5):
            e0.record()
            for _ in range(3):
                y = hip.op_token_attention(dt, x)
            e1.record(); torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) / 3)
        t = sorted(ts)[2]
        if ref is None: ref = y.float()
        err = float((y.float() - ref).abs().max() / ref.abs().max())
        fl = 4.0 * n * (C // 32) * N * N * 32
        print(f"attention {n} x {N} tokens x {C} ch, {name:10s}: {t*1e3:8.1f} us  {fl/t/1e9:7.1f} TFLOP/s (one pass)  vs f32 VALU {err:.2e}", flush=True)
PY
cat $OUT/r06aa_attn_shapes.txt | grep -v amdgpu.ids
for dt in bf16 bf16x3 f16x2; do
  for a in 0 1; do
    echo -n "NOPE_LDM_ATTN=$a: "; NOPE_LDM_ATTN=$a timeout 300 python tools/ldm_step.py 128 --dtype $dt 2>&1 | grep LDM
  done
done | tee $OUT/r06aa_ldm_step_ab.txt
