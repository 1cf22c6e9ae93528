#!/bin/bash
# round 6, run ah: the MFMA attention kernels with the key mask in the ragged block only (and, f32 storage, the score scale folded into Q before its split):
# per-launch times at the three shipped shapes against r06aa_attn_shapes.txt (same script), LDM token tests.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out; mkdir -p $OUT
export HSA_ENABLE_IPC_MODE_LEGACY=0 TMPDIR=/tmp
timeout 900 python -m pytest tests -q -m gpu -k "token or ldm" > $OUT/r06ah_pytest_ldm.log 2>&1; echo "pytest rc=$?"; tail -2 $OUT/r06ah_pytest_ldm.log
timeout 600 python - > $OUT/r06ah_attn_shapes.txt 2>&1 <<'PY'
import os, sys, torch
sys.path.insert(0, ".")
from nope_amd import hip
g = torch.Generator(device="cuda").manual_seed(5)
for (n, N, C) in ((128, 1024, 256), (128, 256, 512), (128, 64, 1024), (128, 1000, 256)):
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
grep -v amdgpu.ids $OUT/r06ah_attn_shapes.txt
for dt in bf16 f16x2; do timeout 300 python tools/ldm_step.py 128 --dtype $dt 2>&1 | grep LDM; done | tee $OUT/r06ah_ldm_step.txt
