#!/bin/bash
# round 6, run an: the 16-bit MFMA attention kernel of the LDM variant instantiated for f16 storage as well (v_mfma_f32_32x32x16_f16; the f16 mode ran the VALU kernel):
# LDM / token tests, per-launch time at the shipped shapes (VALU f16 via NOPE_LDM_ATTN=0 vs MFMA), the f16 128-hypothesis forward with both.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out; mkdir -p $OUT
export HSA_ENABLE_IPC_MODE_LEGACY=0 TMPDIR=/tmp
timeout 900 python -m pytest tests -q -m gpu -k "ldm or token or geglu" > $OUT/r06an_pytest_ldm.log 2>&1; echo "pytest rc=$?"; tail -2 $OUT/r06an_pytest_ldm.log
timeout 600 python - > $OUT/r06an_attn_f16.txt 2>&1 <<'PY'
import os, sys, torch
sys.path.insert(0, ".")
from nope_amd import hip
g = torch.Generator(device="cuda").manual_seed(5)
for (n, N, C) in ((128, 1024, 256), (128, 256, 512), (128, 64, 1024)):
    qkv = torch.randn(n, N, 3 * C, device="cuda", generator=g)
    ref = hip.op_token_attention(0, qkv)
    for dt, name, env in ((2, "f16 VALU", "0"), (2, "f16 MFMA", "1"), (1, "bf16 MFMA", "1")):
        os.environ["NOPE_LDM_ATTN"] = env
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
        err = float((y.float() - ref).abs().max() / ref.abs().max())
        print(f"attention {n} x {N} tokens x {C} ch, {name:10s}: {t*1e3:8.1f} us  vs f32 VALU {err:.2e}", flush=True)
    os.environ.pop("NOPE_LDM_ATTN")
PY
grep -v amdgpu.ids $OUT/r06an_attn_f16.txt
for a in 0 1 0 1; do echo -n "NOPE_LDM_ATTN=$a: "; NOPE_LDM_ATTN=$a timeout 300 python tools/ldm_step.py 128 --dtype f16 2>&1 | grep LDM; done | tee $OUT/r06an_ldm_f16_ab.txt
