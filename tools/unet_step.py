"""One warm-up + one measured U-Net batch (512 pose hypotheses, 32x32 latent, bf16) and nothing else -- the command
the PMC passes of tools/gpu_pmc.sh run, so that every conv_gemm_dma_kernel dispatch counted belongs to the U-Net
(the encoder launches the same kernel on tiny problems).   python tools/unet_step.py [--templates 512] [--dtype bf16]"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from nope_amd.harness import build_model

ap = argparse.ArgumentParser()
ap.add_argument("--templates", type=int, default=512)
ap.add_argument("--dtype", default="bf16")
a = ap.parse_args()
m = build_model(compute_dtype=a.dtype, bank_dtype=a.dtype if a.dtype in ("bf16", "f16") else "f32", device="cuda")
g = torch.Generator().manual_seed(0)
feat = torch.randn(1, 8, 32, 32, generator=g).cuda()
poses = torch.randn(1, a.templates, 6, generator=g).cuda()
for _ in range(2):
    bank = m.generate_templates_from_feat(feat, poses)
torch.cuda.synchronize()
print("ok", tuple(bank.shape))
