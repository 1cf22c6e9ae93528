"""Warm-up + a few scoring launches on a resident bank larger than the Infinity Cache (default 32 queries x 2048 bf16 templates =
1.07 GB) and nothing else -- the command of the PMC passes that measure sim_reg_kernel's FETCH_SIZE / WRITE_SIZE
(tools/gpu_pmc.sh sim tools/sim_step.py).      python tools/sim_step.py [--templates 2048] [--dtype bf16]"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from nope_amd import hip

ap = argparse.ArgumentParser()
ap.add_argument("--templates", type=int, default=2048)
ap.add_argument("--dtype", default="bf16")
a = ap.parse_args()
dt = {"bf16": torch.bfloat16, "f16": torch.float16, "f32": torch.float32}[a.dtype]
B, C, h = 32, 8, 32
bank = torch.randn(B, a.templates, C, h, h, device="cuda", dtype=torch.float16).to(dt)
q = torch.randn(B, C, h, h, device="cuda")
out = torch.empty(B, a.templates, device="cuda")
for _ in range(6):
    hip.similarity(q, bank, out=out)
torch.cuda.synchronize()
print("ok", tuple(out.shape), "algorithmic bytes per launch", B * a.templates * (C * h * h * bank.element_size() + 4))
