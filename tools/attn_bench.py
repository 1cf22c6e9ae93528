"""Tuning aid: the LinearAttention core (`linattn_kernel`) on the U-Net's four levels at 512 hypotheses, f16 storage.
   python tools/attn_bench.py [--reps 20]      (A/B another build with NOPE_HIP_LIB=...)"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from nope_amd import hip

ap = argparse.ArgumentParser()
ap.add_argument("--reps", type=int, default=20)
ap.add_argument("--dtype", default="f16", choices=["bf16", "f16", "f32"])
a = ap.parse_args()
dt = {"f16": hip.F16, "bf16": hip.BF16, "f32": hip.F32}[a.dtype]
g = torch.Generator(device="cuda").manual_seed(5)
for h in (32, 16, 8, 4):
    qkv = torch.randn(512, h, h, 384, device="cuda", generator=g).to(hip.torch_dtype(dt))
    for _ in range(3):
        hip.op_linear_attention(dt, qkv)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(a.reps):
        hip.op_linear_attention(dt, qkv)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1000 / a.reps
    mb = qkv.numel() * qkv.element_size() * (4 / 3) / 1e6        # q, k, v in + out (a third of qkv)
    print(f"linattn {h}x{h} x 512 {a.dtype}: {us:7.1f} us  {mb:.0f} MB algorithmic = {mb / us:.2f} TB/s")
