"""GPU micro-benchmark of the fused GroupNorm apply (+SiLU +emb / +resid) and linear attention at the U-Net's shapes:
time and achieved HBM GB/s (algorithmic bytes: read x [+resid], write y).   python tools/gn_bench.py [--nhyp 512]"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from nope_amd import hip

ap = argparse.ArgumentParser()
ap.add_argument("--nhyp", type=int, default=512)
a = ap.parse_args()
dt = hip.BF16


def timeit(fn, n=10):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / n


for (C, hw) in ((192, 32), (384, 16), (768, 8), (1536, 4)):
    x = torch.randn(a.nhyp, hw, hw, C, device="cuda").to(torch.bfloat16)
    rs = torch.randn_like(x.float()).to(torch.bfloat16)
    ga, be, emb = torch.randn(C, device="cuda"), torch.randn(C, device="cuda"), torch.randn(a.nhyp, C, device="cuda")
    for name, kw, nbuf in (("GN8+SiLU+emb", dict(act_silu=True, emb=emb), 2), ("GN8+SiLU+resid", dict(act_silu=True, resid=rs), 3),
                           ("GN1+resid", dict(act_silu=False, resid=rs), 3)):
        G = 1 if name.startswith("GN1") else 8
        t = timeit(lambda: hip.op_group_norm(dt, x, ga, be, G, **kw))
        gb = x.numel() * 2 * (nbuf + 1) / 1e9      # + the statistics pass reads x once more
        print(f"C={C:4d} {hw:2d}x{hw:<2d} {name:15s} {t * 1e6:7.1f} us  {gb / t:7.0f} GB/s (stats + apply)")
    qkv = torch.randn(a.nhyp, hw, hw, 384, device="cuda").to(torch.bfloat16)
    t = timeit(lambda: hip.op_linear_attention(dt, qkv))
    gb = (qkv.numel() * 2 * (4 / 3) + a.nhyp * hw * hw * 128 * 2) / 1e9   # q once, k twice, v once + out
    print(f"       {hw:2d}x{hw:<2d} linear attention {t * 1e6:7.1f} us  {gb / t:7.0f} GB/s")
