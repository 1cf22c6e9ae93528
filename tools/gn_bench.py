"""Tuning aid: GroupNorm apply (+ SiLU, + pose embedding, + residual) on the U-Net's activation shapes; run under
`rocprofv3 --kernel-trace --stats` to separate the statistics and the apply kernels (tools/gpu_gn.sh)."""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

from nope_amd import hip


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--act", type=int, default=1)
    ap.add_argument("--emb", type=int, default=0)
    ap.add_argument("--resid", type=int, default=0)
    ap.add_argument("--reps", type=int, default=20)
    ap.add_argument("--dtype", default="f16", choices=["bf16", "f16"])
    a = ap.parse_args()
    g = torch.Generator(device="cuda").manual_seed(3)
    dt = hip.F16 if a.dtype == "f16" else hip.BF16
    td = hip.torch_dtype(dt)
    for (h, c) in ((32, 192), (16, 384), (8, 768), (4, 1536)):      # the U-Net's four levels at 512 hypotheses
        x = torch.randn(512, h, h, c, device="cuda", generator=g).to(td)
        gamma, beta = torch.randn(c, device="cuda", generator=g), torch.randn(c, device="cuda", generator=g)
        emb = torch.randn(512, c, device="cuda", generator=g) if a.emb else None
        res = torch.randn(512, h, h, c, device="cuda", generator=g).to(td) if a.resid else None
        for _ in range(3):
            hip.op_group_norm(dt, x, gamma, beta, 8, bool(a.act), emb, res)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(a.reps):
            hip.op_group_norm(dt, x, gamma, beta, 8, bool(a.act), emb, res)
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1000 / a.reps
        mb = x.numel() * 2 / 1e6
        print(f"gn {h}x{h}x{c} act={a.act} emb={a.emb} resid={a.resid}: {us:7.1f} us per stats+apply  ({mb:.0f} MB tensor; apply moves {mb * (2 + a.resid):.0f} MB, stats {mb:.0f} MB)")


if __name__ == "__main__":
    main()
