"""Compute modes against the f32 parity mode at a BASELINE configuration (default configs[2]: 32 queries x 512 templates, 256x256):
per mode the largest score deviation relative to max |score|, queries with the same top-1 / the same top-5, and the step time.
    python tools/mode_parity.py [--batch 32] [--templates 512] [--size 256] [--modes bf16,f16,bf16x3]"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from nope_amd.harness import build_model, synthetic_batch

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=32)
ap.add_argument("--templates", type=int, default=512)
ap.add_argument("--size", type=int, default=256)
ap.add_argument("--seed", type=int, default=77)
ap.add_argument("--modes", default="bf16,f16,bf16x3")
a = ap.parse_args()
b = synthetic_batch(a.batch, a.templates, a.size, seed=a.seed, device="cuda")


def run(mode):
    m = build_model(compute_dtype=mode, bank_dtype=mode if mode in ("bf16", "f16") else "f32", device="cuda")
    m.generate_and_retrieve(b["query"][:1], b["reference"][:1], b["all_relativeR"][:1])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    sim, idx, _ = m.generate_and_retrieve(b["query"], b["reference"], b["all_relativeR"])
    torch.cuda.synchronize()
    return sim, idx, time.perf_counter() - t0


sim32, idx32, t32 = run("f32")
scale = float(sim32.abs().max())
gap = sim32.topk(2, dim=1).values
print(json.dumps({"mode": "f32", "seconds": t32, "hyp_per_s": a.batch * a.templates / t32, "min_top1_gap_rel": float((gap[:, 0] - gap[:, 1]).min()) / scale}))
for mode in a.modes.split(","):
    sim, idx, t = run(mode)
    print(json.dumps({"mode": mode, "score_rel_err": float((sim - sim32).abs().max()) / scale,
                      "per_query_rel_err_max": float(((sim - sim32).abs().max(dim=1).values / sim32.abs().max(dim=1).values).max()),
                      "top1_equal": int((idx[:, 0] == idx32[:, 0]).sum()), "top5_equal": int((idx == idx32).all(dim=1).sum()), "queries": a.batch,
                      "seconds": t, "hyp_per_s": a.batch * a.templates / t}))
