"""Experiment: hypotheses per U-Net launch (512 / 256 / 128) on one stream -- do smaller activation tensors (closer to
the 256 MB Infinity Cache) make the streaming kernels cheaper than the extra launches cost?  python tools/chunk_probe.py"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from nope_amd.harness import build_model

m = build_model(compute_dtype="bf16", bank_dtype="bf16", device="cuda")
g = torch.Generator().manual_seed(0)
feat = torch.randn(1, 8, 32, 32, generator=g).cuda()
poses = torch.randn(1, 512, 6, generator=g).cuda()
for chunk in (512, 256, 128, 512):
    m.max_hyp = chunk
    for _ in range(2):
        m.generate_templates_from_feat(feat, poses)
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(10):
        m.generate_templates_from_feat(feat, poses)
    torch.cuda.synchronize()
    print(f"hypotheses per launch {chunk:4d}: {(time.perf_counter() - t) * 100:.2f} ms per 512")
