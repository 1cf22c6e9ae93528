"""Scoring-kernel micro-benchmark: GB/s of nope_similarity on a resident bank larger than the Infinity Cache.
    NOPE_SIM_VARIANT=<bits> python tools/sim_bench.py      (1 = non-temporal bank loads, 2 = one residency round of long workgroups, 8 = 8-byte loads for 16-bit banks; default 1)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from nope_amd import hip

for dtype, N in ((torch.bfloat16, 512), (torch.bfloat16, 2048), (torch.float32, 512), (torch.float16, 1024), (torch.float16, 4096)):
    B, C, h = 32, 8, 32
    bank = torch.randn(B, N, C, h, h, device="cuda", dtype=torch.float16).to(dtype)
    q = torch.randn(B, C, h, h, device="cuda")
    out = torch.empty(B, N, device="cuda")
    for _ in range(25):
        hip.similarity(q, bank, out=out)
    reps = 30
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(reps + 1)]
    ev[0].record()
    for i in range(reps):
        hip.similarity(q, bank, out=out)
        ev[i + 1].record()
    torch.cuda.synchronize()
    ms = sorted(ev[i].elapsed_time(ev[i + 1]) for i in range(reps))
    med, mn = ms[len(ms) // 2], ms[0]
    byts = B * N * (C * h * h * bank.element_size() + 4)
    print(f"variant {os.environ.get('NOPE_SIM_VARIANT', 'default')} {str(dtype):16s} N={N:5d}: median {byts / med / 1e6:7.1f} GB/s ({byts / med / 8e9:.3f} of 8 TB/s), best {byts / mn / 1e6:7.1f}")
