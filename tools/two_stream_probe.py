"""Experiment: does running two half-batches of hypotheses on two HIP streams (conv of one half under the
streaming kernels of the other) beat one full batch?  python tools/two_stream_probe.py"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from nope_amd.harness import build_model

N = 512
m1 = build_model(compute_dtype="bf16", bank_dtype="bf16", device="cuda")
m2 = build_model(compute_dtype="bf16", bank_dtype="bf16", device="cuda")
g = torch.Generator().manual_seed(0)
feat = torch.randn(1, 8, 32, 32, generator=g).cuda()
poses = torch.randn(1, N, 6, generator=g).cuda()
bank = torch.empty(1, N, 8, 32, 32, dtype=torch.bfloat16, device="cuda")
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()


def single():
    m1.u_net.forward_hypotheses(feat, poses, out=bank, out_dtype="bf16")


def split(parts):
    cur = torch.cuda.current_stream()
    per = N // parts
    for i in range(parts):
        st, mm = (s1, m1) if i % 2 == 0 else (s2, m2)
        st.wait_stream(cur)
        with torch.cuda.stream(st):
            mm.u_net.forward_hypotheses(feat, poses[:, i * per:(i + 1) * per].contiguous(), out=bank[:, i * per:(i + 1) * per], out_dtype="bf16")
    cur.wait_stream(s1); cur.wait_stream(s2)


def timeit(fn, n=10):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / n * 1e3


print("single stream, 512 hyps   : %.2f ms" % timeit(single))
ref = bank.clone()
print("two streams, 2 x 256      : %.2f ms" % timeit(lambda: split(2)))
print("equal:", bool(torch.equal(ref, bank)))
print("two streams, 4 x 128      : %.2f ms" % timeit(lambda: split(4)))
print("single stream again       : %.2f ms" % timeit(single))
