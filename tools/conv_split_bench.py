"""Few-tile / long-K 3x3 convs (the 4 x 4 and 8 x 8 levels of the U-Net at 64 pose hypotheses) with the split-K scratch the runtimes hand
the launcher: time per launch with WARM weights (the same tensor every launch: it stays in the 256 MB Infinity Cache) and COLD ones
(rotating through more weight tensors than the cache holds) -- tells a latency / DRAM-efficiency problem of the weight stream from a
compute problem.   python tools/conv_split_bench.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from nope_amd import hip


def bench(c1, cout, h, n, nw, reps=40, dt=hip.F16):
    g = torch.Generator(device="cuda").manual_seed(1)
    tdt = hip.torch_dtype(dt)
    x = torch.randn(n, h, h, c1, device="cuda", generator=g).to(tdt)
    ws = [torch.randn(cout, c1, 3, 3, device="cuda", generator=g) / (3 * c1 ** 0.5) for _ in range(nw)]
    b = torch.randn(cout, device="cuda", generator=g)
    l = hip.lib()
    packed = [hip.pack_conv_weight(w, dt)[0] for w in ws]
    out = torch.empty(n, h, h, cout, dtype=tdt, device="cuda")
    sk = int(l.dll.nope_op_conv_splitk_bytes(dt, c1, 0, 1, h, h, 0, 9, cout, n))
    scratch = torch.empty(max(sk, 16), dtype=torch.uint8, device="cuda")

    def run(i):
        pw = packed[i % nw]
        l.check(l.dll.nope_op_conv_ws(dt, x.data_ptr(), c1, 1, None, 0, 1, h, h, 0, 9, pw.data_ptr(), b.data_ptr(), None, out.data_ptr(), cout, n, 0, 0, 0,
                                      scratch.data_ptr() if sk else None, sk, torch.cuda.current_stream().cuda_stream), "conv")
    for i in range(nw + 3):
        run(i)
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    ev[0].record()
    for i in range(reps):
        run(i)
    ev[1].record()
    torch.cuda.synchronize()
    return ev[0].elapsed_time(ev[1]) / reps * 1e3, sk


for (c1, cout, h, n) in ((1536, 1536, 4, 64), (768, 768, 8, 64), (1536, 1536, 4, 26), (2304, 1536, 4, 64), (768, 768, 4, 64)):
    for env in ({}, {"NOPE_HALO_SPLIT": "0"}):
        for k, v in env.items():
            os.environ[k] = v
        warm, sk = bench(c1, cout, h, n, 1)
        nw = max(2, int(400e6 / (cout * c1 * 9 * 2)))
        cold, _ = bench(c1, cout, h, n, nw)
        print(f"{c1}->{cout} @{h}x{h} n={n} {env or 'default'}: warm {warm:7.1f} us  cold ({nw} weight tensors) {cold:7.1f} us  (conv + reduce; scratch {sk >> 20} MiB)", flush=True)
        for k in env:
            os.environ.pop(k)
