"""GPU micro-benchmark of the implicit-GEMM conv on the U-Net's dominant shapes (bf16 or f32).
    python tools/conv_bench.py [--dtype bf16] [--nhyp 512] [--pp 0,1,3,5]
Prints per-shape time and TFLOP/s (algorithmic flops 2*M*Cout*taps*Cin) for each kernel policy in --pp
(NOPE_CONV_PP: 0 = 128 x 192 kernel everywhere, 1 = default, 3 / 5 = the 4x4 level on the ping-pong kernel in standard /
position-major row order), interleaved in ONE process over several rounds (median)."""
import argparse
import ctypes as C
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from nope_amd import hip

SHAPES = [  # name, C1, C2, Cout, Hs, mode, ksize, calls/forward
    ("L0 3x3 192->192 @32", 192, 0, 192, 32, hip.CONV_PLAIN, 3, 10),
    ("L0 3x3 384->192 @32 (cat)", 192, 192, 192, 32, hip.CONV_PLAIN, 3, 3),
    ("L1 3x3 384->384 @16", 384, 0, 384, 16, hip.CONV_PLAIN, 3, 6),
    ("L1 3x3 576->384 @16 (cat)", 384, 192, 384, 16, hip.CONV_PLAIN, 3, 2),
    ("L2 3x3 768->768 @8", 768, 0, 768, 8, hip.CONV_PLAIN, 3, 6),
    ("L2 3x3 1152->768 @8 (cat)", 768, 384, 768, 8, hip.CONV_PLAIN, 3, 2),
    ("L3 3x3 1536->1536 @4", 1536, 0, 1536, 4, hip.CONV_PLAIN, 3, 11),
    ("L3 3x3 2304->1536 @4 (cat)", 1536, 768, 1536, 4, hip.CONV_PLAIN, 3, 2),
    ("UP 3x3 384->192 @16->32", 384, 0, 192, 16, hip.CONV_UP2, 3, 1),
    ("UP 3x3 1536->768 @4->8", 1536, 0, 768, 4, hip.CONV_UP2, 3, 1),
    ("DOWN 1x1 768->192... (192*4->192) @32->16", 192, 0, 192, 32, hip.CONV_DOWN2, 1, 1),
    ("1x1 192->384 qkv @32", 192, 0, 384, 32, hip.CONV_PLAIN, 1, 2),
    ("1x1 128->192 out @32", 128, 0, 192, 32, hip.CONV_PLAIN, 1, 2),
    ("1x1 2304->1536 res @4", 1536, 768, 1536, 4, hip.CONV_PLAIN, 1, 2),
]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--dtype", default="bf16")
    ap.add_argument("--nhyp", type=int, default=512)
    ap.add_argument("--reps", type=int, default=10)
    ap.add_argument("--only", default="", help="comma-separated shape indices")
    ap.add_argument("--pp", default="0,1", help="comma-separated NOPE_CONV_PP settings to compare")
    ap.add_argument("--rounds", type=int, default=3)
    a = ap.parse_args()
    pps = a.pp.split(",")
    dt = hip.dtype_code(a.dtype)
    tdt = hip.torch_dtype(dt)
    l = hip.lib()
    tot_ms = {pp: 0.0 for pp in pps}
    tot_fl = 0.0
    for name, c1, c2, cout, hs, mode, ks, calls in SHAPES:
        if a.only and str(SHAPES.index((name, c1, c2, cout, hs, mode, ks, calls))) not in a.only.split(','):
            continue
        cin = c1 + c2
        ntaps = 4 if mode == hip.CONV_DOWN2 else ks * ks
        wshape = (cout, cin * 4, 1, 1) if mode == hip.CONV_DOWN2 else (cout, cin, ks, ks)
        w = torch.randn(wshape, device="cuda") / (cin * ntaps) ** 0.5
        pw, _, _ = hip.pack_conv_weight(w, dt, mode)
        s1 = torch.randn(a.nhyp, hs, hs, c1, device="cuda").to(tdt)
        s2 = torch.randn(a.nhyp, hs, hs, c2, device="cuda").to(tdt) if c2 else None
        ho = 2 * hs if mode == hip.CONV_UP2 else (hs // 2 if mode == hip.CONV_DOWN2 else hs)
        out = torch.empty(a.nhyp, ho, ho, cout, device="cuda", dtype=tdt)
        bias = torch.randn(cout, device="cuda")
        st = torch.cuda.current_stream().cuda_stream

        def run():
            rc = l.dll.nope_op_conv(dt, s1.data_ptr(), c1, 1, None if s2 is None else s2.data_ptr(), c2, 1, hs, hs, mode, ntaps,
                                    pw.data_ptr(), bias.data_ptr(), None, out.data_ptr(), cout, a.nhyp, 0, 0, 0, st)
            assert rc == 0, rc
        fl = 2.0 * a.nhyp * ho * ho * cout * ntaps * cin
        times = {pp: [] for pp in pps}
        for rnd in range(a.rounds):
            for pp in pps:
                os.environ["NOPE_CONV_PP"] = pp
                hip.lib()                          # (notices the changed variable: nope_tuning_reload -- the library caches its switches)
                for _ in range(2):
                    run()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(a.reps):
                    run()
                e1.record()
                torch.cuda.synchronize()
                times[pp].append(e0.elapsed_time(e1) / a.reps)
        tot_fl += fl * calls
        line = f"{name:45s} x{calls:2d}"
        for pp in pps:
            ms = sorted(times[pp])[len(times[pp]) // 2]
            tot_ms[pp] += ms * calls
            line += f" | pp={pp}: {ms*1e3:8.1f} us {fl/ms/1e9:7.1f} TF"
        print(line, flush=True)
    for pp in pps:
        print(f"weighted pp={pp}: {tot_ms[pp]:.2f} ms per forward-equivalent, {tot_fl/tot_ms[pp]/1e9:.1f} TF (algorithmic flops)")


if __name__ == "__main__":
    main()
