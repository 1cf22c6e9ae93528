"""GPU micro-benchmark of the HBM-bound 1x1 convs of a 512-hypothesis step: the streaming kernel (kernels_gemm_stream.hip) against the kernels
it replaces, interleaved in ONE process (median of rounds).  Prints time, algorithmic bytes / time, and whether the outputs are equal bits.
    python tools/stream_bench.py [--dtype bf16x3] [--nhyp 512] [--settings "NOPE_CONV_STREAM=0;NOPE_CONV_STREAM=1;NOPE_CONV_STREAM=3"]"""
import argparse
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from nope_amd import hip

SHAPES = [  # name, C1, C2, Cout, H, calls per step
    ("qkv 192->384 @32", 192, 0, 384, 32, 2), ("out 128->192 @32", 128, 0, 192, 32, 2), ("res 384->192 @32 (cat)", 192, 192, 192, 32, 3),
    ("qkv 192->384 @16", 192, 0, 384, 16, 1), ("qkv 384->384 @16", 384, 0, 384, 16, 1), ("out 128->384 @16", 128, 0, 384, 16, 1), ("out 128->192 @16", 128, 0, 192, 16, 1),
    ("res 576->384 @16 (cat)", 384, 192, 384, 16, 2), ("qkv 384->384 @8", 384, 0, 384, 8, 1), ("qkv 768->384 @8", 768, 0, 384, 8, 1), ("out 128->768 @8", 128, 0, 768, 8, 1),
    ("out 128->384 @8", 128, 0, 384, 8, 1), ("res 1152->768 @8 (cat)", 768, 384, 768, 8, 2), ("out 128->1536 @4", 128, 0, 1536, 4, 3), ("res 2304->1536 @4 (cat)", 1536, 768, 1536, 4, 2),
]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--dtype", default="bf16x3")
    ap.add_argument("--nhyp", type=int, default=512)
    ap.add_argument("--reps", type=int, default=10)
    ap.add_argument("--rounds", type=int, default=3)
    ap.add_argument("--settings", default="NOPE_CONV_STREAM=0;NOPE_CONV_STREAM=1;NOPE_CONV_STREAM=3")
    a = ap.parse_args()
    settings = [dict(kv.split("=") for kv in s.split(",") if kv) for s in a.settings.split(";")]
    dt = hip.dtype_code(a.dtype)
    tdt = hip.torch_dtype(dt)
    es = 4 if tdt == torch.float32 else 2
    l = hip.lib()
    tot = [0.0] * len(settings)
    for name, c1, c2, cout, h, calls in SHAPES:
        cin = c1 + c2
        w = torch.randn(cout, cin, 1, 1, device="cuda") / cin ** 0.5
        pw, _, _ = hip.pack_conv_weight(w, dt, hip.CONV_PLAIN)
        s1 = torch.randn(a.nhyp, h, h, c1, device="cuda").to(tdt)
        s2 = torch.randn(a.nhyp, h, h, c2, device="cuda").to(tdt) if c2 else None
        out = torch.empty(a.nhyp, h, h, cout, device="cuda", dtype=tdt)
        bias = torch.randn(cout, device="cuda")
        st = torch.cuda.current_stream().cuda_stream

        def run():
            rc = l.dll.nope_op_conv(dt, s1.data_ptr(), c1, 1, None if s2 is None else s2.data_ptr(), c2, 1, h, h, hip.CONV_PLAIN, 1,
                                    pw.data_ptr(), bias.data_ptr(), None, out.data_ptr(), cout, a.nhyp, 0, 0, 0, st)
            assert rc == 0, rc
        byts = a.nhyp * h * h * (cin + cout) * es
        times = [[] for _ in settings]
        outs = []
        for rnd in range(a.rounds):
            for i, env in enumerate(settings):
                for k, v in env.items():
                    os.environ[k] = v
                hip.lib()                          # (nope_tuning_reload)
                for _ in range(2):
                    run()
                if rnd == 0:
                    torch.cuda.synchronize()
                    outs.append(out.clone())
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(a.reps):
                    run()
                e1.record()
                torch.cuda.synchronize()
                times[i].append(e0.elapsed_time(e1) / a.reps)
                for k in env:
                    os.environ.pop(k)
        line = f"{name:26s} x{calls}"
        for i in range(len(settings)):
            ms = sorted(times[i])[len(times[i]) // 2]
            tot[i] += ms * calls
            line += f" | {ms * 1e3:7.1f} us {byts / ms / 1e9:5.2f} TB/s{'' if i == 0 else (' =' if torch.equal(outs[i], outs[0]) else ' !=')}"
        print(line, flush=True)
    print("per step: " + " | ".join(f"{a.settings.split(';')[i]}: {tot[i]:.3f} ms" for i in range(len(settings))))


if __name__ == "__main__":
    main()
