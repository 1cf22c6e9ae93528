"""What bounds the per-tap ping-pong kernel (conv_gemm_pp_kernel)?  Times its large launches of the default step per compute mode with the
kernel's tuning ablations (NOPE_PP_VARIANT: 16 = no DMA stream, 32 = no MFMA, 64 = no epilogue, 128 = launch only, 1024 = no B pieces, 512 = A pieces
on the first tap of a channel chunk only, 1 = fragment reads before the DMA pieces -- wrong results, right instruction streams; sums combine), and
two launches of the tap-resident kernel beside them.      python tools/pp_stream_probe.py [--nhyp 512]"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from nope_amd import hip

SHAPES = [  # name, C1, C2, Cout, Hs, mode, ksize
    ("3x3 1536->1536 @4 (tap-resident)", 1536, 0, 1536, 4, hip.CONV_PLAIN, 3),
    ("3x3 192->192 @32 (tap-resident)", 192, 0, 192, 32, hip.CONV_PLAIN, 3),
    ("phase convs 768->384 @8->16", 768, 0, 384, 8, hip.CONV_UP2P, 3),
    ("phase convs 384->192 @16->32", 384, 0, 192, 16, hip.CONV_UP2P, 3),
    ("1x1 2304->1536 @4 (cat)", 1536, 768, 1536, 4, hip.CONV_PLAIN, 1),
    ("1x1 384->192 @32 (cat)", 192, 192, 192, 32, hip.CONV_PLAIN, 1),
]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--nhyp", type=int, default=512)
    ap.add_argument("--reps", type=int, default=10)
    a = ap.parse_args()
    for name, c1, c2, cout, hs, mode, ks in SHAPES:
        for dtn in ("bf16x3", "f16x2", "f16"):
            dt = hip.dtype_code(dtn)
            tdt = hip.torch_dtype(dt)
            w = torch.randn(cout, c1 + c2, ks, ks, device="cuda") / ((c1 + c2) * ks * ks) ** 0.5
            s1 = torch.randn(a.nhyp, hs, hs, c1, device="cuda").to(tdt)
            s2 = torch.randn(a.nhyp, hs, hs, c2, device="cuda").to(tdt) if c2 else None
            pw, cin, ntaps = hip.pack_conv_weight(w, dt, mode)
            ho = 2 * hs if mode == hip.CONV_UP2P else hs
            out = torch.empty(a.nhyp, ho, ho, cout, device="cuda", dtype=tdt)
            st = torch.cuda.current_stream().cuda_stream

            def run(l):
                rc = l.dll.nope_op_conv(dt, s1.data_ptr(), c1, 1, None if s2 is None else s2.data_ptr(), c2, 1, hs, hs, mode, ntaps,
                                        pw.data_ptr(), None, None, out.data_ptr(), cout, a.nhyp, 0, 0, 0, st)
                assert rc == 0, rc
            line = f"{name:34s} {dtn:7s}"
            for var in ("0", "16", "32", "48", "112", "128") if ks == 3 else ("0", "16", "32", "1024", "512", "48", "49", "112"):
                os.environ["NOPE_PP_VARIANT"] = var
                l = hip.lib()                      # (notices the changed variable: nope_tuning_reload)
                for _ in range(2):
                    run(l)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(a.reps):
                    run(l)
                e1.record()
                torch.cuda.synchronize()
                line += f"  v{var}: {e0.elapsed_time(e1) / a.reps * 1e3:7.1f} us"
            os.environ.pop("NOPE_PP_VARIANT")
            print(line, flush=True)


if __name__ == "__main__":
    main()
