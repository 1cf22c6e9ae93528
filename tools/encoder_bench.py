"""Template-encoder latency on the device through the C ABI, per compute mode, for B in {1, 2, 8} images of 256x256 (and the f16 output
against the f32 mode's).   python tools/encoder_bench.py"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from nope_amd.encoder import FeatureExtractor
from nope_amd.weights import synth_init_


def timeit(fn, n=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / n * 1e3


def main():
    encs = {}
    for cdt in ("f32", "bf16x3", "f16", "bf16"):
        e = FeatureExtractor(8, 0.2, False, compute_dtype=cdt)
        synth_init_(e, 2022, prefix="encoder.")
        encs[cdt] = e.cuda()
    for B in (1, 2, 8):
        img = torch.rand(B, 3, 256, 256, device="cuda") * 2 - 1
        ref = encs["f32"].encode_image(img)
        row = [f"B={B}"]
        for cdt, e in encs.items():
            out = e.encode_image(img)
            err = float((out - ref).abs().max() / ref.abs().max())
            row.append(f"{cdt}: {timeit(lambda: e.encode_image(img)):.3f} ms (vs f32 {err:.1e})")
        print("  ".join(row), flush=True)


if __name__ == "__main__":
    main()
