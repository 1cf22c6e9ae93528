"""Template-encoder latency on the device: the C-ABI path (f32 / bf16 compute) next to the same nn.Module tree
executed by PyTorch-ROCm (MIOpen), for B in {1, 2, 8} images of 256x256.   python tools/encoder_bench.py"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from nope_amd.encoder import FeatureExtractor
from nope_amd.weights import synth_init_


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / n * 1e3


def main():
    encs = {}
    for cdt in ("f32", "bf16"):
        e = FeatureExtractor(8, 0.2, False, compute_dtype=cdt)
        synth_init_(e, 2022, prefix="encoder.")
        encs[cdt] = e.cuda()
    for B in (1, 2, 8):
        img = torch.rand(B, 3, 256, 256, device="cuda") * 2 - 1
        ref = encs["f32"].projector(encs["f32"].backbone(img))
        row = [f"B={B}"]
        for cdt, e in encs.items():
            out = e.encode_image(img)
            err = float((out - ref).abs().max() / ref.abs().max())
            row.append(f"hip {cdt}: {timeit(lambda: e.encode_image(img)):.3f} ms (vs torch {err:.1e})")
        with torch.no_grad():
            row.append(f"torch/MIOpen f32: {timeit(lambda: encs['f32'].projector(encs['f32'].backbone(img))):.3f} ms")
        print("  ".join(row))


if __name__ == "__main__":
    main()
