"""The small-tile conv kernel (nope_amd/csrc/kernels_gemm_small.hip): the launches of reference-sized banks (26 / 91 / 341 templates,
src/dataloader/shapeNet.py:248-263), of a 64-template shard of a 512-template bank, and of the one-image encoder pass.

CPU: the kernel source runs under tests/hipemu in its adversarial settings -- LDS-DMA landing as LATE as the hardware allows (at the
covering COUNTED vmcnt wait: the ring keeps up to two younger stages in flight) and as EARLY (at issue), waves scheduled one at a
time as far ahead of each other as the workgroup barriers permit -- against torch convolutions and, bit for bit, against the
128 x 192 kernel (same K order, one accumulator per output element).
GPU: the same cases, then the U-Net's real launch shapes at 64 hypotheses, bit-identical to the 128 x 192 kernel without split-K."""
import os

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_small_tile_kernel_under_adversarial_interpreter(emu_jobs):
    """tests/conftest.py: EMU_JOBS.  small_10 / small_32: the op-level cases per compute mode and tile; small_unet16: a whole f16 U-Net schedule (fused
    statistics per 16 / 64 rows, fused PreNorm, NCHW bank); small_unet16split (NOPE_EMU_FULL=1): ... with its 3x3 convs split along K on the
    tap-resident kernel; small_x2: the f16 + MX-fp8 tile on the small-tile kernel (A split in registers), bit-identical to the ping-pong kernels."""
    emu_jobs.collect(["small_10", "small_32", "small_unet16", "small_unet16split", "small_x2"])


@pytest.mark.gpu
def test_small_tile_small_shapes_gpu(gpu):
    from tests import small_emu_case
    assert small_emu_case.run(gpu, "cuda", dts=(1, 0, 3, 2)) < 1.0
    assert small_emu_case.run_x2(gpu, "cuda", tiles=(0, 1, 2, 3)) < 1.0
    errs = {cdt: small_emu_case.run_unet(gpu, "cuda", 64, cdt, n_hyp=5, hw=16, tile=t) for cdt, t in (("f32", 3), ("bf16x3", 1), ("f16", 0), ("bf16", 3))}
    print("U-Net (u_net_dim 64) with every eligible conv on the small-tile kernel: " + ", ".join(f"{k} {v:.2e}" for k, v in errs.items()))
    assert errs["f32"] < 1e-4 and errs["bf16x3"] < 1e-4 and errs["f16"] < 8e-3 and errs["bf16"] < 6e-2


@pytest.mark.gpu
@pytest.mark.parametrize("dt", [1, 0, 3, 2])
def test_small_tile_bit_identical_to_128_tile_kernel(gpu, dt):
    """The U-Net's launch shapes at 64 pose hypotheses (and the encoder's at one image): the small-tile kernel against the
    128 x 192 kernel WITHOUT split-K -- same K order, one accumulator per output element -> equal bits, run to run too."""
    hip = gpu
    g = torch.Generator(device="cuda").manual_seed(6)
    tdt = hip.torch_dtype(dt)
    n = 64 if dt != 0 else 16
    shapes = [  # C1, C2, Cout, H, mode, ksize
        (192, 0, 192, 16, hip.CONV_PLAIN, 3), (384, 0, 384, 8, hip.CONV_PLAIN, 3), (768, 384, 768, 8, hip.CONV_PLAIN, 3),
        (1536, 0, 1536, 4, hip.CONV_PLAIN, 3), (1536, 0, 384, 4, hip.CONV_PLAIN, 1), (128, 0, 192, 32, hip.CONV_PLAIN, 1),
        (768, 0, 384, 8, hip.CONV_UP2P, 3), (192, 0, 384, 16, hip.CONV_DOWN2, 1), (256, 0, 256, 32, hip.CONV_STRIDE2, 3),
    ]
    if dt == 0:
        shapes = shapes[:2] + shapes[4:5] + shapes[6:]
    for c1, c2, cout, h, mode, ks in shapes:
        cin = c1 + c2
        w = torch.randn((cout, cin * 4, 1, 1) if mode == hip.CONV_DOWN2 else (cout, cin, ks, ks), device="cuda", generator=g) / (cin * ks * ks) ** 0.5
        s1 = torch.randn(n, h, h, c1, device="cuda", generator=g).to(tdt)
        s2 = torch.randn(n, h, h, c2, device="cuda", generator=g).to(tdt) if c2 else None
        b = torch.randn(cout, device="cuda", generator=g)
        outs = {}
        for small, tile in (("0", "0"), ("2", "0"), ("2", "1"), ("2", "2")):
            os.environ["NOPE_CONV_SMALL"], os.environ["NOPE_SMALL_TILE"], os.environ["NOPE_CONV_PP"] = small, tile, "0"
            ys = [hip.op_conv(dt, s1, w, b, src2=s2, mode=mode) for _ in range(3)]
            torch.cuda.synchronize()
            assert all(torch.equal(ys[0], y) for y in ys[1:]), ("not reproducible", small, tile, c1, c2, cout, h, mode)
            outs[(small, tile)] = ys[0]
        for k in ("NOPE_CONV_SMALL", "NOPE_SMALL_TILE", "NOPE_CONV_PP"):
            os.environ.pop(k)
        for k, y in outs.items():
            assert torch.equal(outs[("0", "0")], y), ("small-tile kernel != 128-tile kernel", k, c1, c2, cout, h, mode, ks)
        assert bool(torch.isfinite(outs[("2", "0")].float()).all()) and float(outs[("2", "0")].float().abs().max()) > 0.1
