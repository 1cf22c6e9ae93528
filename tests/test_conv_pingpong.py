"""The 256 x 192 eight-wave ping-pong conv kernel (nope_amd/csrc/kernels_gemm_pp.hip).

CPU: the kernel source runs under tests/hipemu in its two adversarial settings -- LDS-DMA landing as LATE as the
hardware allows (at the covering vmcnt wait) and as EARLY (at issue), waves scheduled one at a time as far ahead of
each other as the workgroup barriers permit -- so a misplaced wait, a missing barrier or a ring that is one stage
too short gives wrong numbers here, before a GPU-minute is spent (the mutations are described in DESIGN.md).
GPU: same small shapes, then the U-Net's real launch shapes, where the kernel must agree BIT FOR BIT with the
128 x 192 kernel (same K order, same MFMA shape, same accumulation chain) on every one of several repetitions."""
import os

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_pingpong_kernel_under_adversarial_interpreter(emu_jobs):
    """tests/conftest.py: EMU_JOBS.  pp_light_*: the op-level cases of the per-tap and tap-resident kernels in the two adversarial settings; x2_light / x2_unet:
    the tap-resident kernel's NOPE_F16X2 tile (in-LDS operand rewrite + MX-scaled fp8 MFMA) with late DMA + shuffled waves, and a whole U-Net in the
    f16x2 mode (second weight pack per 3x3 layer, block scale from the packed tail, the other launches as bf16x3) with the tap-resident kernel forced
    onto every eligible launch, against the oracle; up2p_light: the up-sampling phase convs on the tap-resident kernel (four K steps per channel
    chunk) in bf16x3 and f16x2, bit for bit the per-tap kernel; pp_unet16 (NOPE_EMU_FULL=1; on the GPU otherwise: test_pingpong_small_shapes_gpu): a
    whole 16-bit U-Net schedule.  (The remaining op-level cases -- tile walk, split-K, residual epilogue, widest map -- run on the GPU,
    test_f16x2_tile_gpu, and by hand: python tests/x2_emu_case.py.)"""
    emu_jobs.collect(["pp_light_13", "pp_light_02", "x2_light", "x2_unet", "up2p_light", "pp_unet16"])


@pytest.mark.gpu
def test_f16x2_tile_gpu(gpu):
    """The same cases on the device: against the restated arithmetic (2e-6: operand rounding, pre-scales, block scale, slot layout) and the f32
    convolution (3e-5); then the shapes of the 512-hypothesis step -- reproducible, tile walk == one tile per workgroup, split-K == fixed-order sum
    within rounding -- against the bf16x3 launch of the same operands."""
    from tests import x2_emu_case
    hip = gpu
    assert x2_emu_case.run(hip, "cuda") < 1.0
    e = x2_emu_case.run_unet(hip, "cuda")
    print(f"U-Net (u_net_dim 64, 5 hypotheses, 16 x 16) in the f16x2 mode, tap-resident kernel forced: {e:.2e} vs the oracle")
    assert e < 1e-4
    # ---- off the benchmark's activation range (u_net_dim 64): the handle re-centres the per-layer shifts and repeats the call
    res = x2_emu_case.run_unet_range(hip, "cuda")
    for S, e, events, x2_on in res:
        print(f"f16x2 U-Net, reference embedding x {S:g}: {e:.2e} vs the oracle; repeated forwards {[(ev['code'], ev['layers_out_of_range'], float('%.3g' % ev['max_abs'])) for ev in events]}; two-pass tile {'on' if x2_on else 'OFF (bf16x3)'}")
        assert e < 1e-4 and x2_on, (S, e, events)
    by = {S: events for S, e, events, x2_on in res}
    assert res[-1][0] > 1e6 and any(ev["attempt"] == -1 for ev in res[-1][2]), "last entry: the default 'poison' mode (NaN output, verdict at the next forward)"
    assert by[1.0] == [] and by[1e2] == [], "inside the initial window nothing is repeated"
    assert any(ev["code"] == hip.ERR_RANGE for ev in by[1e4]), "|a| ~ 1e4 leaves the t = 0 window: shifts moved, call repeated"
    assert any(ev["code"] == hip.ERR_RANGE for ev in by[3e5]), "beyond 65504 (where a plain f16 operand saturates) the shifts still repair it"
    g = torch.Generator(device="cuda").manual_seed(6)
    for c1, c2, cout, h, n in ((192, 0, 192, 32, 512), (192, 192, 192, 32, 256), (384, 192, 384, 16, 512), (768, 0, 768, 8, 512), (1536, 0, 1536, 4, 512)):
        cin = c1 + c2
        w = torch.randn(cout, cin, 3, 3, device="cuda", generator=g) / (cin * 9) ** 0.5
        s1 = torch.randn(n, h, h, c1, device="cuda", generator=g)
        s2 = torch.randn(n, h, h, c2, device="cuda", generator=g) if c2 else None
        b = torch.randn(cout, device="cuda", generator=g)
        ys = [hip.op_conv(hip.F16X2, s1, w, b, src2=s2) for _ in range(3)]
        torch.cuda.synchronize()
        assert all(torch.equal(ys[0], y) for y in ys[1:]), ("not reproducible", c1, c2, cout, h)
        y3 = hip.op_conv(hip.BF16X3, s1, w, b, src2=s2)
        e = float((ys[0] - y3).abs().max() / y3.abs().max())
        print(f"f16x2 vs bf16x3, {cin} -> {cout} @ {h}x{h} x {n}: {e:.2e}")
        assert e < 3e-5
        os.environ["NOPE_HALO_PERSIST"] = "0"
        y1 = hip.op_conv(hip.F16X2, s1, w, b, src2=s2)
        os.environ.pop("NOPE_HALO_PERSIST")
        assert torch.equal(ys[0], y1), ("tile walk differs", c1, c2, cout, h)


@pytest.mark.gpu
def test_up2p_tap_resident_gpu(gpu):
    """The up-sampling phase convs on the tap-resident kernel (f32-storage modes): the op-level cases on the device, then the three up-sampling
    launches of a 512-hypothesis step -- reproducible and the bits of the per-tap ping-pong kernel (NOPE_UP2P_HALO=0), in bf16x3 and f16x2."""
    from tests import up2p_emu_case
    hip = gpu
    assert up2p_emu_case.run(hip, "cuda") < 1.0
    g = torch.Generator(device="cuda").manual_seed(6)
    for dt in (hip.F16X2, hip.BF16X3):
        for cin, cout, h in ((384, 192, 16), (768, 384, 8), (1536, 768, 4)):
            w = torch.randn(cout, cin, 3, 3, device="cuda", generator=g) / (cin * 9) ** 0.5
            x = torch.randn(512, h, h, cin, device="cuda", generator=g)
            b = torch.randn(cout, device="cuda", generator=g)
            os.environ["NOPE_UP2P_HALO"] = "2"         # (2: bf16x3 layers too; the default takes the tap-resident form for f16x2 layers only)
            try:
                ys = [hip.op_conv(dt, x, w, b, mode=hip.CONV_UP2P) for _ in range(3)]
                os.environ["NOPE_UP2P_HALO"] = "0"
                y_tap = hip.op_conv(dt, x, w, b, mode=hip.CONV_UP2P)
            finally:
                os.environ.pop("NOPE_UP2P_HALO")
            torch.cuda.synchronize()
            assert all(torch.equal(ys[0], y) for y in ys[1:]), ("not reproducible", dt, cin, cout, h)
            assert torch.equal(ys[0], y_tap), ("tap-resident phase convs != per-tap kernel", dt, cin, cout, h)
            assert bool(torch.isfinite(ys[0]).all()) and float(ys[0].abs().max()) > 0.1


@pytest.mark.gpu
def test_pingpong_small_shapes_gpu(gpu):
    from tests import pp_emu_case
    assert pp_emu_case.run(gpu, "cuda", dts=(1, 0, 3, 2)) < 1.0
    errs = {cdt: pp_emu_case.run_unet(gpu, "cuda", 64, cdt, n_hyp=5, hw=16) for cdt in ("f32", "bf16x3", "f16", "bf16")}
    print("U-Net (u_net_dim 64) with every eligible conv on the ping-pong kernel: " + ", ".join(f"{k} {v:.2e}" for k, v in errs.items()))
    assert errs["f32"] < 1e-4 and errs["bf16x3"] < 1e-4 and errs["f16"] < 8e-3 and errs["bf16"] < 6e-2


@pytest.mark.gpu
@pytest.mark.parametrize("dt", [1, 0, 3, 2])
def test_pingpong_bit_identical_to_128_tile_kernel(gpu, dt):
    hip = gpu
    g = torch.Generator(device="cuda").manual_seed(5)
    tdt = hip.torch_dtype(dt)
    n = {1: 512, 2: 512, 3: 256, 0: 128}[dt]          # f32: 16x fewer MFMA flops per second -- keep the test short
    shapes = [  # C1, C2, Cout, H, mode, ksize
        (192, 0, 192, 32, hip.CONV_PLAIN, 3), (192, 192, 192, 32, hip.CONV_PLAIN, 3), (384, 192, 384, 16, hip.CONV_PLAIN, 3),
        (768, 0, 768, 8, hip.CONV_PLAIN, 3), (192, 0, 384, 32, hip.CONV_PLAIN, 1), (128, 0, 192, 32, hip.CONV_PLAIN, 1),
        (384, 0, 192, 16, hip.CONV_UP2P, 3), (192, 0, 384, 32, hip.CONV_DOWN2, 1), (1536, 0, 1536, 4, hip.CONV_PLAIN, 3),
    ]
    if dt == 0:
        shapes = shapes[:1] + shapes[3:4] + shapes[6:8]
    for c1, c2, cout, h, mode, ks in shapes:
        cin = c1 + c2
        w = torch.randn((cout, cin * 4, 1, 1) if mode == hip.CONV_DOWN2 else (cout, cin, ks, ks), device="cuda", generator=g) / (cin * ks * ks) ** 0.5
        s1 = torch.randn(n, h, h, c1, device="cuda", generator=g).to(tdt)
        s2 = torch.randn(n, h, h, c2, device="cuda", generator=g).to(tdt) if c2 else None
        b = torch.randn(cout, device="cuda", generator=g)
        outs = {}
        # 0: 128 x 192 kernel; 3: ping-pong schedule everywhere (3x3 convs on the tap-resident kernel); 19: 3x3 convs per tap
        for pp in ("0", "3", "19"):
            os.environ["NOPE_CONV_PP"] = pp
            ys = [hip.op_conv(dt, s1, w, b, src2=s2, mode=mode) for _ in range(3)]
            torch.cuda.synchronize()
            assert all(torch.equal(ys[0], y) for y in ys[1:]), ("not reproducible", pp, c1, c2, cout, h, mode)
            outs[pp] = ys[0]
        os.environ.pop("NOPE_CONV_PP")
        # (the 128-tile kernel runs the 4x4 shape position-major: the padding taps it skips only ever add exact zeros)
        assert torch.equal(outs["0"], outs["3"]), ("ping-pong / tap-resident != 128-tile kernel", c1, c2, cout, h, mode, ks)
        assert torch.equal(outs["0"], outs["19"]), ("per-tap ping-pong != 128-tile kernel", c1, c2, cout, h, mode, ks)
        assert bool(torch.isfinite(outs["3"].float()).all()) and float(outs["3"].float().abs().max()) > 0.1
        if mode == hip.CONV_PLAIN:
            # the epilogue without a residual (bf16: packed two-rows-per-dword panel) against the one with (f32 panel), and
            # workgroups walking several tiles against one tile per workgroup
            y0 = hip.op_conv(dt, s1, w, b, src2=s2, mode=mode, resid=torch.zeros_like(outs["3"]))
            os.environ["NOPE_HALO_PERSIST"] = "0"
            y1 = hip.op_conv(dt, s1, w, b, src2=s2, mode=mode)
            os.environ.pop("NOPE_HALO_PERSIST")
            assert torch.equal(outs["3"], y0), ("epilogue panels differ", c1, c2, cout, h, ks)
            assert torch.equal(outs["3"], y1), ("tile walk differs", c1, c2, cout, h, ks)
