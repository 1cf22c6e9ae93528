"""Runs the small-tile conv kernel (kernels_gemm_small.hip) on small shapes under tests/hipemu and checks it against torch
convolutions and, bit for bit, against the 128 x 192 kernel.  Executed as a subprocess by tests/test_conv_small.py with different
interpreter settings (HIPEMU_DMA=late: LDS-DMA lands at the covering wait -- the counted vmcnt waits of the ring are what this
exercises; HIPEMU_SHUFFLE: wave scheduling order)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "hipemu"))
import torch
import torch.nn.functional as F

import build_emu
from nope_amd import hip
from oracle import nope_ref as R
from tests.util import rel


def run(hip, dev, dts=(1, 0), tiles=(0, 1, 2, 3), light=False):
    g = torch.Generator().manual_seed(177)
    rn = lambda *s: torch.randn(*s, generator=g)
    d = lambda x: x.to(dev)
    worst = 0.0
    for dt in dts:
        q = lambda x: x.to(hip.torch_dtype(dt)).float()
        tol = {0: 2e-5, 1: 4e-2, 2: 5e-3, 3: 3e-5}[dt]      # f32, bf16, f16, bf16x3
        C = 32 if dt in (0, 3) else 64           # one 128-byte K step per tap and source

        def both(fn, what, ref, t=tol, bit_equal=True):
            """fn() on the small-tile kernel (every tile variant) and on the 128 x 192 kernel: all against `ref`, and equal bits."""
            nonlocal worst
            os.environ["NOPE_CONV_SMALL"] = "0"
            os.environ["NOPE_CONV_PP"] = "0"
            y_big = fn()
            os.environ.pop("NOPE_CONV_PP")
            for tile in tiles:
                os.environ["NOPE_CONV_SMALL"] = "2"
                os.environ["NOPE_SMALL_TILE"] = str(tile)
                y = fn()
                os.environ.pop("NOPE_SMALL_TILE")
                yy = y.cpu().float() if tuple(y.shape) == tuple(ref.shape) else hip.to_nchw(y, dt).cpu()      # (NCHW f32 outputs come back as they are)
                e = rel(yy, ref)
                worst = max(worst, e / t)
                assert e < t, (what, dt, tile, e)
                if bit_equal and tile != 3:
                    assert torch.equal(y, y_big), (what, dt, tile, "small-tile kernel differs from the 128 x 192 kernel")
                if tile == 3:        # two K groups: (even steps) + (odd steps) -- another association of the same sum, but always the same one
                    os.environ["NOPE_SMALL_TILE"] = "3"
                    assert torch.equal(y, fn()), (what, dt, "the two-group tile is not reproducible")
                    os.environ.pop("NOPE_SMALL_TILE")
            os.environ.pop("NOPE_CONV_SMALL")

        # 3x3 over a virtual concat (broadcast first source), ragged M (270 rows), Cout = 200 (4 tiles of 64, the last one ragged), bias
        x1, x2 = rn(1, C, 10, 9), rn(3, C, 10, 9)
        w, b = rn(200, 2 * C, 3, 3) / 30, rn(200)
        both(lambda: hip.op_conv(dt, hip.to_nhwc(d(x1), dt), d(w), d(b), src2=hip.to_nhwc(d(x2), dt), rep1=3, rep2=1, n_hyp=3),
             "3x3 concat", F.conv2d(torch.cat((q(x1).expand(3, -1, -1, -1), q(x2)), 1), q(w), b, padding=1))
        # 4x4 maps, many samples per tile, ReLU
        xs, ws_, bs = rn(40, C, 4, 4), rn(24, C, 3, 3) / (3 * C ** 0.5), rn(24)
        both(lambda: hip.op_conv(dt, hip.to_nhwc(d(xs), dt), d(ws_), d(bs), act_relu=True), "3x3 4x4 relu", F.relu(F.conv2d(q(xs), q(ws_), bs, padding=1)))
        # 1x1 with one, two, three and five K steps (the ring wraps), residual
        for nkc in (1, 2, 3, 5):
            x3, w3, r3 = rn(2, nkc * C, 12, 11), rn(72, nkc * C, 1, 1) / (nkc * C) ** 0.5, rn(2, 72, 12, 11)
            both(lambda: hip.op_conv(dt, hip.to_nhwc(d(x3), dt), d(w3), None, resid=hip.to_nhwc(d(r3), dt)), f"1x1 nk={nkc}", F.conv2d(q(x3), q(w3)) + q(r3))
        # NCHW f32 output with a channel count that is no whole vector (the U-Net's last conv: 8 channels)
        x6, w6, b6 = rn(3, C, 8, 8), rn(8, C, 1, 1) / C ** 0.5, rn(8)
        both(lambda: hip.op_conv(dt, hip.to_nhwc(d(x6), dt), d(w6), d(b6), out_nchw=True, out_dtype=0), "1x1 nchw", F.conv2d(q(x6), q(w6), b6))
        # ... with a residual (NHWC, element type of the launch) and ReLU behind it: the NCHW epilogue adds it like every other one
        r6 = rn(3, 8, 8, 8)
        both(lambda: hip.op_conv(dt, hip.to_nhwc(d(x6), dt), d(w6), d(b6), resid=hip.to_nhwc(d(r6), dt), out_nchw=True, out_dtype=0, act_relu=True), "1x1 nchw + residual",
             F.relu(F.conv2d(q(x6), q(w6), b6) + q(r6)))
        # split-K on the tap-resident kernel (few 256-row tiles, long K): 20 channel chunks over 16 splits (one or two chunks each: both A
        # stage parities), 4 chunks over 4 splits; deterministic, and equal to the unsplit result up to the association of the partials
        os.environ["NOPE_HALO_SPLIT_MIN_CHUNKS"] = "2"
        if dt in (0, 3):
            os.environ["NOPE_CONV_PP"] = "3"        # (the f32 modes stay off the ping-pong kernels unless asked)
        for nch in (20, 4):
            xk, wk, bk_ = rn(40, nch * C, 4, 4), rn(24, nch * C, 3, 3) / (3 * (nch * C) ** 0.5), rn(24)
            rk = rn(40, 24, 4, 4)
            f = lambda: hip.op_conv(dt, hip.to_nhwc(d(xk), dt), d(wk), d(bk_), resid=hip.to_nhwc(d(rk), dt), act_relu=True, split_k=True)
            y = f()
            assert torch.equal(y, f()), "split-K launch is not reproducible"
            e = rel(hip.to_nchw(y, dt).cpu(), F.relu(F.conv2d(q(xk), q(wk), bk_, padding=1) + q(rk)))
            worst = max(worst, e / tol)
            assert e < tol, ("halo split-K", dt, nch, e)
        os.environ.pop("NOPE_HALO_SPLIT_MIN_CHUNKS")
        os.environ.pop("NOPE_CONV_PP", None)
        if light:
            continue
        # nearest-x2 + 3x3 as four 2x2 phase convs; space-to-depth + 1x1; stride-2 3x3 and 1x1 (encoder)
        wu, bu = rn(40, C, 3, 3) / 24, rn(40)
        both(lambda: hip.op_conv(dt, hip.to_nhwc(d(x2), dt), d(wu), d(bu), mode=hip.CONV_UP2P), "up2p",
             R.hard_upsample(q(x2), {"1.weight": wu, "1.bias": bu}, ""), {0: tol, 1: 6e-2, 2: 8e-3, 3: tol}[dt])
        x4 = rn(5, C, 12, 10)
        wd, bd = rn(72, 4 * C, 1, 1) / 16, rn(72)
        both(lambda: hip.op_conv(dt, hip.to_nhwc(d(x4), dt), d(wd), d(bd), mode=hip.CONV_DOWN2), "down2", R.hard_downsample(q(x4), {"1.weight": q(wd), "1.bias": bd}, ""))
        ws2, bs2 = rn(48, C, 3, 3) / (3 * C ** 0.5), rn(48)
        both(lambda: hip.op_conv(dt, hip.to_nhwc(d(x4), dt), d(ws2), d(bs2), mode=hip.CONV_STRIDE2, act_relu=True), "stride2 3x3",
             F.relu(F.conv2d(q(x4), q(ws2), bs2, stride=2, padding=1)))
        ws1 = rn(48, C, 1, 1) / C ** 0.5
        both(lambda: hip.op_conv(dt, hip.to_nhwc(d(x4), dt), d(ws1), None, mode=hip.CONV_STRIDE2), "stride2 1x1", F.conv2d(q(x4), q(ws1), stride=2))
    return worst


def run_x2(hip, dev, tiles=(0, 1, 3)):
    """NOPE_F16X2 on the small-tile kernel (round 6: reference-sized banks and 8-way shards keep the two-pass tile): the layer's second
    pack as B, A rows staged as raw f32 and split in registers.  Against the restated arithmetic of the tile (2e-6), the f32 convolution
    (3e-5) and -- same K order, same three terms per 32-channel step into one accumulator -- bit for bit against the ping-pong kernels."""
    from tests.x2_emu_case import X2, up2p_conv, up2p_phase_weights, x2_reference
    g = torch.Generator().manual_seed(178)
    rn = lambda *s: torch.randn(*s, generator=g)
    d = lambda x: x.to(dev)
    C = 32
    worst = 0.0
    os.environ["NOPE_X2_SMALL"] = "1"          # (off by default: measured slower than bf16x3 on this kernel, kernels_gemm.hip: plan_takes_x2)

    def both(fn, what, want, want32):
        nonlocal worst
        os.environ["NOPE_CONV_PP"], os.environ["NOPE_CONV_SMALL"] = "13", "0"
        y_pp = fn()
        os.environ.pop("NOPE_CONV_PP")
        for tile in tiles:
            os.environ["NOPE_CONV_SMALL"], os.environ["NOPE_SMALL_TILE"] = "2", str(tile)
            y = fn()
            yy = y.cpu().float() if tuple(y.shape) == tuple(want.shape) else hip.to_nchw(y, 0).cpu()
            e, e32 = rel(yy, want), rel(yy, want32)
            worst = max(worst, e / 2e-6, e32 / 3e-5)
            assert e < 2e-6 and e32 < 3e-5, (what, tile, e, e32)
            if tile != 3:
                assert torch.equal(y, y_pp), (what, tile, "small-tile f16x2 differs from the ping-pong kernel")
            else:
                assert torch.equal(y, fn()), (what, "the two-group tile is not reproducible")
            os.environ.pop("NOPE_SMALL_TILE")
        os.environ.pop("NOPE_CONV_SMALL")

    x1, x2 = rn(3, C, 10, 9), rn(3, C, 10, 9)
    w, b = rn(200, 2 * C, 3, 3) / 30, rn(200)
    xc = torch.cat((x1, x2), 1)
    both(lambda: hip.op_conv(X2, hip.to_nhwc(d(x1), 0), d(w), d(b), src2=hip.to_nhwc(d(x2), 0)), "3x3 concat", x2_reference(xc, w, b), F.conv2d(xc, w, b, padding=1))
    for nkc in (1, 3, 5):
        x3, w3, r3 = rn(2, nkc * C, 12, 11), rn(72, nkc * C, 1, 1) / (nkc * C) ** 0.5, rn(2, 72, 12, 11)
        c1 = lambda a, ww: F.conv2d(a, ww)
        both(lambda: hip.op_conv(X2, hip.to_nhwc(d(x3), 0), d(w3), None, resid=hip.to_nhwc(d(r3), 0)), f"1x1 nk={nkc} + residual",
             x2_reference(x3, w3, None, c1) + r3, F.conv2d(x3, w3) + r3)
    wu, bu = rn(40, C, 3, 3) / 24, rn(40)
    both(lambda: hip.op_conv(X2, hip.to_nhwc(d(x2), 0), d(wu), d(bu), mode=hip.CONV_UP2P), "up2p",
         x2_reference(x2, up2p_phase_weights(wu), bu, up2p_conv), F.conv2d(F.interpolate(x2, scale_factor=2, mode="nearest"), wu, bu, padding=1))
    x4, wd, bd = rn(5, C, 12, 10), rn(72, 4 * C, 1, 1) / 16, rn(72)
    wd2 = wd.view(72, C, 2, 2)
    cs2 = lambda a, ww: F.conv2d(a, ww, None, stride=2)
    both(lambda: hip.op_conv(X2, hip.to_nhwc(d(x4), 0), d(wd), d(bd), mode=hip.CONV_DOWN2), "down2", x2_reference(x4, wd2, bd, cs2), F.conv2d(x4, wd2, bd, stride=2))
    # the range shift: activations x 2^12 under t = 12 = exactly 2^12 x the t = 0 result (|a| in [0.5, 1.5): the f16 part stays normal)
    xr, wr = torch.sign(rn(3, C, 10, 9)) * (0.5 + torch.rand(3, C, 10, 9, generator=g)), rn(40, C, 3, 3) / (3 * C ** 0.5)
    os.environ["NOPE_CONV_SMALL"], os.environ["NOPE_SMALL_TILE"] = "2", "0"
    y0 = hip.op_conv(X2, hip.to_nhwc(d(xr), 0), d(wr), None)
    y12 = hip.op_conv(X2, hip.to_nhwc(d(xr * 4096.0), 0), d(wr), None, x2_shift=12)
    os.environ.pop("NOPE_CONV_SMALL"); os.environ.pop("NOPE_SMALL_TILE")
    os.environ.pop("NOPE_X2_SMALL")
    assert torch.equal(y12, y0 * 4096.0), "small-tile f16x2: range shift 12 is not an exact rescaling"
    return worst


def run_unet(hip, dev, dim, cdt, n_hyp=2, hw=8, tile=0):
    """Whole U-Net schedule with every eligible conv on the small-tile kernel (fused GroupNorm statistics, fused PreNorm, concat
    sources, phase convs, space-to-depth, NCHW bank output) against the oracle."""
    from nope_amd.u_net import UNet
    from nope_amd.weights import synth_init_
    from tests.util import StubEncoder
    os.environ["NOPE_CONV_SMALL"] = "2"
    os.environ["NOPE_SMALL_TILE"] = str(tile)
    try:
        u = UNet(u_net_dim=dim, rot_representation_dim=6, encoder=StubEncoder(8), pose_mlp_name="single_layer", compute_dtype=cdt)
        synth_init_(u, 2022)
        sd = {k: v.clone() for k, v in u.own_state_dict().items()}
        u = u.to(dev)
        g = torch.Generator().manual_seed(23)
        x, pose = torch.randn(1, 8, hw, hw, generator=g), torch.randn(1, n_hyp, 6, generator=g)
        y = u.forward_hypotheses(x.to(dev), pose.to(dev)).cpu()[0]
        want = R.unet_forward(sd, x.expand(n_hyp, -1, -1, -1), pose[0])
        return rel(y, want)
    finally:
        os.environ.pop("NOPE_CONV_SMALL")
        os.environ.pop("NOPE_SMALL_TILE")


if __name__ == "__main__":
    hip._set_library_for_testing(hip.NopeLib(build_emu.build()))
    if "--unet16" in sys.argv:
        e = run_unet(hip, "cpu", 64, "f16", n_hyp=1, hw=8)
        assert e < 8e-3, e
        print(f"unet f16 (u_net_dim 64) on the small-tile kernel: rel err {e:.2e}")
        print("small_emu_case OK")
        sys.exit(0)
    if "--unet16split" in sys.argv:
        # every 3x3 conv with >= 2 channel chunks as a split-K launch of the tap-resident kernel: raw partials through the per-wave panels,
        # the reduce kernel that also emits the GroupNorm statistics (per sample at 4 x 4, per 64-row block at 8 x 8)
        os.environ["NOPE_HALO_SPLIT_MIN_CHUNKS"] = "2"
        e = run_unet(hip, "cpu", 64, "f16", n_hyp=2, hw=8)
        assert e < 8e-3, e
        print(f"unet f16 (u_net_dim 64) with split-K on the tap-resident kernel: rel err {e:.2e}")
        print("small_emu_case OK")
        sys.exit(0)
    if "--x2" in sys.argv:
        print(f"small-tile f16x2 worst/tol {run_x2(hip, 'cpu'):.3f}")
        print("small_emu_case OK")
        sys.exit(0)
    if "--unet32" in sys.argv:
        e = run_unet(hip, "cpu", 32, "f32", n_hyp=2, hw=8, tile=1)
        assert e < 1e-4, e
        print(f"unet f32 (u_net_dim 32) on the small-tile kernel, 128 x 128 tiles: rel err {e:.2e}")
        print("small_emu_case OK")
        sys.exit(0)
    dts = tuple(int(v) for v in sys.argv[sys.argv.index("--dts") + 1].split(",")) if "--dts" in sys.argv else (1, 0)
    tiles = tuple(int(v) for v in sys.argv[sys.argv.index("--tiles") + 1].split(",")) if "--tiles" in sys.argv else (0, 1, 2, 3)
    w = run(hip, "cpu", dts=dts, tiles=tiles, light="--light" in sys.argv)
    print(f"small_emu_case OK worst/tol {w:.3f}")
