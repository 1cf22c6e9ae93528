"""Runs the 256 x 192 ping-pong conv kernel (kernels_gemm_pp.hip) on small shapes under tests/hipemu and checks it
against torch convolutions.  Executed as a subprocess by tests/test_conv_pingpong.py with different interpreter
settings (HIPEMU_DMA=late: LDS-DMA lands at the covering wait; HIPEMU_SHUFFLE: wave scheduling order)."""
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


def run(hip, dev, dts=(1, 0), light=False):
    os.environ["NOPE_CONV_PP"] = "13"          # ping-pong kernel wherever it applies, position-major on it too
    g = torch.Generator().manual_seed(77)
    rn = lambda *s: torch.randn(*s, generator=g)
    d = lambda x: x.to(dev)
    worst = 0.0
    for dt in dts:
        q = lambda x: x.to(hip.torch_dtype(dt)).float()
        tol = {0: 2e-5, 1: 4e-2, 2: 5e-3, 3: 3e-5}[dt]      # f32, bf16, f16, bf16x3 (f32 storage, split-precision MFMA)
        C = 32 if dt in (0, 3) else 64           # one 128-byte K step per tap and source

        def chk(y, ref, what, t=tol):
            nonlocal worst
            e = rel(hip.to_nchw(y, dt).cpu(), ref)
            worst = max(worst, e / t)
            assert e < t, (what, dt, e)
        # 3x3 over a virtual concat (broadcast first source), ragged M (270 rows = 2 tiles), two N tiles, bias: 18 K steps
        x1, x2 = rn(1, C, 10, 9), rn(3, C, 10, 9)
        w, b = rn(200, 2 * C, 3, 3) / 30, rn(200)
        y = hip.op_conv(dt, hip.to_nhwc(d(x1), dt), d(w), d(b), src2=hip.to_nhwc(d(x2), dt), rep1=3, rep2=1, n_hyp=3)
        chk(y, F.conv2d(torch.cat((q(x1).expand(3, -1, -1, -1), q(x2)), 1), q(w), b, padding=1), "3x3 concat")
        # the same concat without a broadcast source: tap-resident 3x3 kernel, chunks from two sources, ragged M, 2 N tiles
        x1f = rn(3, C, 10, 9)
        y = hip.op_conv(dt, hip.to_nhwc(d(x1f), dt), d(w), d(b), src2=hip.to_nhwc(d(x2), dt))
        chk(y, F.conv2d(torch.cat((q(x1f), q(x2)), 1), q(w), b, padding=1), "halo 3x3 concat")
        # ... and with a broadcast SECOND source (the U-Net's final block: per-hypothesis activations ++ per-reference skip): still
        # the tap-resident kernel, per-piece offsets for the broadcast source
        w2 = torch.cat((w[:, C:], w[:, :C]), 1)
        y = hip.op_conv(dt, hip.to_nhwc(d(x2), dt), d(w2), d(b), src2=hip.to_nhwc(d(x1), dt), rep1=1, rep2=3, n_hyp=3)
        chk(y, F.conv2d(torch.cat((q(x2), q(x1).expand(3, -1, -1, -1)), 1), q(w2), b, padding=1), "halo 3x3 concat, broadcast second source")
        # tap-resident 3x3 kernel (A stage = tile pixels + halo, loaded once per channel chunk): maps of 4x4 pixels with many
        # samples per tile (halo rows belong to neighbouring samples), widest supported map (W = 32), a single channel chunk
        xs, ws_, bs = rn(40, C, 4, 4), rn(24, C, 3, 3) / (3 * C ** 0.5), rn(24)
        y = hip.op_conv(dt, hip.to_nhwc(d(xs), dt), d(ws_), d(bs))
        chk(y, F.conv2d(q(xs), q(ws_), bs, padding=1), "halo 4x4 x 40 samples")
        # (bf16 without a residual leaves through the packed two-rows-per-dword panel, with one through the f32 panel: same bits)
        y0 = hip.op_conv(dt, hip.to_nhwc(d(xs), dt), d(ws_), d(bs), resid=torch.zeros_like(y))
        assert torch.equal(y, y0), "packed epilogue panel differs from the f32 panel"
        xw = rn(1, 2 * C, 9, 32)
        ww = rn(16, 2 * C, 3, 3) / (3 * (2 * C) ** 0.5)
        y = hip.op_conv(dt, hip.to_nhwc(d(xw), dt), d(ww), None)
        chk(y, F.conv2d(q(xw), q(ww), padding=1), "halo W=32")
        os.environ["NOPE_CONV_PP"] = "29"      # same shapes per tap on the ping-pong kernel: identical bits
        y2 = hip.op_conv(dt, hip.to_nhwc(d(xw), dt), d(ww), None)
        os.environ["NOPE_CONV_PP"] = "13"
        assert torch.equal(y, y2), "tap-resident kernel differs from the per-tap kernel"
        if dt != 0:
            # a workgroup walking several tiles (next tile's prologue in flight under the epilogue): 8 workgroups, 3 tiles each
            # with one channel chunk, then 2 tiles each with two chunks (the walk ends on the other A stage); same bits as
            # one tile per workgroup
            # (third case: two weight panels -- every XCD takes both panels of its run of M tiles, 16 workgroups x 2 tiles)
            # (and four panels over a 4 x 2 XCD grid: each XCD two panels of its run, 16 workgroups x 4 tiles)
            for (ns, cc, side, co, wgs, gn) in ((384, C, 4, 24, "8", "1"), (256, 2 * C, 4, 24, "8", "1"), (64, C, 8, 200, "16", "1"), (64, C, 8, 600, "16", "2")):
                xq, wq, bq = rn(ns, cc, side, side), rn(co, cc, 3, 3) / (3 * cc ** 0.5), rn(co)
                os.environ["NOPE_CONV_PP"] = "11"      # (4x4 maps in (sample, pixel) row order too)
                os.environ["NOPE_HALO_PERSIST"] = wgs
                os.environ["NOPE_XCD_GN"] = gn
                y = hip.op_conv(dt, hip.to_nhwc(d(xq), dt), d(wq), d(bq))
                os.environ["NOPE_HALO_PERSIST"] = "0"
                os.environ["NOPE_XCD_MAP"] = "1"
                y2 = hip.op_conv(dt, hip.to_nhwc(d(xq), dt), d(wq), d(bq))
                for k in ("NOPE_HALO_PERSIST", "NOPE_XCD_MAP", "NOPE_XCD_GN"):
                    os.environ.pop(k)
                os.environ["NOPE_CONV_PP"] = "13"
                chk(y, F.conv2d(q(xq), q(wq), bq, padding=1), f"halo walk {ns}x{cc}")
                assert torch.equal(y, y2), "walking workgroups differ from one tile per workgroup"
        # 1x1, one K step (nk = 1) and two (nk = 2), residual
        w1, rs = rn(24, C, 1, 1) / 8, rn(3, 24, 10, 9)
        y = hip.op_conv(dt, hip.to_nhwc(d(x2), dt), d(w1), None, resid=hip.to_nhwc(d(rs), dt))
        chk(y, F.conv2d(q(x2), q(w1)) + q(rs), "1x1 nk=1")
        x3, w2 = rn(2, 2 * C, 12, 11), rn(40, 2 * C, 1, 1) / 10
        y = hip.op_conv(dt, hip.to_nhwc(d(x3), dt), d(w2), d(rn(40)) * 0)
        chk(y, F.conv2d(q(x3), q(w2)), "1x1 nk=2")
        if light:
            continue
        # three K steps (nk = 3: the B ring wraps once)
        x5, w5 = rn(2, 3 * C, 12, 11), rn(16, 3 * C, 1, 1) / 12
        y = hip.op_conv(dt, hip.to_nhwc(d(x5), dt), d(w5), None)
        chk(y, F.conv2d(q(x5), q(w5)), "1x1 nk=3")
        # nearest-x2 + 3x3 as four 2x2 phase convs; space-to-depth + 1x1
        wu, bu = rn(40, C, 3, 3) / 24, rn(40)
        y = hip.op_conv(dt, hip.to_nhwc(d(x2), dt), d(wu), d(bu), mode=hip.CONV_UP2P)
        chk(y, R.hard_upsample(q(x2), {"1.weight": wu, "1.bias": bu}, ""), "up2p", {0: tol, 1: 6e-2, 2: 8e-3, 3: tol}[dt])
        x4 = rn(5, C, 12, 10)
        wd, bd = rn(72, 4 * C, 1, 1) / 16, rn(72)
        y = hip.op_conv(dt, hip.to_nhwc(d(x4), dt), d(wd), d(bd), mode=hip.CONV_DOWN2)
        chk(y, R.hard_downsample(q(x4), {"1.weight": q(wd), "1.bias": bd}, ""), "down2")
        # position-major rows on the ping-pong kernel: 256 samples of a 2x2 map, residual
        xp, wp, bp, rp = rn(256, C, 2, 2), rn(24, C, 3, 3) / (3 * C ** 0.5), rn(24), rn(256, 24, 2, 2)
        y = hip.op_conv(dt, hip.to_nhwc(d(xp), dt), d(wp), d(bp), resid=hip.to_nhwc(d(rp), dt))
        chk(y, F.conv2d(q(xp), q(wp), bp, padding=1) + q(rp), "posmajor")
    os.environ.pop("NOPE_CONV_PP")
    return worst


def run_unet(hip, dev, dim, cdt, n_hyp=2, hw=8):
    """Whole U-Net schedule with every eligible conv on the ping-pong kernel (fused GroupNorm statistics, fused PreNorm,
    concat sources, phase convs, space-to-depth) against the oracle."""
    from nope_amd.u_net import UNet
    from nope_amd.weights import synth_init_
    from tests.util import StubEncoder
    os.environ["NOPE_CONV_PP"] = "9"
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
        os.environ.pop("NOPE_CONV_PP")


if __name__ == "__main__":
    hip._set_library_for_testing(hip.NopeLib(build_emu.build()))
    if "--unet16" in sys.argv:
        # 16-bit storage needs 64-channel K steps for the LDS-DMA kernels: u_net_dim 64 on an 8 x 8 map (level 0: whole 64-row blocks
        # per sample -> fused GroupNorm statistics with the range test, fused PreNorm through the packed epilogue panel)
        e = run_unet(hip, "cpu", 64, "f16", n_hyp=1, hw=8)
        assert e < 8e-3, e
        print(f"unet f16 (u_net_dim 64) on the LDS-DMA / ping-pong kernels: rel err {e:.2e}")
        print("pp_emu_case OK")
        sys.exit(0)
    dts = tuple(int(v) for v in sys.argv[sys.argv.index("--dts") + 1].split(",")) if "--dts" in sys.argv else (1, 0)
    w = run(hip, "cpu", dts=dts, light="--light" in sys.argv)
    if "--unet" in sys.argv:
        e = run_unet(hip, "cpu", 32, "f32")
        assert e < 1e-4, e
        print(f"unet f32 on the ping-pong kernel: rel err {e:.2e}")
    print(f"pp_emu_case OK worst/tol {w:.3f}")
