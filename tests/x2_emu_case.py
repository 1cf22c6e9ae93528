"""The ping-pong kernels' NOPE_F16X2 instantiations (f16 hi x hi + one MX-scaled fp8 MFMA for both cross terms, kernels_gemm_pp.hip /
Tile<f16x2_t>: the tap-resident 3x3 kernel, which rewrites its staged rows in LDS, and the per-tap kernel -- 1x1, space-to-depth and
phase convs --, which splits them in registers) on small shapes under tests/hipemu: against the f32 convolution (tolerance of the mode)
AND against a torch restatement of the exact arithmetic the kernels perform (operands rounded as convert_piece / prep_hi + prep_lo /
encode_w_x2_kernel round them: tight tolerance -- a wrong slot, byte position, pre-scale or block scale shows here, not in the loose check).  Run by tests/test_conv_pingpong.py with the
interpreter's adversarial settings; `run(hip, "cuda")` is the GPU form of the same cases."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "hipemu"))
import torch
import torch.nn.functional as F

from tests.util import rel

X2 = 4                      # hip.F16X2
A_LO_SHIFT, A_SHIFT, W_LO_EXTRA = 9, -2, 11      # nope_common.h: kX2ALoShift, kX2AShift, kX2WLoExtra


def q8(x, log2_scale):
    s = 2.0 ** log2_scale
    return (x * s).clamp(-448.0, 448.0).to(torch.float8_e4m3fn).float() / s


def x2_reference(x, w, b, conv=None):
    """out = f16(a) f16(w) + e4m3(a_lo 2^9) e4m3(w 2^sw) 2^-(9+sw) + e4m3(a 2^-2) e4m3(w_lo 2^(sw+11)) 2^-(9+sw), f64 accumulation.
    w: the weights AS THE GEMM SEES THEM (the phase-summed sets of an UP2P conv); conv(a, w): the bilinear map, default a 3x3 conv, padding 1"""
    conv = conv or (lambda a, ww: F.conv2d(a, ww, None, padding=1))
    m = float(w.abs().max())
    e = torch.frexp(torch.tensor(m))[1].item() - 1          # floor(log2 m)
    sw = 7 - e
    ah, wh = x.clamp(-65504, 65504).half().float(), w.half().float()
    al, wl = x - ah, w - wh
    d = torch.float64
    out = conv(ah.to(d), wh.to(d))
    out = out + conv(q8(al, A_LO_SHIFT).to(d), q8(w, sw).to(d))
    out = out + conv(q8(x, A_SHIFT).to(d), q8(wl, sw + W_LO_EXTRA).to(d))
    if b is not None:
        out = out + b.to(d)[None, :, None, None]
    return out.float()


def up2p_phase_weights(w):
    """nearest-x2 + 3x3 as four 2x2 phase convs over the SOURCE grid (model_utils.py:161-165; pack_up2p_w_kernel): (4, Cout, Cin, 2, 2), phase
    (py, px), tap (ty, tx) = source pixel (y + ty + py - 1, x + tx + px - 1), carrying the sum of the 3x3 taps that land on it"""
    rows = {(0, 0): [0], (0, 1): [1, 2], (1, 0): [0, 1], (1, 1): [2]}
    out = torch.zeros(4, w.shape[0], w.shape[1], 2, 2, dtype=w.dtype)
    for py in range(2):
        for px in range(2):
            for ty in range(2):
                for tx in range(2):
                    out[py * 2 + px, :, :, ty, tx] = w[:, :, rows[(py, ty)]][:, :, :, rows[(px, tx)]].sum((2, 3))
    return out


def up2p_conv(a, wp):
    n, _, H, W = a.shape
    ap = F.pad(a, (1, 1, 1, 1))
    out = a.new_zeros(n, wp.shape[1], 2 * H, 2 * W)
    for py in range(2):
        for px in range(2):
            out[:, :, py::2, px::2] = F.conv2d(ap[:, :, py:py + H + 1, px:px + W + 1], wp[py * 2 + px])
    return out


def run(hip, dev, light=False):
    os.environ["NOPE_CONV_PP"] = "13"          # ping-pong kernels at any tile count
    g = torch.Generator().manual_seed(91)
    rn = lambda *s: torch.randn(*s, generator=g)
    d = lambda t: t.to(dev)
    C = 32
    worst = 0.0

    def chk(y, x, w, b, what, extra=None, want32=None, conv=None, w_gemm=None):
        nonlocal worst
        y = hip.to_nchw(y, 0).cpu()
        want32 = F.conv2d(x, w, b, padding=1) if want32 is None else want32
        want = x2_reference(x, w if w_gemm is None else w_gemm, b, conv)
        if extra is not None:
            want32, want = want32 + extra, want + extra
        e_exact, e32 = rel(y, want), rel(y, want32)
        worst = max(worst, e_exact / 2e-6, e32 / 3e-5)
        assert e_exact < 2e-6, (what, "vs the restated arithmetic", e_exact)
        assert e32 < 3e-5, (what, "vs the f32 convolution", e32)
    try:
        # concat of two sources, ragged M (270 rows = 2 tiles), two N tiles, bias: chunks from both sources
        x1, x2 = rn(3, C, 10, 9), rn(3, C, 10, 9)
        w, b = rn(200, 2 * C, 3, 3) / 30, rn(200)
        y = hip.op_conv(X2, hip.to_nhwc(d(x1), 0), d(w), d(b), src2=hip.to_nhwc(d(x2), 0))
        chk(y, torch.cat((x1, x2), 1), w, b, "halo 3x3 concat")
        # broadcast second source
        xb = rn(1, C, 10, 9)
        y = hip.op_conv(X2, hip.to_nhwc(d(x2), 0), d(w), d(b), src2=hip.to_nhwc(d(xb), 0), rep1=1, rep2=3, n_hyp=3)
        chk(y, torch.cat((x2, xb.expand(3, -1, -1, -1)), 1), w, b, "halo 3x3 concat, broadcast second source")
        # 4x4 maps, many samples per tile; activations with a wide dynamic range (large values saturate the fp8 parts, tiny ones flush):
        # the restated arithmetic saturates / flushes the same way
        xs = rn(40, C, 4, 4) * torch.exp(2.5 * rn(40, C, 4, 4))
        xs[0, 0, 0, 0], xs[1, 3, 2, 1] = 3000.0, -70000.0
        ws_, bs = rn(24, C, 3, 3) / (3 * C ** 0.5), rn(24)
        y = hip.op_conv(X2, hip.to_nhwc(d(xs), 0), d(ws_), d(bs))
        yy = hip.to_nchw(y, 0).cpu()
        want = x2_reference(xs, ws_, bs)
        assert rel(yy, want) < 2e-6, ("halo 4x4 x 40 samples, wide range", rel(yy, want))
        # the per-layer range shift t (tail word 3): activations scaled by 2^12 under t = 12 give EXACTLY 2^12 x the t = 0 result (every
        # operand conversion and the block scale move by the same power of two) -- tap-resident and per-tap kernels; at t = 0 the same
        # input saturates the e4m3(a) operand (|a| > 1792) and the result is worse than 1e-4
        # (|a| in [0.5, 1.5): the f16 hi part -- the one operand without a pre-scale -- stays a normal number at every scale used here)
        xr, wr = torch.sign(rn(3, C, 10, 9)) * (0.5 + torch.rand(3, C, 10, 9, generator=g)), rn(40, C, 3, 3) / (3 * C ** 0.5)
        y0 = hip.op_conv(X2, hip.to_nhwc(d(xr), 0), d(wr), None)
        y12 = hip.op_conv(X2, hip.to_nhwc(d(xr * 4096.0), 0), d(wr), None, x2_shift=12)
        assert torch.equal(y12, y0 * 4096.0), ("halo 3x3: range shift 12 is not an exact rescaling", float((y12 / 4096.0 - y0).abs().max()), float(y0.abs().max()))
        ysat = hip.op_conv(X2, hip.to_nhwc(d(xr * 4096.0), 0), d(wr), None)
        e_sat = rel(hip.to_nchw(ysat, 0).cpu(), F.conv2d(xr * 4096.0, wr, None, padding=1))
        assert 5e-5 < e_sat < 2e-3, ("saturated cross-term operands should cost plain-f16 accuracy", e_sat)
        w1r = rn(40, C, 1, 1) / C ** 0.5
        y0 = hip.op_conv(X2, hip.to_nhwc(d(xr), 0), d(w1r), None)
        ym = hip.op_conv(X2, hip.to_nhwc(d(xr / 1024.0), 0), d(w1r), None, x2_shift=-10)
        assert torch.equal(ym * 1024.0, y0), "per-tap 1x1: range shift -10 is not an exact rescaling"
        # residual in the epilogue; widest supported map
        if not light:
            xw, ww = rn(1, 2 * C, 9, 32), rn(16, 2 * C, 3, 3) / (3 * (2 * C) ** 0.5)
            rs = rn(1, 16, 9, 32)
            y = hip.op_conv(X2, hip.to_nhwc(d(xw), 0), d(ww), None, resid=hip.to_nhwc(d(rs), 0))
            chk(y, xw, ww, None, "halo W=32 + residual", extra=rs)
            # a workgroup walking several tiles, and split-K over the channel chunks: same bits as the plain launch
            xq, wq, bq = rn(64, 4 * C, 8, 8), rn(200, 4 * C, 3, 3) / (3 * (4 * C) ** 0.5), rn(200)
            y0 = hip.op_conv(X2, hip.to_nhwc(d(xq), 0), d(wq), d(bq))
            chk(y0, xq, wq, bq, "halo 64 x 8x8, 4 chunks")
            os.environ["NOPE_CONV_PP"] = "11"
            os.environ["NOPE_HALO_PERSIST"] = "16"
            y1 = hip.op_conv(X2, hip.to_nhwc(d(xq), 0), d(wq), d(bq))
            os.environ.pop("NOPE_HALO_PERSIST")
            os.environ["NOPE_CONV_PP"] = "13"
            assert torch.equal(y0, y1), "walking workgroups differ from one tile per workgroup"
            os.environ["NOPE_HALO_SPLIT_MIN_CHUNKS"] = "2"
            y2 = hip.op_conv(X2, hip.to_nhwc(d(xq[:8]), 0), d(wq), d(bq), split_k=True)
            os.environ.pop("NOPE_HALO_SPLIT_MIN_CHUNKS")
            chk(y2, xq[:8], wq, bq, "halo split-K")
        # ---- the per-tap kernel (operand split in registers): 1x1 over a concat of two sources (two K steps, one per source), ragged M and N
        w1, b1 = rn(200, 2 * C, 1, 1) / 8, rn(200)
        y = hip.op_conv(X2, hip.to_nhwc(d(x1), 0), d(w1), d(b1), src2=hip.to_nhwc(d(x2), 0))
        xc = torch.cat((x1, x2), 1)
        chk(y, xc, w1, b1, "per-tap 1x1 concat", want32=F.conv2d(xc, w1, b1), conv=lambda a, ww: F.conv2d(a, ww))
        # space-to-depth (HardDownsample, model_utils.py:167-172): 1x1 over "b (c p1 p2) h w" = a 2x2 stride-2 conv
        xd, wd, bd = rn(3, C, 8, 10), rn(48, 4 * C, 1, 1) / 11, rn(48)
        y = hip.op_conv(X2, hip.to_nhwc(d(xd), 0), d(wd), d(bd), mode=hip.CONV_DOWN2)
        wd2 = wd.view(48, C, 2, 2)
        chk(y, xd, wd2, bd, "per-tap space-to-depth", want32=F.conv2d(xd, wd2, bd, stride=2), conv=lambda a, ww: F.conv2d(a, ww, None, stride=2))
        # nearest-x2 + 3x3 as four phase convs (HardUpsample, model_utils.py:161-165): the block scale comes from the PHASE-SUMMED weights
        xu, wu, bu = rn(3, 2 * C, 5, 6), rn(40, 2 * C, 3, 3) / 24, rn(40)
        y = hip.op_conv(X2, hip.to_nhwc(d(xu), 0), d(wu), d(bu), mode=hip.CONV_UP2P)
        chk(y, xu, wu, bu, "per-tap phase convs", want32=F.conv2d(F.interpolate(xu, scale_factor=2, mode="nearest"), wu, bu, padding=1),
            conv=up2p_conv, w_gemm=up2p_phase_weights(wu))
        if not light:
            # 3x3 in position-major row order (the 4x4 level at hundreds of hypotheses): padding taps skipped per tile
            xp, wp_, bp = rn(256, C, 4, 4), rn(24, C, 3, 3) / 17, rn(24)
            y = hip.op_conv(X2, hip.to_nhwc(d(xp), 0), d(wp_), d(bp))
            chk(y, xp, wp_, bp, "per-tap 3x3, position-major")
        # a launch no ping-pong kernel takes (here: too few tiles under the default plan): the element type refuses it
        os.environ.pop("NOPE_CONV_PP")
        try:
            hip.op_conv(X2, hip.to_nhwc(d(x2), 0), d(rn(24, C, 1, 1)), None)
            raise AssertionError("NOPE_F16X2 accepted a launch that runs on the small-tile kernel")
        except hip.NopeError:
            pass
    finally:
        os.environ.pop("NOPE_CONV_PP", None)
    return worst


def run_unet_range(hip, dev, dim=64, n_hyp=5, hw=16, scales=(1.0, 1e2, 1e3, 1e4, 3e5)):
    """The f16x2 mode OFF the benchmark's activation range: the reference embedding scaled by S reaches the first ResnetBlock's conv
    un-normalised (model_utils.py:271-272: block1 sees the residual stream as it is), so |a| grows with S.  Every forward must stay inside
    the mode's accuracy: the U-Net handle reads the per-layer maxima after each call (nope_unet_x2_range_check), re-centres the shifts and
    repeats the call when a layer left its window (at any magnitude: the tile works on a * 2^-t).  Returns [(S, rel err vs oracle, range events, x2 still on)]."""
    import warnings
    from nope_amd.u_net import UNet
    from nope_amd.weights import synth_init_
    from oracle import nope_ref as R
    from tests.util import StubEncoder
    os.environ["NOPE_CONV_PP"] = "9"
    saved_mode = os.environ.get("NOPE_X2_RANGE_CHECK")
    os.environ["NOPE_X2_RANGE_CHECK"] = "2"         # range_mode "repeat": synchronise, read the verdict, issue the forward again (hip.UNetHandle)
    out = []
    try:
        u = UNet(u_net_dim=dim, rot_representation_dim=6, encoder=StubEncoder(8), pose_mlp_name="single_layer", compute_dtype="f16x2")
        synth_init_(u, 2022)
        sd = {k: v.clone() for k, v in u.own_state_dict().items()}
        u = u.to(dev)
        g = torch.Generator().manual_seed(23)
        x, pose = torch.randn(1, 8, hw, hw, generator=g), torch.randn(1, n_hyp, 6, generator=g)
        for S in scales:
            n0 = len(u._handle.range_events) if u._handle is not None else 0
            with warnings.catch_warnings():
                warnings.simplefilter("ignore", RuntimeWarning)
                y = u.forward_hypotheses((x * S).to(dev), pose.to(dev)).cpu()[0]
            h = u._handle
            want = R.unet_forward(sd, (x * S).expand(n_hyp, -1, -1, -1), pose[0])
            out.append((S, rel(y, want), h.range_events[n0:], h.x2_enabled))
        # the default mode ("poison"): no host look in the step -- a forward whose layers left their windows returns NaNs, the next forward
        # finds the verdict, re-centres the shifts (one warning) and is accurate
        h = u._handle
        h.range_mode = "poison"
        S = scales[-1] * 64.0
        n0 = len(h.range_events)
        with warnings.catch_warnings():
            warnings.simplefilter("ignore", RuntimeWarning)
            y1 = u.forward_hypotheses((x * S).to(dev), pose.to(dev)).cpu()[0]
            if str(dev) != "cpu":
                torch.cuda.synchronize()
            y2 = u.forward_hypotheses((x * S).to(dev), pose.to(dev)).cpu()[0]
        want = R.unet_forward(sd, (x * S).expand(n_hyp, -1, -1, -1), pose[0])
        assert bool(torch.isnan(y1).all()), "an out-of-range forward must leave NaNs, not inaccurate values"
        assert any(ev["attempt"] == -1 for ev in h.range_events[n0:]), "the next forward reports the verdict"
        out.append((S, rel(y2, want), h.range_events[n0:], h.x2_enabled))
        return out
    finally:
        os.environ.pop("NOPE_CONV_PP")
        if saved_mode is None:
            os.environ.pop("NOPE_X2_RANGE_CHECK", None)
        else:
            os.environ["NOPE_X2_RANGE_CHECK"] = saved_mode


def run_unet(hip, dev, dim=64, n_hyp=5, hw=16):
    """Whole U-Net in the f16x2 compute mode against the oracle (tolerance of the mode) with the tap-resident kernel wherever it applies."""
    from nope_amd.u_net import UNet
    from nope_amd.weights import synth_init_
    from oracle import nope_ref as R
    from tests.util import StubEncoder
    os.environ["NOPE_CONV_PP"] = "9"       # ping-pong kernels (tap-resident for the 3x3 convs) at any tile count
    try:
        u = UNet(u_net_dim=dim, rot_representation_dim=6, encoder=StubEncoder(8), pose_mlp_name="single_layer", compute_dtype="f16x2")
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
    import build_emu
    from nope_amd import hip
    hip._set_library_for_testing(hip.NopeLib(build_emu.build()))
    if "--unet" in sys.argv:
        e = run_unet(hip, "cpu", 32, n_hyp=1, hw=8)
        assert e < 1e-4, e
        print("x2 unet ok", e)
    else:
        print("x2 ok", run(hip, "cpu", light="--light" in sys.argv))
