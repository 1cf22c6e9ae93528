"""Runs the streaming 1x1 conv kernel (nope_amd/csrc/kernels_gemm_stream.hip) on small shapes and checks it against torch convolutions and,
bit for bit, against the 128 x 192 LDS-DMA kernel it replaces (same tile, same MFMA stage, same K order, same epilogue).  Executed as a
subprocess by tests/test_conv_stream.py under tests/hipemu (HIPEMU_DMA=late: LDS-DMA lands at the covering COUNTED vmcnt wait -- the
five-stage activation ring that runs across tile boundaries and the wave roles are what this exercises; HIPEMU_SHUFFLE: wave order),
and imported by the GPU test with the real library.  NOPE_STREAM_GRID lets a walk of a few tiles stand for the 256-workgroup launch."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "hipemu"))
import torch
import torch.nn.functional as F

from oracle import nope_ref as R
from tests.util import rel


def _env(**kw):
    old = {k: os.environ.get(k) for k in kw}
    for k, v in kw.items():
        if v is None:
            os.environ.pop(k, None)
        else:
            os.environ[k] = str(v)
    return old


def _restore(old):
    for k, v in old.items():
        if v is None:
            os.environ.pop(k, None)
        else:
            os.environ[k] = v


def run(hip, dev, dts=(3, 1), light=False):
    """op_conv cases: K loops of 1..7 steps (shorter and longer than the ring), walks of 1..4 tiles, one / two / four weight panels,
    two sources, bias, residual, ReLU.  Returns worst error / tolerance."""
    g = torch.Generator().manual_seed(611)
    rn = lambda *s: torch.randn(*s, generator=g)
    d = lambda x: x.to(dev)
    worst = 0.0
    for dt in dts:
        q = lambda x: x.to(hip.torch_dtype(dt)).float()
        tol = {1: 4e-2, 2: 5e-3, 3: 3e-5}[dt]      # bf16, f16, bf16x3
        C = 32 if dt == 3 else 64                  # channels per 128-byte K step

        def both(fn, what, ref, grid, min_iters=1):
            nonlocal worst
            old = _env(NOPE_CONV_STREAM=0, NOPE_CONV_PP=0, NOPE_CONV_SMALL=0)
            y_old = fn()
            _restore(old)
            old = _env(NOPE_CONV_STREAM=1, NOPE_STREAM_GRID=grid, NOPE_STREAM_MIN_ITERS=min_iters, NOPE_CONV_PP=0, NOPE_CONV_SMALL=0, NOPE_CONV_TRACE=1)
            y = fn()
            y2 = fn()
            _restore(old)
            e = rel(hip.to_nchw(y, dt).cpu(), ref)
            worst = max(worst, e / tol)
            assert e < tol, (what, dt, e)
            assert torch.equal(y, y2), (what, dt, "not reproducible")
            assert torch.equal(y, y_old), (what, dt, "streaming kernel differs from the 128 x 192 kernel")

        cases = [  # (K steps, samples of 16 x 16 = 2 tiles each, Cout, grid)
            (1, 8, 192, 8), (2, 8, 192, 8), (3, 12, 192, 8), (5, 16, 384, 16), (7, 8, 192, 8), (6, 16, 768, 32),
        ]
        if light:
            cases = cases[2:5]
        for nkc, n, cout, grid in cases:
            x, w, b = rn(n, nkc * C, 16, 16), rn(cout, nkc * C, 1, 1) / (nkc * C) ** 0.5, rn(cout)
            both(lambda: hip.op_conv(dt, hip.to_nhwc(d(x), dt), d(w), d(b)), f"1x1 nk={nkc} n={n} cout={cout}", F.conv2d(q(x), q(w), b), grid)
        # two sources (the res_conv on a concatenated skip: K steps 0..1 from the first, 2..4 from the second), residual, ReLU
        x1, x2 = rn(8, 2 * C, 16, 16), rn(8, 3 * C, 16, 16)
        w, b, r = rn(200, 5 * C, 1, 1) / (5 * C) ** 0.5, rn(200), rn(8, 200, 16, 16)
        both(lambda: hip.op_conv(dt, hip.to_nhwc(d(x1), dt), d(w), d(b), src2=hip.to_nhwc(d(x2), dt), resid=hip.to_nhwc(d(r), dt), act_relu=True),
             "1x1 concat + residual + relu", F.relu(F.conv2d(torch.cat((q(x1), q(x2)), 1), q(w), b) + q(r)), 16)
        # a shape the planner must NOT hand to the streaming kernel (rows not a multiple of the tile): falls back, still equal
        x3, w3 = rn(3, C, 10, 9), rn(72, C, 1, 1) / C ** 0.5
        both(lambda: hip.op_conv(dt, hip.to_nhwc(d(x3), dt), d(w3), None), "1x1 ragged (fallback)", F.conv2d(q(x3), q(w3)), 8)
    return worst


def run_unet(hip, dev, dim, cdt, n_hyp=4, hw=16):
    """Whole U-Net schedule with the attention blocks' 1x1 convs (fused PreNorm on the qkv conv, residual on the output conv, GroupNorm
    statistics) on the streaming kernel: against the oracle, and bit for bit against the same forward without it."""
    from nope_amd.u_net import UNet
    from nope_amd.weights import synth_init_
    from tests.util import StubEncoder
    u = UNet(u_net_dim=dim, rot_representation_dim=6, encoder=StubEncoder(8), pose_mlp_name="single_layer", compute_dtype=cdt)
    synth_init_(u, 2022)
    sd = {k: v.clone() for k, v in u.own_state_dict().items()}
    u = u.to(dev)
    g = torch.Generator().manual_seed(23)
    x, pose = torch.randn(1, 8, hw, hw, generator=g), torch.randn(1, n_hyp, 6, generator=g)
    old = _env(NOPE_CONV_STREAM=0)
    y0 = u.forward_hypotheses(x.to(dev), pose.to(dev)).cpu()[0]
    _restore(old)
    old = _env(NOPE_CONV_STREAM=3, NOPE_STREAM_GRID=8, NOPE_STREAM_MIN_ITERS=1)
    y = u.forward_hypotheses(x.to(dev), pose.to(dev)).cpu()[0]
    _restore(old)
    want = R.unet_forward(sd, x.expand(n_hyp, -1, -1, -1), pose[0])
    return rel(y, want), bool(torch.equal(y, y0))


if __name__ == "__main__":
    import build_emu
    from nope_amd import hip
    hip._set_library_for_testing(hip.NopeLib(build_emu.build()))
    if "--unet" in sys.argv:
        e, same = run_unet(hip, "cpu", 64, "bf16x3", n_hyp=4, hw=16)
        print(f"unet bf16x3 (u_net_dim 64, 4 hypotheses at 16 x 16) with its 1x1 convs on the streaming kernel: rel err {e:.2e}, bit-identical {same}")
        assert e < 1e-4, e
        print("stream_emu_case OK")
        sys.exit(0)
    dts = tuple(int(v) for v in sys.argv[sys.argv.index("--dts") + 1].split(",")) if "--dts" in sys.argv else (3, 1)
    w = run(hip, "cpu", dts=dts, light="--light" in sys.argv)
    print(f"stream_emu_case OK worst/tol {w:.3f}")
