"""The four phase convs of an up-sampling (HardUpsample, model_utils.py:161-165) on the tap-resident kernel (conv3x3_halo_kernel<..., UP>,
kernels_gemm_pp.hip: six positions per channel chunk, four of them K steps) in the f32-storage modes (bf16x3, f16x2), on small shapes under
tests/hipemu: against the f32 convolution of the nearest-x2 map, against the restated f16x2 arithmetic, and BIT FOR BIT against the per-tap
ping-pong kernel (NOPE_UP2P_HALO=0) -- same K order, same operands, same accumulation chain.  One, two and three channel chunks (both A stages,
the last-chunk path), maps from 4 x 4 with many samples per tile to the widest supported one (30 pixels: every wave has a fifth piece), one and two weight
panels, ragged M.  Run by tests/test_conv_pingpong.py with the interpreter's adversarial settings; `run(hip, "cuda")` is the GPU form."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "hipemu"))
import torch
import torch.nn.functional as F

from tests.util import rel
from tests.x2_emu_case import X2, up2p_conv, up2p_phase_weights, x2_reference

BF16X3 = 3


def launch_trace(fn):
    """stderr of the library while fn() runs with NOPE_CONV_TRACE set: one line per conv launch (kernel, shape, grid)."""
    import tempfile
    sys.stderr.flush()
    saved = os.dup(2)
    os.environ["NOPE_CONV_TRACE"] = "1"
    with tempfile.TemporaryFile(mode="w+b") as f:
        os.dup2(f.fileno(), 2)
        try:
            out = fn()
        finally:
            os.dup2(saved, 2)
            os.close(saved)
            os.environ.pop("NOPE_CONV_TRACE")
        f.seek(0)
        return out, f.read().decode(errors="replace")


def run(hip, dev, light=False):
    os.environ["NOPE_CONV_PP"] = "13"          # ping-pong kernels at any tile count
    os.environ["NOPE_UP2P_HALO"] = "2"         # (the default, 1, takes the tap-resident form for f16x2 layers only: measured flat for bf16x3)
    g = torch.Generator().manual_seed(314)
    rn = lambda *s: torch.randn(*s, generator=g)
    d = lambda t: t.to(dev)
    C = 32
    worst = 0.0
    # samples, channels, H, W, Cout
    shapes = [(3, C, 5, 6, 40), (3, 2 * C, 10, 9, 200), (40, 3 * C, 4, 4, 24), (1, 2 * C, 9, 30, 16)]
    if light:
        shapes = shapes[1:2]       # two chunks (both A stages, the last-chunk path), ragged M, two weight panels
    try:
        for n, c, h, w_, co in shapes:
            x, w, b = rn(n, c, h, w_), rn(co, c, 3, 3) / (3 * c ** 0.5), rn(co)
            want32 = F.conv2d(F.interpolate(x, scale_factor=2, mode="nearest"), w, b, padding=1)
            for dt in (BF16X3, X2):
                y, tr = launch_trace(lambda: hip.op_conv(dt, hip.to_nhwc(d(x), 0), d(w), d(b), mode=hip.CONV_UP2P))
                assert "conv halo256 mode 3" in tr, ("the plan did not pick the tap-resident kernel", tr)
                os.environ["NOPE_UP2P_HALO"] = "0"
                y_tap = hip.op_conv(dt, hip.to_nhwc(d(x), 0), d(w), d(b), mode=hip.CONV_UP2P)
                os.environ["NOPE_UP2P_HALO"] = "2"
                assert torch.equal(y, y_tap), ("tap-resident phase convs differ from the per-tap kernel", dt, n, c, h, w_, co,
                                               float((y - y_tap).abs().max()))
                yy = hip.to_nchw(y, 0).cpu()
                e32 = rel(yy, want32)
                worst = max(worst, e32 / 3e-5)
                assert e32 < 3e-5, ("vs the f32 convolution", dt, n, c, h, w_, co, e32)
                if dt == X2:
                    e = rel(yy, x2_reference(x, up2p_phase_weights(w), b, up2p_conv))
                    worst = max(worst, e / 2e-6)
                    assert e < 2e-6, ("vs the restated arithmetic", n, c, h, w_, co, e)
        # the range shift of the layer: an exact rescaling here too (SHIFT instantiation)
        xr, wr = torch.sign(rn(3, 2 * C, 6, 5)) * (0.5 + torch.rand(3, 2 * C, 6, 5, generator=g)), rn(40, 2 * C, 3, 3) / (3 * (2 * C) ** 0.5)
        y0 = hip.op_conv(X2, hip.to_nhwc(d(xr), 0), d(wr), None, mode=hip.CONV_UP2P)
        y12 = hip.op_conv(X2, hip.to_nhwc(d(xr * 4096.0), 0), d(wr), None, mode=hip.CONV_UP2P, x2_shift=12)
        assert torch.equal(y12, y0 * 4096.0), "phase convs: range shift 12 is not an exact rescaling"
    finally:
        os.environ.pop("NOPE_CONV_PP", None)
        os.environ.pop("NOPE_UP2P_HALO", None)
    return worst


if __name__ == "__main__":
    import build_emu
    from nope_amd import hip
    hip._set_library_for_testing(hip.NopeLib(build_emu.build()))
    print("up2p ok", run(hip, "cpu", light="--light" in sys.argv))
