"""The lean wide epilogue of the f32-storage modes (epilogue_wide, LEANM = 1: conv_gemm_common.h) against the generic row loop it replaces, bit
for bit, on every kernel that has a LEAN instantiation: the 128 x 192 LDS-DMA kernel, the streaming 1x1 kernel, the per-tap ping-pong kernel
(1x1 and space-to-depth) and the tap-resident 3x3 kernel, in bf16x3 and f16x2, with bias / residual / ReLU; plus shapes the launcher must
keep OFF the lean path (ragged rows, a channel count that is no multiple of 32); channel counts of 192 + 64 and 192 + 32 stay ON it (whole 32-column passes).  Run by tests/test_conv_stream.py under tests/hipemu and on
the GPU.  The fused-PreNorm and GroupNorm-statistics forms are covered by the U-Net cases (stream_emu_case.run_unet: bit-identical too)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "hipemu"))
import torch
import torch.nn.functional as F

from tests.stream_emu_case import _env, _restore
from tests.util import rel

X2 = 4


def run(hip, dev, dts=(3, X2), light=False):
    g = torch.Generator().manual_seed(612)
    rn = lambda *s: torch.randn(*s, generator=g)
    d = lambda x: x.to(dev)
    worst = 0.0
    C = 32
    for dt in dts:
        tol = 3e-5

        def both(fn, what, ref, **env):
            nonlocal worst
            old = _env(NOPE_EPILOGUE_LEAN=0, **env)
            y0 = fn()
            _restore(old)
            old = _env(NOPE_EPILOGUE_LEAN=1, **env)
            y1 = fn()
            _restore(old)
            e = rel(hip.to_nchw(y1, 0).cpu(), ref)
            worst = max(worst, e / tol)
            assert e < tol, (what, dt, e)
            assert torch.equal(y0, y1), (what, dt, "lean epilogue differs from the generic one")

        # 128 x 192 LDS-DMA kernel: 1x1, 2 K steps, 256 rows x 384 columns; bias; bias + residual + ReLU
        x, w, b, r = rn(1, 2 * C, 16, 16), rn(384, 2 * C, 1, 1) / 8, rn(384), rn(1, 384, 16, 16)
        if dt == 3:
            both(lambda: hip.op_conv(dt, hip.to_nhwc(d(x), 0), d(w), d(b)), "dma 1x1", F.conv2d(x, w, b), NOPE_CONV_PP=0, NOPE_CONV_SMALL=0, NOPE_CONV_STREAM=0)
            both(lambda: hip.op_conv(dt, hip.to_nhwc(d(x), 0), d(w), d(b), resid=hip.to_nhwc(d(r), 0), act_relu=True), "dma 1x1 + residual + relu",
                 F.relu(F.conv2d(x, w, b) + r), NOPE_CONV_PP=0, NOPE_CONV_SMALL=0, NOPE_CONV_STREAM=0)
            # the streaming kernel (8 samples: 16 tiles, a walk of two)
            x8, r8 = rn(8, 2 * C, 16, 16), rn(8, 192, 16, 16)
            w8, b8 = rn(192, 2 * C, 1, 1) / 8, rn(192)
            both(lambda: hip.op_conv(dt, hip.to_nhwc(d(x8), 0), d(w8), d(b8), resid=hip.to_nhwc(d(r8), 0)), "stream 1x1 + residual", F.conv2d(x8, w8, b8) + r8,
                 NOPE_CONV_PP=0, NOPE_CONV_SMALL=0, NOPE_CONV_STREAM=1, NOPE_STREAM_GRID=8, NOPE_STREAM_MIN_ITERS=1)
            # lean with a ragged last panel: 256 columns = 192 + 64 (whole 32-column passes: the LDM variant's channel counts), with residual
            w6, b6, r6 = rn(256, 2 * C, 1, 1) / 8, rn(256), rn(1, 256, 16, 16)
            both(lambda: hip.op_conv(dt, hip.to_nhwc(d(x), 0), d(w6), d(b6), resid=hip.to_nhwc(d(r6), 0)), "dma 1x1, 256 columns + residual", F.conv2d(x, w6, b6) + r6,
                 NOPE_CONV_PP=0, NOPE_CONV_SMALL=0, NOPE_CONV_STREAM=0)
            # NOT lean: 200 columns (the second weight panel is ragged), 270 rows
            wr, br = rn(200, 2 * C, 1, 1) / 8, rn(200)
            both(lambda: hip.op_conv(dt, hip.to_nhwc(d(x), 0), d(wr), d(br)), "dma 1x1 ragged columns", F.conv2d(x, wr, br), NOPE_CONV_PP=0, NOPE_CONV_SMALL=0, NOPE_CONV_STREAM=0)
            x3 = rn(3, C, 10, 9)
            w3 = rn(192, C, 1, 1) / C ** 0.5
            both(lambda: hip.op_conv(dt, hip.to_nhwc(d(x3), 0), d(w3), None), "dma 1x1 ragged rows", F.conv2d(x3, w3), NOPE_CONV_PP=0, NOPE_CONV_SMALL=0, NOPE_CONV_STREAM=0)
        # the ping-pong kernels at any tile count (NOPE_CONV_PP bit 3); f16x2 is checked against its own restated arithmetic elsewhere: here
        # only lean == generic and the f32 convolution within the mode's tolerance
        pp = dict(NOPE_CONV_PP=11, NOPE_CONV_SMALL=0, NOPE_CONV_STREAM=0)
        xs, ws_, bs, rs = rn(2, 2 * C, 16, 16), rn(192, 2 * C, 3, 3) / 24, rn(192), rn(2, 192, 16, 16)
        both(lambda: hip.op_conv(dt, hip.to_nhwc(d(xs), 0), d(ws_), d(bs), resid=hip.to_nhwc(d(rs), 0)), "tap-resident 3x3 + residual", F.conv2d(xs, ws_, bs, padding=1) + rs, **pp)
        w7, b7 = rn(224, 2 * C, 3, 3) / 24, rn(224)      # 224 columns = 192 + 32: one pass of the second panel's first wave column, none of its second
        both(lambda: hip.op_conv(dt, hip.to_nhwc(d(xs), 0), d(w7), d(b7)), "tap-resident 3x3, 224 columns", F.conv2d(xs, w7, b7, padding=1), **pp)
        if not light:
            both(lambda: hip.op_conv(dt, hip.to_nhwc(d(xs), 0), d(ws_), d(bs), act_relu=True), "tap-resident 3x3 + relu", F.relu(F.conv2d(xs, ws_, bs, padding=1)), **pp)
            both(lambda: hip.op_conv(dt, hip.to_nhwc(d(x), 0), d(w), d(b)), "per-tap 1x1", F.conv2d(x, w, b), **pp)
            xd, wd, bd = rn(2, C, 32, 32), rn(192, 4 * C, 1, 1) / 12, rn(192)
            both(lambda: hip.op_conv(dt, hip.to_nhwc(d(xd), 0), d(wd), d(bd), mode=hip.CONV_DOWN2), "per-tap space-to-depth", F.conv2d(xd, wd.view(192, C, 2, 2), bd, stride=2), **pp)
    return worst


if __name__ == "__main__":
    import build_emu
    from nope_amd import hip
    hip._set_library_for_testing(hip.NopeLib(build_emu.build()))
    dts = tuple(int(v) for v in sys.argv[sys.argv.index("--dts") + 1].split(",")) if "--dts" in sys.argv else (3, X2)
    w = run(hip, "cpu", dts=dts, light="--light" in sys.argv)
    print(f"lean_emu_case OK worst/tol {w:.3f}")
