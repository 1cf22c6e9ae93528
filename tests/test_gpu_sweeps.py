"""GPU-only randomized sweeps of the operator entry points of the C ABI. Conv: against torch's own convolution (float64) on the
same (dtype-rounded) operands: 120 seeded draws over tap geometry, channel counts on and off the LDS-DMA path, virtual
concat with broadcast sources, bias / residual / ReLU / NCHW epilogues and the launch regimes that switch code paths
(position-major rows on small maps with a multiple of 128 samples, persistent tile walk above 512 tiles), each draw under three launch
policies: the default (round 4: small-tile kernel for launches that cannot fill the chip), the round-3 kernels alone (NOPE_CONV_SMALL=0)
and with the split-K scratch the runtimes hand the launcher (split-K on the tap-resident and the 128 x 192 kernel)."""
import os
import random

import pytest
import torch
import torch.nn.functional as F

from tests.util import rel

pytestmark = pytest.mark.gpu


def _ref(mode, hip, x, w, b):
    if mode == hip.CONV_PLAIN:
        return F.conv2d(x, w, b, padding=w.shape[-1] // 2)
    if mode == hip.CONV_STRIDE2:
        return F.conv2d(x, w, b, stride=2, padding=w.shape[-1] // 2)
    if mode in (hip.CONV_UP2, hip.CONV_UP2P):
        return F.conv2d(F.interpolate(x, scale_factor=2, mode="nearest"), w, b, padding=1)
    # DOWN2: einops "b c (h p1) (w p2) -> b (c p1 p2) h w" + 1x1 (model_utils.py:168-172)
    B, C, H, W = x.shape
    u = x.view(B, C, H // 2, 2, W // 2, 2).permute(0, 1, 3, 5, 2, 4).reshape(B, C * 4, H // 2, W // 2)
    return F.conv2d(u, w, b)


@pytest.mark.parametrize("dt", [0, 1, 2, 3])
def test_conv_random_sweep(gpu, dt):
    hip = gpu
    rng = random.Random(1234 + dt)
    g = torch.Generator(device="cuda").manual_seed(99 + dt)
    rn = lambda *s: torch.randn(*s, generator=g, device="cuda")
    q = lambda t: t.to(hip.torch_dtype(dt)).float()
    # f32: accumulation order only; bf16 / f16: one output rounding (2^-9 / 2^-12) + order; bf16x3 (f32 storage, three bf16 MFMA
    # passes per product): ~2^-17 per product, the dropped lo x lo term and order
    tol = {0: 1e-5, 1: 8e-3, 2: 1e-3, 3: 3e-5}[dt]
    unit = 32 if dt in (0, 3) else 64                 # channels per 128-byte K step
    worst = 0.0
    for it in range(60):
        mode = rng.choice([hip.CONV_PLAIN] * 4 + [hip.CONV_STRIDE2, hip.CONV_UP2, hip.CONV_UP2P, hip.CONV_DOWN2])
        ks = 1 if (mode in (hip.CONV_DOWN2,) or (mode in (hip.CONV_PLAIN, hip.CONV_STRIDE2) and rng.random() < 0.35)) else 3
        on_dma = rng.random() < 0.7
        c1 = unit * rng.choice([1, 2, 3]) if on_dma else 8 * rng.choice([1, 3, 5])
        c2 = 0
        if mode == hip.CONV_PLAIN and rng.random() < 0.3:
            c2 = unit * rng.choice([1, 2]) if on_dma else 8 * rng.choice([1, 2])
        cout = rng.choice([8, 24, 64, 192, 200, 384])
        regime = rng.random()
        if mode == hip.CONV_PLAIN and ks == 3 and regime < 0.25:
            n, h, w_ = 128 * rng.choice([1, 2]), rng.choice([2, 4]), rng.choice([2, 4])      # position-major
        elif mode == hip.CONV_PLAIN and regime < 0.4 and cout <= 64:
            n, h, w_ = 512, 16, 16                                                             # 1024 tiles: persistent walk
        else:
            n, h, w_ = rng.choice([1, 2, 3, 5]), 2 * rng.choice([1, 2, 3, 4]), 2 * rng.choice([1, 2, 3, 5])
        rep1 = 1
        if c2 and n % 2 == 0 and rng.random() < 0.5:
            rep1 = 2
        x1 = rn(n // rep1, c1, h, w_)
        x2 = rn(n, c2, h, w_) if c2 else None
        cin = c1 + c2
        wshape = (cout, cin * 4, 1, 1) if mode == hip.CONV_DOWN2 else (cout, cin, ks, ks)
        wt = rn(*wshape) / (wshape[1] * ks * ks) ** 0.5
        b = rn(cout) if rng.random() < 0.6 else None
        ho, wo = (2 * h, 2 * w_) if mode in (hip.CONV_UP2, hip.CONV_UP2P) else ((h // 2, w_ // 2) if mode in (hip.CONV_DOWN2, hip.CONV_STRIDE2) else (h, w_))
        use_res = rng.random() < 0.4
        relu = rng.random() < 0.3
        nchw = mode != hip.CONV_UP2P and rng.random() < 0.25       # (with and without a residual: every kernel's NCHW epilogue adds it)
        rs = rn(n, cout, ho, wo) if use_res else None
        ys = []
        for env, split_k in (({}, False), ({"NOPE_CONV_SMALL": "0"}, False), ({"NOPE_HALO_SPLIT_MIN_CHUNKS": "2"}, True)):
            os.environ.update(env)
            ys.append(hip.op_conv(dt, hip.to_nhwc(x1, dt), wt, b, src2=None if x2 is None else hip.to_nhwc(x2, dt), mode=mode, rep1=rep1,
                                  resid=None if rs is None else hip.to_nhwc(rs, dt), n_hyp=n, out_nchw=nchw, out_dtype=hip.F32, act_relu=relu,
                                  split_k=split_k))
            for k in env:
                os.environ.pop(k)
        xin = q(x1).repeat_interleave(rep1, 0)
        if x2 is not None:
            xin = torch.cat((xin, q(x2)), 1)
        # float64 reference (torch falls back to its direct kernel: no Winograd / FFT rounding in the comparison)
        want = _ref(mode, hip, xin.double(), q(wt).double(), None if b is None else b.double())
        if rs is not None:
            want = want + q(rs).double()
        if relu:
            want = F.relu(want)
        for pol, y in enumerate(ys):
            got = y if nchw else hip.to_nchw(y, dt)
            e = rel(got.double(), want)
            worst = max(worst, e)
            assert e < tol, (it, pol, mode, ks, c1, c2, cout, n, h, w_, rep1, use_res, relu, nchw, e)
    print(f"conv sweep dtype {dt}: worst rel err {worst:.2e}")


def test_conv_random_sweep_f16x2(gpu):
    """dtype 4 (NOPE_F16X2: f32 storage, one f16 + one MX-fp8 MFMA pass) exists on the ping-pong kernels only, so its sweep draws
    from the shapes they take -- 3x3 (tap-resident, whole and split along K), 1x1, space-to-depth and the four phase convs of an
    up-sampling (per-tap), one or two sources, a broadcast second source, bias / residual / ReLU, ragged tile edges in M and Cout --
    forced onto them at any tile count (NOPE_CONV_PP=11, what smoke() does), against torch's float64 convolution of the same f32
    operands.  Bound: 3e-5 (the hi x hi f16 product is exact in f32, the cross terms carry 2^-11 x e4m3's 2^-4)."""
    hip = gpu
    rng = random.Random(4321)
    g = torch.Generator(device="cuda").manual_seed(103)
    rn = lambda *s: torch.randn(*s, generator=g, device="cuda")
    dt = hip.F16X2
    worst = 0.0
    os.environ["NOPE_CONV_PP"] = "11"
    try:
        for it in range(60):
            mode = rng.choice([hip.CONV_PLAIN] * 5 + [hip.CONV_UP2P, hip.CONV_DOWN2])
            ks = 1 if (mode == hip.CONV_DOWN2 or (mode == hip.CONV_PLAIN and rng.random() < 0.3)) else 3
            c1 = 32 * rng.choice([1, 2, 3, 6, 12])
            c2 = 32 * rng.choice([1, 2, 6]) if (mode == hip.CONV_PLAIN and rng.random() < 0.35) else 0
            cout = rng.choice([8, 24, 64, 192, 200, 384])
            if rng.random() < 0.25:
                n, h, w_ = rng.choice([64, 128, 200]), rng.choice([4, 8]), rng.choice([4, 8])      # many tiles: tile walk, XCD maps
            else:
                n, h, w_ = rng.choice([1, 2, 3, 5, 9]), 2 * rng.choice([1, 2, 4, 8]), 2 * rng.choice([1, 2, 3, 8])
            rep2 = 2 if (c2 and n % 2 == 0 and rng.random() < 0.5) else 1
            x1 = rn(n, c1, h, w_)
            x2 = rn(n // rep2, c2, h, w_) if c2 else None
            cin = c1 + c2
            wshape = (cout, cin * 4, 1, 1) if mode == hip.CONV_DOWN2 else (cout, cin, ks, ks)
            wt = rn(*wshape) / (wshape[1] * ks * ks) ** 0.5
            b = rn(cout) if rng.random() < 0.6 else None
            ho, wo = (2 * h, 2 * w_) if mode == hip.CONV_UP2P else ((h // 2, w_ // 2) if mode == hip.CONV_DOWN2 else (h, w_))
            rs = rn(n, cout, ho, wo) if rng.random() < 0.4 else None
            relu = rng.random() < 0.3
            ys = []
            for env, split_k in (({}, False), ({"NOPE_HALO_SPLIT_MIN_CHUNKS": "2"}, True)):
                os.environ.update(env)
                ys.append(hip.op_conv(dt, hip.to_nhwc(x1, dt), wt, b, src2=None if x2 is None else hip.to_nhwc(x2, dt), mode=mode, rep2=rep2,
                                      resid=None if rs is None else hip.to_nhwc(rs, dt), n_hyp=n, out_dtype=hip.F32, act_relu=relu, split_k=split_k))
                for k in env:
                    os.environ.pop(k)
            xin = x1 if x2 is None else torch.cat((x1, x2.repeat_interleave(rep2, 0)), 1)
            want = _ref(mode, hip, xin.double(), wt.double(), None if b is None else b.double())
            if rs is not None:
                want = want + rs.double()
            if relu:
                want = F.relu(want)
            for pol, y in enumerate(ys):
                e = rel(hip.to_nchw(y, dt).double(), want)
                worst = max(worst, e)
                assert e < 3e-5, (it, pol, mode, ks, c1, c2, cout, n, h, w_, rep2, rs is not None, relu, e)
    finally:
        os.environ.pop("NOPE_CONV_PP", None)
    print(f"conv sweep f16x2 (ping-pong kernels forced): worst rel err {worst:.2e}")


def test_similarity_topk_random_sweep(gpu):
    """40 seeded draws of the scoring + top-k entry points against the CPU restatement: batch, template count (ragged
    against the per-workgroup split), channel count (register and LDS query paths), map size, bank dtype, k."""
    from oracle import nope_ref as R
    hip = gpu
    rng = random.Random(77)
    g = torch.Generator().manual_seed(5)
    worst = 0.0
    for it in range(40):
        B = rng.choice([1, 2, 5])
        N = rng.choice([1, 3, 17, 64, 257, 1000])
        C = rng.choice([8, 8, 16, 4, 24])
        h, w = rng.choice([(32, 32), (16, 16), (8, 8), (4, 8), (16, 32)])
        bdt = rng.choice([torch.float32, torch.bfloat16, torch.float16])
        q = torch.randn(B, C, h, w, generator=g)
        bank = torch.randn(B, N, C, h, w, generator=g)
        if N > 2:
            bank[B - 1, N // 2] = q[B - 1].to(bdt).float() if bdt != torch.float32 else q[B - 1]
            if bdt != torch.float32:
                q[B - 1] = q[B - 1].to(bdt).float()      # planted exact match must be exact in the bank dtype too
        bank_q = bank.to(bdt)
        if C > 16 and C * h * w > 16384:          # documented limit of the generic path: the query tile lives in 64 KiB of LDS
            with pytest.raises(hip.NopeError, match="unsupported"):
                hip.similarity(q.cuda(), bank_q.cuda())
            continue
        s = hip.similarity(q.cuda(), bank_q.cuda())
        want = R.similarity_scores(q, bank_q.float())
        e = rel(s.cpu(), want)
        worst = max(worst, e)
        assert e < 2e-5, (it, B, N, C, h, w, bdt, e)
        k = min(5, N)
        vals, idx = hip.topk(s, k)
        if N > 2:
            assert float(s[B - 1, N // 2]) == 0.0 and int(idx[B - 1, 0]) == N // 2
        sv, si = torch.sort(s.cpu(), dim=1, descending=True, stable=True)
        assert torch.equal(idx.cpu(), si[:, :k]) and torch.equal(vals.cpu(), sv[:, :k])
    print(f"similarity sweep: worst rel err {worst:.2e}")


@pytest.mark.parametrize("dt", [0, 1])
def test_groupnorm_attention_random_sweep(gpu, dt):
    """Seeded draws of the fused GroupNorm(+SiLU +embedding +residual) and of the two attention cores at the shapes the
    U-Net uses them (up to 32x32 maps, 192-1536 channels, tens of samples) against float64 torch arithmetic."""
    hip = gpu
    rng = random.Random(4321 + dt)
    g = torch.Generator(device="cuda").manual_seed(17 + dt)
    rn = lambda *s: torch.randn(*s, generator=g, device="cuda")
    q = lambda t: t.to(hip.torch_dtype(dt)).double()
    tol = 1e-5 if dt == 0 else 1e-2
    for it in range(30):
        C = rng.choice([64, 128, 192, 384, 768, 1536])
        G = rng.choice([1, 8])
        h, w = rng.choice([(32, 32), (16, 16), (8, 8), (4, 4), (6, 10)])
        n = rng.choice([1, 3, 16]) if C * h * w > 200000 else rng.choice([2, 33, 64])
        act, use_emb, use_res = rng.random() < 0.7, rng.random() < 0.5, rng.random() < 0.5
        x = rn(n, C, h, w) * 2 + 0.3
        ga, be = rn(C), rn(C)
        emb = rn(n, C) if use_emb else None
        rs = rn(n, C, h, w) if use_res else None
        y = hip.op_group_norm(dt, hip.to_nhwc(x, dt), ga, be, G, act_silu=act, emb=emb, resid=None if rs is None else hip.to_nhwc(rs, dt))
        ref = F.group_norm(q(x), G, ga.double(), be.double())
        if act:
            ref = F.silu(ref)
        if emb is not None:
            ref = ref + emb.double()[:, :, None, None]
        if rs is not None:
            ref = ref + q(rs)
        e = rel(hip.to_nchw(y, dt).double(), ref)
        assert e < tol, ("gn", it, C, G, h, w, n, act, use_emb, use_res, e)
    for it in range(12):
        h, w = rng.choice([(32, 32), (16, 16), (8, 8), (4, 4), (5, 7)])
        n = rng.choice([1, 4, 9])
        qkv = rn(n, 384, h, w)
        qq, kk, vv = (t.reshape(n, 4, 32, h * w) for t in q(qkv).chunk(3, 1))
        ctx = torch.einsum("bhdn,bhen->bhde", kk.softmax(-1), vv)
        o = torch.einsum("bhde,bhdn->bhen", ctx, qq.softmax(-2) * 32 ** -0.5).reshape(n, 128, h, w)
        y = hip.op_linear_attention(dt, hip.to_nhwc(qkv, dt))
        e = rel(hip.to_nchw(y, dt).double(), o)
        assert e < (2e-5 if dt == 0 else 2e-2), ("linattn", it, h, w, n, e)
        if h * w <= 64:
            sim = torch.einsum("bhdi,bhdj->bhij", qq * 32 ** -0.5, kk).softmax(-1)
            o = torch.einsum("bhij,bhdj->bhid", sim, vv).permute(0, 1, 3, 2).reshape(n, 128, h, w)
            y = hip.op_linear_attention(dt, hip.to_nhwc(qkv, dt), full=True)
            e = rel(hip.to_nchw(y, dt).double(), o)
            assert e < (2e-5 if dt == 0 else 2e-2), ("attn", it, h, w, n, e)
