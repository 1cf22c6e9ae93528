"""Numeric study for VERDICT r4 item 1 (CPU, before any kernel): what does a SPLIT-PRECISION 3x3 convolution cost in accuracy on the
real U-Net and the real scoring step?      python tests/split_error_study.py [--templates 32] [--schemes ...]

(Lives under tests/ because it drives the CPU oracle, which only tests/, smoke() and bench.py's cpu_baseline leg may import.)

Background.  north_star asks for similarity scores within 1e-4 (relative) of the reference's fp32 path
(/root/reference/src/model/model.py:260-262) and a bit-exact argmax.  Two modes bracket it today: `f16` (one MFMA pass per product,
8.7e-4) and `bf16x3` (f32 storage, three bf16 MFMA passes over (hi, lo) operand splits, 5.8e-6).  The cross terms of a hi / lo split
carry <= 2^-11 of the result, so they do not need 11-bit operands.  The scheme studied here, for every 3x3 convolution the
tap-resident kernel runs (94 % of the U-Net's work; all other launches keep the bf16x3 arithmetic = f32 here):

    a = a_hi + a_lo,  a_hi = f16(a)          w = w_hi + w_lo,  w_hi = f16(w)            (activations stay f32 in HBM)
    out = a_hi . w_hi                         on v_mfma_f32_32x32x16_f16                  1   pass
        + q8(a_lo) . q8(w)  +  q8(a) . q8(w_lo)   as ONE v_mfma_scale_f32_32x32x64_f8f6f4:    0.5 pass (fp8 operands: 2 x the f16 rate)
          K = [a_lo | a] x [w ; w_lo] over the same 32 channels, the E8M0 block scales of the instruction undo the power-of-two
          pre-scaling that moves the lo parts into the fp8 range

q8 is OCP e4m3 or e5m2 after a power-of-two pre-scale (a fixed one for the activations, one per layer for the weights, both undone by
the instruction's block scale), saturating.  The script replaces the oracle's 3x3 convolutions by an emulation of each scheme
(operands rounded as the kernel would round them, products and sums in f32 -- the MFMA accumulates exact products in f32), runs
BASELINE configs[1] (one 256 x 256 query, the first `--templates` of the benchmark's template poses, full-size U-Net at the
benchmark's weights) and reports the score error exactly as bench.py does:  max |score - score_f32| / max |score_f32|,  plus top-5
equality.  Gate: score error <= 5e-5 and top-5 equal.
"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as TF

from oracle import nope_ref as R
from nope_amd.harness import synthetic_batch
from nope_amd.weights import synth_tensor

E4M3, E5M2 = torch.float8_e4m3fn, torch.float8_e5m2
F8MAX = {E4M3: 448.0, E5M2: 57344.0}


def q8(x, fmt, log2_scale):
    """fp8 value of x * 2^log2_scale (round to nearest even, saturating), scaled back: what the MFMA sees after its block scale."""
    s = 2.0 ** log2_scale
    return (x * s).clamp(-F8MAX[fmt], F8MAX[fmt]).to(fmt).to(torch.float32) / s


def layer_log2_scale(w, fmt):
    """power of two that puts max |w| just below the format's largest value (chosen once per layer at pack time)"""
    m = float(w.abs().max())
    return int(torch.floor(torch.log2(torch.tensor(F8MAX[fmt] / m)))) if m > 0 else 0


class Scheme:
    def __init__(self, name):
        self.name = name
        self.cache = {}

    def conv3(self, a, w, b):
        raise NotImplementedError

    def pack(self, w):          # per-layer weight operands, cached (the pack kernel runs once per checkpoint)
        k = w.data_ptr()
        if k not in self.cache:
            self.cache[k] = self._pack(w)
        return self.cache[k]


class Plain(Scheme):            # f32 (the parity mode)
    def conv3(self, a, w, b):
        return TF.conv2d(a, w, b, padding=1)


class F16(Scheme):              # today's f16 mode restricted to these launches: one pass, both operands rounded to f16
    def _pack(self, w):
        return w.half().float()

    def conv3(self, a, w, b):
        return TF.conv2d(a.half().float(), self.pack(w), b, padding=1)


class Bf16x3(Scheme):           # today's bf16x3: (lo, hi) + (hi, lo) + (hi, hi) on bf16 splits
    def _pack(self, w):
        hi = w.bfloat16().float()
        return hi, (w - hi).bfloat16().float()

    def conv3(self, a, w, b):
        wh, wl = self.pack(w)
        ah = a.bfloat16().float()
        al = (a - ah).bfloat16().float()
        return TF.conv2d(al, wh, None, padding=1) + TF.conv2d(ah, wl, None, padding=1) + TF.conv2d(ah, wh, b, padding=1)


class F16Cross8(Scheme):
    """f16 hi x hi + fp8 cross terms.  fa / fw: formats of the (activation, weight) operands of the cross MFMA; sa_lo / sa: log2 pre-scales
    of a_lo and a (fixed); the weight pre-scales are per layer."""

    def __init__(self, name, fa_lo=E4M3, fa=E4M3, fw=E4M3, fw_lo=E4M3, sa_lo=8, sa=0, drop=()):
        super().__init__(name)
        self.fa_lo, self.fa, self.fw, self.fw_lo, self.sa_lo, self.sa, self.drop = fa_lo, fa, fw, fw_lo, sa_lo, sa, drop

    def _pack(self, w):
        hi = w.half().float()
        lo = w - hi
        return hi, q8(w, self.fw, layer_log2_scale(w, self.fw)), q8(lo, self.fw_lo, layer_log2_scale(lo, self.fw_lo))

    def conv3(self, a, w, b, stride=1, padding=1):
        wh, w8, wl8 = self.pack(w)
        ah = a.half().float()
        out = TF.conv2d(ah, wh, b, stride, padding)
        if "alo" not in self.drop:
            out = out + TF.conv2d(q8(a - ah, self.fa_lo, self.sa_lo), w8, None, stride, padding)
        if "wlo" not in self.drop:
            out = out + TF.conv2d(q8(a, self.fa, self.sa), wl8, None, stride, padding)
        return out


class A16W(Scheme):
    """activation operand = f16(a) only; weights hi + lo with lo in `lo_fmt` (None: exact weights; "f16": f16 lo; an fp8 format: per-layer pre-scale)"""

    def __init__(self, name, lo_fmt=None, a8=None):
        super().__init__(name)
        self.lo_fmt, self.a8 = lo_fmt, a8

    def _pack(self, w):
        hi = w.half().float()
        lo = w - hi
        if self.lo_fmt is None:
            return hi, lo
        if self.lo_fmt == "f16":
            return hi, (lo * 2048.0).half().float() / 2048.0
        return hi, q8(lo, self.lo_fmt, layer_log2_scale(lo, self.lo_fmt))

    def conv3(self, a, w, b):
        wh, wl = self.pack(w)
        ah = a.half().float()
        al = ah if self.a8 is None else q8(ah, self.a8, 0)
        return TF.conv2d(ah, wh, b, padding=1) + TF.conv2d(al, wl, None, padding=1)


SCHEMES = {
    "a16.w32": lambda: A16W("a16.w32"),
    "a16.(w16+w16)": lambda: A16W("a16.(w16+w16)", "f16"),
    "a16.w16+a8.w8": lambda: A16W("a16.w16+a8.w8", E4M3, E4M3),
    "a16.w16+a8(e5m2).w8": lambda: A16W("a16.w16+a8(e5m2).w8", E4M3, E5M2),
    "f32": lambda: Plain("f32"),
    "f16": lambda: F16("f16"),
    "bf16x3": lambda: Bf16x3("bf16x3"),
    # the candidates
    "f16+e4m3": lambda: F16Cross8("f16+e4m3"),
    "f16+e4m3(alo:2^9,a:2^-2)": lambda: F16Cross8("f16+e4m3(alo:2^9,a:2^-2)", sa_lo=9, sa=-2),
    # one clamp for all three conversions (|a| <= 2^14 / 2^15 / 2^16 first): a * 2^-6 / 2^-7 / 2^-8 and a_lo * 2^5 / 2^4 / 2^3 cannot overflow e4m3 then
    "f16+e4m3(alo:2^5,a:2^-6)": lambda: F16Cross8("f16+e4m3(alo:2^5,a:2^-6)", sa_lo=5, sa=-6),
    "f16+e4m3(alo:2^4,a:2^-7)": lambda: F16Cross8("f16+e4m3(alo:2^4,a:2^-7)", sa_lo=4, sa=-7),
    "f16+e4m3(alo:2^3,a:2^-8)": lambda: F16Cross8("f16+e4m3(alo:2^3,a:2^-8)", sa_lo=3, sa=-8),
    "f16+e4m3(alo:2^7,a:2^-4)": lambda: F16Cross8("f16+e4m3(alo:2^7,a:2^-4)", sa_lo=7, sa=-4),
    "f16+e4m3(alo:2^11)": lambda: F16Cross8("f16+e4m3(alo:2^11)", sa_lo=11),
    "f16+e4m3(alo:2^5)": lambda: F16Cross8("f16+e4m3(alo:2^5)", sa_lo=5),
    "f16+e5m2(a),e4m3": lambda: F16Cross8("f16+e5m2(a),e4m3", fa=E5M2),
    "f16+e5m2": lambda: F16Cross8("f16+e5m2", E5M2, E5M2, E5M2, E5M2, sa_lo=11),
    # ablations: which cross term matters
    "f16+only a.w_lo": lambda: F16Cross8("f16+only a.w_lo", drop=("alo",)),
    "f16+only a_lo.w": lambda: F16Cross8("f16+only a_lo.w", drop=("wlo",)),
}


class FShim:
    """Stands in for torch.nn.functional inside the oracle.  3x3 stride-1 convolutions the tap-resident kernel would run go through
    `scheme`; `store` (None = f32) is the STORAGE type of every activation tensor: each consumer (conv, norm, linear) sees its input rounded
    to it and each producer's output is rounded to it (a fused chain is rounded once too often -- a slight over-estimate); `other` is the
    operand type of the remaining convs / linears (None = f32, which is what bf16x3 delivers to ~1e-6)."""

    def __init__(self, scheme, store=None, other=None, gn_in=None):
        self.scheme, self.store, self.other, self.gn_in = scheme, store, other, gn_in
        self.n3 = 0
        self.wcache = {}

    def __getattr__(self, k):
        return getattr(TF, k)

    def st(self, x):
        return x if self.store is None else x.to(self.store).float()

    def wq(self, w):
        if self.other is None or self.other == "a16":
            return w
        k = w.data_ptr()
        if k not in self.wcache:
            self.wcache[k] = w.to(self.other).float()
        return self.wcache[k]

    def conv2d(self, x, w, b=None, stride=1, padding=0, *a, **kw):
        x = self.st(x)
        if w.shape[-1] == 3 and stride == 1 and padding == 1 and w.shape[1] % 32 == 0:
            self.n3 += 1
            return self.st(self.scheme.conv3(x, w, b))
        if self.other == "x2both":          # the full two-term tile on the other convs whatever the 3x3 scheme drops
            if w.shape[1] % 32 == 0:
                if not hasattr(self, "_both"):
                    self._both = F16Cross8("both", sa_lo=9, sa=-2)
                return self.st(self._both.conv3(x, w, b, stride, padding))
            return self.st(TF.conv2d(x, w, b, stride, padding, *a, **kw))
        if self.other == "x2":              # the same split-precision arithmetic on every other conv whose K the MX MFMA can tile
            if w.shape[1] % 32 == 0:
                return self.st(self.scheme.conv3(x, w, b, stride, padding))
            return self.st(TF.conv2d(x, w, b, stride, padding, *a, **kw))
        if self.other is not None:
            x = x.to(H if self.other == "a16" else self.other).float()
        return self.st(TF.conv2d(x, self.wq(w), b, stride, padding, *a, **kw))

    def group_norm(self, x, *a, **kw):
        G = a[0] if a else kw["num_groups"]
        if self.gn_in is not None and G > 1:
            # round 6 question: ONLY the conv -> GroupNorm(8) intermediate stored in 16 bits (written once by the conv epilogue, read once by
            # the apply pass, then normalised).  The statistics come from the conv's f32 accumulators (as the fused epilogue statistics
            # do), the apply pass reads the rounded tensor.
            w = a[1] if len(a) > 1 else kw.get("weight")
            bb = a[2] if len(a) > 2 else kw.get("bias")
            eps = a[3] if len(a) > 3 else kw.get("eps", 1e-5)
            n, c = x.shape[:2]
            xg = x.reshape(n, G, -1)
            mean = xg.mean(-1, keepdim=True)
            var = xg.var(-1, unbiased=False, keepdim=True)
            y = ((x.to(self.gn_in).float().reshape(n, G, -1) - mean) * torch.rsqrt(var + eps)).reshape(x.shape)
            return y * w.view(1, c, 1, 1) + bb.view(1, c, 1, 1)
        return self.st(TF.group_norm(self.st(x), *a, **kw))

    def batch_norm(self, x, *a, **kw):      # (encoder: folded into the conv in the library -- no rounding of its own)
        return TF.batch_norm(x, *a, **kw)

    def linear(self, x, w, b=None):
        return TF.linear(x, w, b)           # pose embedding linears run in f32 in every mode


# name -> (3x3 scheme, U-Net storage, other-conv operand type, encoder storage / operand type, bank + query storage for the scoring kernel)
H = torch.float16
CONFIGS = {
    "f32": ("f32", None, None, None, None),
    # today's f16 mode, emulated whole: f16 storage everywhere, f16 operands everywhere, f16 encoder, f16 bank
    "f16 mode (all f16)": ("f16", H, H, H, H),
    # ... and which part of it costs the accuracy
    "f16 mode, f32 encoder": ("f16", H, H, None, H),
    "f16 mode, f32 bank+query": ("f16", H, H, H, None),
    "f16 mode, f32 storage in U-Net": ("f16", None, H, H, H),
    "only: f16 encoder": ("f32", None, None, H, None),
    "only: f16 bank+query": ("f32", None, None, None, H),
    "only: f16 storage in U-Net": ("f32", H, None, None, None),
    "only: f16 operands, 3x3": ("f16", None, None, None, None),
    "only: f16 operands, other convs": ("f32", None, H, None, None),
    # hypothesis: the WEIGHT rounding (the same error at every pixel of every template: coherent) is what costs the f16 mode its accuracy
    "f16 mode, exact weights (U-Net+enc)": ("a16.w32", H, "a16", "a16", H),
    "f16 mode, exact weights (U-Net), f32 encoder": ("a16.w32", H, "a16", None, H),
    "f16 mode, 3x3 w16+w16, others exact": ("a16.(w16+w16)", H, "a16", "a16", H),
    "f16 mode, 3x3 w16 + a8.w8, others exact": ("a16.w16+a8.w8", H, "a16", "a16", H),
    "f16 mode, 3x3 w16 + a8(e5m2).w8, others exact": ("a16.w16+a8(e5m2).w8", H, "a16", "a16", H),
    "f16 mode, 3x3 exact w, others f16 w": ("a16.w32", H, H, H, H),
    # today's bf16x3 on the 3x3 launches (everything else f32 = what bf16x3 delivers there)
    "bf16x3 (3x3 only)": ("bf16x3", None, None, None, None),
    # candidates: f32 storage, split-precision 3x3 launches
    "f16+e4m3": ("f16+e4m3", None, None, None, None),
    "f16+e4m3(alo:2^9,a:2^-2)": ("f16+e4m3(alo:2^9,a:2^-2)", None, None, None, None),
    "f16+e4m3(alo:2^9,a:2^-2), f16 bank": ("f16+e4m3(alo:2^9,a:2^-2)", None, None, None, H),
    # ... and the remaining convs (1x1, up / down-sampling: bf16x3 today) with f16-rounded activation operands against exact weights: would a
    # two-pass a16 x (w_hi + w_lo) tile do for them?
    "f16+e4m3, other convs a16 x exact w": ("f16+e4m3(alo:2^9,a:2^-2)", None, "a16", None, None),
    # ... or the full split-precision tile on them too (every conv with Cin % 32 == 0)
    "f16+e4m3 on every conv": ("f16+e4m3(alo:2^9,a:2^-2)", None, "x2", None, None),
    # round 6: the timed mode as built (the tile on every eligible conv) + ONLY the conv -> GroupNorm(8) intermediate in 16 bits
    "f16+e4m3 on every conv, f16 conv->GN(8) intermediate": ("f16+e4m3(alo:2^9,a:2^-2)", None, "x2", None, None, H),
    "f16+e4m3 on every conv, bf16 conv->GN(8) intermediate": ("f16+e4m3(alo:2^9,a:2^-2)", None, "x2", None, None, torch.bfloat16),
    "f32, f16 conv->GN(8) intermediate": ("f32", None, None, None, None, H),
    # round 6 sizing: ONE cross term (a . w_lo) on the 3x3 launches -- the tap-resident kernel could then pair two taps in one K = 64 MX
    # instruction (5 instead of 9 per channel chunk) -- the per-tap launches keeping both
    "f16+only a.w_lo on 3x3, both terms on the other convs": ("f16+only a.w_lo", None, "x2both", None, None),
    "f16+only a.w_lo on every conv": ("f16+only a.w_lo", None, "x2", None, None),
    "f16+e4m3(alo:2^7,a:2^-4)": ("f16+e4m3(alo:2^7,a:2^-4)", None, None, None, None),
    "f16+e4m3(alo:2^5,a:2^-6)": ("f16+e4m3(alo:2^5,a:2^-6)", None, None, None, None),
    "f16+e4m3(alo:2^4,a:2^-7)": ("f16+e4m3(alo:2^4,a:2^-7)", None, None, None, None),
    "f16+e4m3(alo:2^3,a:2^-8)": ("f16+e4m3(alo:2^3,a:2^-8)", None, None, None, None),
    "f16+e4m3(alo:2^11)": ("f16+e4m3(alo:2^11)", None, None, None, None),
    "f16+e4m3(alo:2^5)": ("f16+e4m3(alo:2^5)", None, None, None, None),
    "f16+e5m2(a),e4m3": ("f16+e5m2(a),e4m3", None, None, None, None),
    "f16+e5m2": ("f16+e5m2", None, None, None, None),
    "f16+only a.w_lo": ("f16+only a.w_lo", None, None, None, None),
    "f16+only a_lo.w": ("f16+only a_lo.w", None, None, None, None),
}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--templates", type=int, default=32)
    ap.add_argument("--configs", default=";".join(CONFIGS), help="';'-separated names")
    ap.add_argument("--seed", type=int, default=2022)
    ap.add_argument("--chunk", type=int, default=8)
    a = ap.parse_args()
    from nope_amd.u_net import UNet
    from nope_amd.encoder import FeatureExtractor
    from nope_amd.harness import TEMPLATE_BASE
    cfg = TEMPLATE_BASE["u_net"]
    enc = FeatureExtractor(**cfg["encoder"], compute_dtype="f32")
    esd = {k: synth_tensor(a.seed, "encoder." + k, tuple(v.shape)) for k, v in enc.state_dict().items()}
    unet = UNet(u_net_dim=cfg["u_net_dim"], rot_representation_dim=6, encoder=enc, pose_mlp_name=cfg["pose_mlp_name"], compute_dtype="f32")
    sd = {k: synth_tensor(a.seed, k, tuple(v.shape)) for k, v in unet.state_dict().items() if not k.startswith("encoder.")}
    del unet, enc
    b = synthetic_batch(1, 512, 256, seed=a.seed)          # bench.py's configs[1] batch
    poses = b["all_relativeR"][:, :a.templates]
    print(f"BASELINE configs[1]: one 256 x 256 query, the first {a.templates} of the 512 template poses, full-size U-Net + ResNet-50 encoder at the benchmark's "
          f"synthetic weights (seed {a.seed}); error = max |score - score_f32| / max |score_f32| as bench.py's parity record", flush=True)

    results = {}
    enc_cache = {}
    for name in a.configs.split(";"):
        sname, store, other, enc_t, bank_t = CONFIGS[name][:5]
        gn_in = CONFIGS[name][5] if len(CONFIGS[name]) > 5 else None
        t0 = time.time()
        with torch.no_grad():
            if enc_t not in enc_cache:
                R.F = FShim(Plain("f32"), store=H if enc_t == "a16" else enc_t, other=enc_t)
                enc_cache[enc_t] = (R.encode_image(esd, b["reference"]), R.encode_image(esd, b["query"]))
            ref_feat, qry_feat = enc_cache[enc_t]
            shim = FShim(SCHEMES[sname](), store=store, other=other, gn_in=gn_in)
            R.F = shim
            bank = R.generate_templates(sd, ref_feat, poses, chunk=a.chunk)
            R.F = TF
            if bank_t is not None:
                bank, qry_feat = bank.to(bank_t).float(), qry_feat.to(bank_t).float()
            sim, idx = R.retrieval(qry_feat, bank)
        results[name] = (bank, sim, idx)
        line = f"{name:48s} {time.time() - t0:6.1f} s"
        if "f32" in results:
            bank0, sim0, idx0 = results["f32"]
            scale = float(sim0.abs().max())
            err = float((sim - sim0).abs().max()) / scale
            merr = float((bank - bank0).abs().max() / bank0.abs().max())
            gap = sim0.topk(2, dim=1).values
            margin = float((gap[:, 0] - gap[:, 1]).min()) / scale
            line += f"  score_rel_err {err:.3e}  map_rel_err {merr:.3e}  top5_equal {bool((idx == idx0).all())}  (f32 top-1 gap {margin:.2e})"
        print(line, flush=True)


if __name__ == "__main__":
    main()
