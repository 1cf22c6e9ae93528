"""CPU oracle for the NOPE inference hot path  --  TEST INFRASTRUCTURE, NOT PRODUCT CODE.

A from-scratch, functional (no nn.Module) fp32 restatement of the reference's algorithm
for the path SURVEY.md §8 names.  Only `tests/`, `__graft_entry__.smoke()` and the
`cpu_baseline` leg of `bench.py` may import this package; `nope_amd/` never does.

Pinning status: the reference ships no golden vectors / tests, so the
oracle is pinned against OUTPUTS OF THE REFERENCE ITSELF, run in the build container by
`tests/golden/make_golden.py` (reference imported from /root/reference with non-arithmetic
dependencies stubbed) and committed under `tests/golden/*.npz`.
`tests/test_oracle_golden.py` checks every function below against those fixtures.

All `file:line` citations are relative to the reference tree (nv-nguyen/nope).
Weights are addressed by the reference's state-dict keys (SURVEY.md §8 "State-dict keys").

Tensors are torch CPU fp32 (fp64 optional through `dtype=`); this is a floating-point
path, so the restatement uses torch's CPU conv/matmul primitives and nothing else.
"""
from __future__ import annotations

import math
from typing import Dict, Optional, Sequence, Tuple

import torch
import torch.nn.functional as F

Tensor = torch.Tensor
SD = Dict[str, Tensor]

HEADS = 4       # model_utils.py:368,394 (Attention / LinearAttention defaults)
DIM_HEAD = 32


# --------------------------------------------------------------------------------------
# building blocks  (src/model/u_net/denoising_diffusion_pytorch/model_utils.py)
# --------------------------------------------------------------------------------------
def block(x: Tensor, sd: SD, p: str, groups: int) -> Tensor:
    """`Block.forward` without scale_shift: conv3x3(pad 1) -> GroupNorm(groups) -> SiLU.
    model_utils.py:237-253 (the default U-Net never passes scale_shift, :272)."""
    x = F.conv2d(x, sd[p + "proj.weight"], sd[p + "proj.bias"], padding=1)
    x = F.group_norm(x, groups, sd[p + "norm.weight"], sd[p + "norm.bias"], eps=1e-5)
    return F.silu(x)


def resnet_block(x: Tensor, emb: Optional[Tensor], sd: SD, p: str, groups: int) -> Tensor:
    """`ResnetBlock.forward`, model_utils.py:271-279.
    h = block1(x); h = Linear(SiLU(emb))[:, :, None, None] + h; h = block2(h);
    return h + res_conv(x)   (res_conv = 1x1 conv iff dim != dim_out, :269)."""
    h = block(x, sd, p + "block1.", groups)
    if emb is not None and (p + "mlp.1.weight") in sd:
        e = F.linear(F.silu(emb), sd[p + "mlp.1.weight"], sd[p + "mlp.1.bias"])
        h = e[:, :, None, None] + h
    h = block(h, sd, p + "block2.", groups)
    if (p + "res_conv.weight") in sd:
        res = F.conv2d(x, sd[p + "res_conv.weight"], sd[p + "res_conv.bias"])
    else:
        res = x
    return h + res


def linear_attention(x: Tensor, sd: SD, p: str) -> Tensor:
    """`LinearAttention.forward`, model_utils.py:403-418 (heads 4 x dim_head 32).
    q softmax over the head dim, k softmax over the n = h*w pixels, q *= 32^-0.5,
    ctx[d,e] = sum_n k[d,n] v[e,n]; out[e,n] = sum_d ctx[d,e] q[d,n];
    then to_out = conv1x1 + GroupNorm(1, dim) (:401)."""
    b, c, h, w = x.shape
    qkv = F.conv2d(x, sd[p + "to_qkv.weight"])
    q, k, v = (t.reshape(b, HEADS, DIM_HEAD, h * w) for t in qkv.chunk(3, dim=1))
    q = q.softmax(dim=-2) * (DIM_HEAD ** -0.5)
    k = k.softmax(dim=-1)
    ctx = torch.einsum("bhdn,bhen->bhde", k, v)
    out = torch.einsum("bhde,bhdn->bhen", ctx, q).reshape(b, HEADS * DIM_HEAD, h, w)
    out = F.conv2d(out, sd[p + "to_out.0.weight"], sd[p + "to_out.0.bias"])
    return F.group_norm(out, 1, sd[p + "to_out.1.weight"], sd[p + "to_out.1.bias"], eps=1e-5)


def attention(x: Tensor, sd: SD, p: str) -> Tensor:
    """`Attention.forward`, model_utils.py:376-390: full softmax attention over the
    h*w tokens, scale 32^-0.5 applied to q, max-subtracted softmax."""
    b, c, h, w = x.shape
    qkv = F.conv2d(x, sd[p + "to_qkv.weight"])
    q, k, v = (t.reshape(b, HEADS, DIM_HEAD, h * w) for t in qkv.chunk(3, dim=1))
    q = q * (DIM_HEAD ** -0.5)
    sim = torch.einsum("bhdi,bhdj->bhij", q, k)
    sim = sim - sim.amax(dim=-1, keepdim=True)
    attn = sim.softmax(dim=-1)
    out = torch.einsum("bhij,bhdj->bhid", attn, v)                     # (b, heads, n, d)
    out = out.permute(0, 1, 3, 2).reshape(b, HEADS * DIM_HEAD, h, w)   # "b h (x y) d -> b (h d) x y"
    return F.conv2d(out, sd[p + "to_out.weight"], sd[p + "to_out.bias"])


def residual_prenorm(x: Tensor, sd: SD, p: str, fn) -> Tensor:
    """`Residual(PreNorm(dim, fn))`, model_utils.py:198-204, 226-234:  fn(GroupNorm(1)(x)) + x."""
    y = F.group_norm(x, 1, sd[p + "fn.norm.weight"], sd[p + "fn.norm.bias"], eps=1e-5)
    return fn(y, sd, p + "fn.fn.") + x


def pixel_unshuffle_c_p1_p2(x: Tensor) -> Tensor:
    """einops `Rearrange("b c (h p1) (w p2) -> b (c p1 p2) h w", p1=2, p2=2)`, model_utils.py:170."""
    b, c, hh, ww = x.shape
    x = x.reshape(b, c, hh // 2, 2, ww // 2, 2)          # b c h p1 w p2
    return x.permute(0, 1, 3, 5, 2, 4).reshape(b, c * 4, hh // 2, ww // 2)


def hard_downsample(x: Tensor, sd: SD, p: str) -> Tensor:
    """`HardDownsample`, model_utils.py:168-172: space-to-depth 2x2 then conv1x1 (key `<p>1.*`).  With `use_hard_up_down=False`
    (u_net.py:54-59) the slot holds `Downsample` = Conv2d(4, stride 2, pad 1) instead (model_utils.py:129-136, key `<p>weight`)."""
    if p + "weight" in sd:
        return F.conv2d(x, sd[p + "weight"], sd[p + "bias"], stride=2, padding=1)
    return F.conv2d(pixel_unshuffle_c_p1_p2(x), sd[p + "1.weight"], sd[p + "1.bias"])


def hard_upsample(x: Tensor, sd: SD, p: str) -> Tensor:
    """`HardUpsample`, model_utils.py:161-165: nearest x2 then conv3x3 pad 1 (key `<p>1.*`).  With `use_hard_up_down=False` the
    slot holds `Upsample` = ConvTranspose2d(4, stride 2, pad 1) instead (model_utils.py:119-126, key `<p>weight`)."""
    if p + "weight" in sd:
        return F.conv_transpose2d(x, sd[p + "weight"], sd[p + "bias"], stride=2, padding=1)
    x = x.repeat_interleave(2, dim=2).repeat_interleave(2, dim=3)
    return F.conv2d(x, sd[p + "1.weight"], sd[p + "1.bias"], padding=1)


# --------------------------------------------------------------------------------------
# U-Net forward  (src/model/u_net/denoising_diffusion_pytorch/u_net.py:160-198)
# --------------------------------------------------------------------------------------
def pose_mlp(pose: Tensor, sd: SD) -> Tensor:
    """`pose_mlp`, u_net.py:61-76: "single_layer" = Linear(6, 4*dim); "two_layers" adds
    GELU + Linear; "posEncoding" = the parameter-free SinusoidalPosEmb(dim = 4*u_net_dim / 6) of src/model/utils.py:36-51
    (per pose component: frequencies exp(-j * ln(1e4) / (half - 1)), all sines then all cosines).  The variant is
    inferred from the keys present."""
    if "pose_mlp.0.weight" not in sd:
        classes = sd["downs.0.0.mlp.1.weight"].shape[1]
        half = classes // pose.shape[1] // 2
        freq = torch.exp(torch.arange(half) * -(math.log(10000) / (half - 1)))
        emb = (pose[:, :, None] * freq[None, None, :]).reshape(pose.shape[0], -1)
        return torch.cat((emb.sin(), emb.cos()), dim=-1)
    c = F.linear(pose, sd["pose_mlp.0.weight"], sd["pose_mlp.0.bias"])
    if "pose_mlp.2.weight" in sd:
        c = F.linear(F.gelu(c), sd["pose_mlp.2.weight"], sd["pose_mlp.2.bias"])
    return c


def unet_forward(sd: SD, x: Tensor, pose: Tensor, groups: int = 8, n_levels: int = 4) -> Tensor:
    """`UNet.forward(x, pose)`.  `sd` holds the U-Net's own keys (no `u_net.` prefix,
    `encoder.*` ignored).  Exact op order of u_net.py:160-198, including the mid block
    applied twice with the same weights (:177-183) and `final_conv.0` called without the
    embedding (:154-157,197)."""
    x = F.conv2d(x, sd["init_conv.weight"], sd["init_conv.bias"], padding=1)
    r = x.clone()
    c = pose_mlp(pose, sd)
    hs = []
    for l in range(n_levels):
        p = f"downs.{l}."
        x = resnet_block(x, c, sd, p + "0.", groups)
        hs.append(x)
        x = resnet_block(x, c, sd, p + "1.", groups)
        x = residual_prenorm(x, sd, p + "2.", linear_attention)
        hs.append(x)
        if l < n_levels - 1:
            x = hard_downsample(x, sd, p + "3.")
        else:
            x = F.conv2d(x, sd[p + "3.weight"], sd[p + "3.bias"], padding=1)
    for _ in range(2):                                   # u_net.py:177-183
        x = resnet_block(x, c, sd, "mid_block1.", groups)
        x = residual_prenorm(x, sd, "mid_attn.", attention)
        x = resnet_block(x, c, sd, "mid_block2.", groups)
    for l in range(n_levels):
        p = f"ups.{l}."
        x = torch.cat((x, hs.pop()), dim=1)
        x = resnet_block(x, c, sd, p + "0.", groups)
        x = torch.cat((x, hs.pop()), dim=1)
        x = resnet_block(x, c, sd, p + "1.", groups)
        x = residual_prenorm(x, sd, p + "2.", linear_attention)
        if l < n_levels - 1:
            x = hard_upsample(x, sd, p + "3.")
        else:
            x = F.conv2d(x, sd[p + "3.weight"], sd[p + "3.bias"], padding=1)
    x = torch.cat((x, r), dim=1)
    x = resnet_block(x, c, sd, "final_res_block.", groups)
    x = resnet_block(x, None, sd, "final_conv.0.", groups)
    return F.conv2d(x, sd["final_conv.1.weight"], sd["final_conv.1.bias"])


# --------------------------------------------------------------------------------------
# template encoder  (src/model/encoder/template.py:24-53, resnet.py:55-152)
# --------------------------------------------------------------------------------------
_R50_LAYERS = (3, 4, 6, 3)
_R50_STRIDES = (1, 2, 2, 1)      # resnet.py:102-105 (layer4 stride 1); conv1 stride 2 (:98)


def _bn_eval(x: Tensor, sd: SD, p: str) -> Tensor:
    return F.batch_norm(x, sd[p + "running_mean"], sd[p + "running_var"], sd[p + "weight"],
                        sd[p + "bias"], training=False, eps=1e-5)


def _bottleneck(x: Tensor, sd: SD, p: str, stride: int) -> Tensor:
    """`Bottleneck.forward`, resnet.py:70-90."""
    out = F.relu(_bn_eval(F.conv2d(x, sd[p + "conv1.weight"]), sd, p + "bn1."))
    out = F.relu(_bn_eval(F.conv2d(out, sd[p + "conv2.weight"], stride=stride, padding=1), sd, p + "bn2."))
    out = _bn_eval(F.conv2d(out, sd[p + "conv3.weight"]), sd, p + "bn3.")
    if (p + "downsample.0.weight") in sd:
        x = _bn_eval(F.conv2d(x, sd[p + "downsample.0.weight"], stride=stride), sd, p + "downsample.1.")
    return F.relu(out + x)


def encode_image(sd: SD, image: Tensor, normalize: bool = False) -> Tensor:
    """`FeatureExtractor.encode_image`, template.py:47-53: ResNet-50 without maxpool /
    avgpool / fc (resnet.py:135-152, `use_avg_pooling_and_fc=False`) giving /8, then
    ReLU -> 1x1(2048->256) -> ReLU -> 1x1(256->descriptor) (template.py:34-39).
    `sd` holds the encoder's keys (`backbone.*`, `projector.*`)."""
    x = F.conv2d(image, sd["backbone.conv1.weight"], stride=2, padding=3)
    x = F.relu(_bn_eval(x, sd, "backbone.bn1."))
    for li, (nblk, stride) in enumerate(zip(_R50_LAYERS, _R50_STRIDES), start=1):
        for bi in range(nblk):
            x = _bottleneck(x, sd, f"backbone.layer{li}.{bi}.", stride if bi == 0 else 1)
    x = F.conv2d(F.relu(x), sd["projector.1.weight"])
    x = F.conv2d(F.relu(x), sd["projector.3.weight"])
    if normalize:
        x = F.normalize(x, dim=1)
    return x


# --------------------------------------------------------------------------------------
# task level  (src/model/model.py)
# --------------------------------------------------------------------------------------
def similarity_scores(query_feat: Tensor, bank: Tensor) -> Tensor:
    """The "l2" metric of `PoseConditional.retrieval`, model.py:257-262 (second witness:
    loss.py:118-143):  score[b,n] = - sum_{h,w} || (q[b] - t[b,n])**2 ||_2 over C.
    query_feat (B,C,h,w); bank (B,N,C,h,w) -> (B,N)."""
    d = (query_feat[:, None] - bank) ** 2
    d = torch.sqrt((d * d).sum(dim=2))
    return -d.sum(dim=3).sum(dim=2)


def topk_desc_lowest_index(scores: Tensor, k: int = 5) -> Tensor:
    """`similarity.topk(k=5, dim=1)` indices, model.py:265, with the build's tie rule
    (SURVEY.md §8 c3): descending score, ties -> lowest index (argmax semantics for k=1);
    NaN ranks above everything (torch.topk convention).  Returns int64 (B,k)."""
    s = scores.detach().to(torch.float64).clone()
    s[torch.isnan(s)] = float("inf")
    # stable sort on -s gives lowest index first among equals
    order = torch.sort(-s, dim=1, stable=True).indices
    return order[:, :k].contiguous()


def retrieval(query_feat: Tensor, bank: Tensor, k: int = 5) -> Tuple[Tensor, Tensor]:
    s = similarity_scores(query_feat, bank)
    return s, topk_desc_lowest_index(s, k)


def generate_templates(unet_sd: SD, reference_feat: Tensor, all_relativeR: Tensor,
                       groups: int = 8, chunk: int = 8) -> Tensor:
    """`PoseConditional.generate_templates`, model.py:193-252, with the loop-invariant
    `encode_image(reference)` hoisted (the reference recomputes it per template, :115 via
    :219 - same value every time).  reference_feat (B,C,h,w); all_relativeR (B,N,6)
    -> bank (B,N,C,h,w).  Hypotheses are evaluated `chunk` at a time; every op in the
    U-Net is per-sample, so batching does not change the arithmetic."""
    B, N = all_relativeR.shape[:2]
    outs = []
    for b in range(B):
        rows = []
        for s in range(0, N, chunk):
            pose = all_relativeR[b, s:s + chunk]
            x = reference_feat[b:b + 1].expand(pose.shape[0], -1, -1, -1)
            rows.append(unet_forward(unet_sd, x, pose, groups))
        outs.append(torch.cat(rows, 0))
    return torch.stack(outs, 0)


def forward_loss(unet_sd: SD, query_feat: Tensor, reference_feat: Tensor, relativeR: Tensor,
                 loss_type: str = "l1", groups: int = 8) -> Tensor:
    """`PoseConditional.forward`, model.py:106-111 with `compute_loss` :96-104:
    per-sample mean of |pred - gt| (l1) or (pred - gt)^2 (l2), then batch mean."""
    pred = unet_forward(unet_sd, reference_feat, relativeR, groups)
    d = (pred - query_feat).abs() if loss_type == "l1" else (pred - query_feat) ** 2
    return d.flatten(1).mean(dim=1).mean()


def geodesic_deg(predR: Tensor, gtR: Tensor) -> Tensor:
    """Restatement of the angle behind `GeodesicError` for non-symmetric objects
    (loss.py:14-22,80-95): acos((tr(R1 R2^T) - 1) / 2) in degrees.  pytorch3d's
    `so3_relative_angle(eps=1e-2)` is an un-vendored dependency -> parity unpinned
    (SURVEY.md §8 c4); used for harness plumbing only."""
    rel = predR.to(torch.float64) @ gtR.to(torch.float64).transpose(-1, -2)
    cos = ((rel.diagonal(dim1=-2, dim2=-1).sum(-1)) - 1.0) / 2.0
    return torch.rad2deg(torch.acos(cos.clamp(-1.0, 1.0)))


# --------------------------------------------------------------------------------------
# LDM cross-attention variant  (src/model/u_net/ldm/adapt_openaimodel.py:130-158)
# --------------------------------------------------------------------------------------
def _ldm_res(x: Tensor, emb: Tensor, sd: SD, p: str) -> Tensor:
    """ResBlock._forward, ldm/openaimodel.py:262-288 (no up/down).  `use_scale_shift_norm` (FiLM, :277-281) is read off the
    shape of emb_layers.1: twice the block's channels -> out_norm(h) * (1 + scale) + shift instead of h + emb_out."""
    h = F.conv2d(F.silu(F.group_norm(x, 32, sd[p + "in_layers.0.weight"], sd[p + "in_layers.0.bias"], 1e-5)),
                 sd[p + "in_layers.2.weight"], sd[p + "in_layers.2.bias"], padding=1)
    e = F.linear(F.silu(emb), sd[p + "emb_layers.1.weight"], sd[p + "emb_layers.1.bias"])
    if e.shape[1] == 2 * h.shape[1]:
        scale, shift = e[:, :, None, None].chunk(2, dim=1)
        h = F.group_norm(h, 32, sd[p + "out_layers.0.weight"], sd[p + "out_layers.0.bias"], 1e-5) * (1 + scale) + shift
    else:
        h = F.group_norm(h + e[:, :, None, None], 32, sd[p + "out_layers.0.weight"], sd[p + "out_layers.0.bias"], 1e-5)
    h = F.conv2d(F.silu(h), sd[p + "out_layers.3.weight"], sd[p + "out_layers.3.bias"], padding=1)
    if p + "skip_connection.weight" in sd:
        x = F.conv2d(x, sd[p + "skip_connection.weight"], sd[p + "skip_connection.bias"])
    return x + h


def _ldm_cross_attention(x: Tensor, ctx: Tensor, sd: SD, p: str, d_head: int = 32) -> Tensor:
    """CrossAttention.forward, ldm/attention.py:168-189.  x (B,N,C); ctx (B,J,Cc)."""
    B, N, C = x.shape
    h = C // d_head
    q, k, v = F.linear(x, sd[p + "to_q.weight"]), F.linear(ctx, sd[p + "to_k.weight"]), F.linear(ctx, sd[p + "to_v.weight"])
    split = lambda t: t.reshape(B, t.shape[1], h, d_head).permute(0, 2, 1, 3).reshape(B * h, t.shape[1], d_head)
    q, k, v = split(q), split(k), split(v)
    attn = (torch.einsum("bid,bjd->bij", q, k) * d_head ** -0.5).softmax(dim=-1)
    out = torch.einsum("bij,bjd->bid", attn, v).reshape(B, h, N, d_head).permute(0, 2, 1, 3).reshape(B, N, C)
    return F.linear(out, sd[p + "to_out.0.weight"], sd[p + "to_out.0.bias"])


def _ldm_transformer(x: Tensor, ctx: Tensor, sd: SD, p: str) -> Tensor:
    """SpatialTransformer.forward, ldm/attention.py:264-277, with its `depth` BasicTransformerBlocks (:192-212, :251-258) in sequence."""
    B, C, H, W = x.shape
    t = F.conv2d(F.group_norm(x, 32, sd[p + "norm.weight"], sd[p + "norm.bias"], 1e-6), sd[p + "proj_in.weight"], sd[p + "proj_in.bias"])
    t = t.reshape(B, C, H * W).permute(0, 2, 1)
    d = 0
    while f"{p}transformer_blocks.{d}.norm1.weight" in sd:
        b = f"{p}transformer_blocks.{d}."
        ln = lambda y, n: F.layer_norm(y, (C,), sd[b + n + ".weight"], sd[b + n + ".bias"], 1e-5)
        y = ln(t, "norm1")
        t = _ldm_cross_attention(y, y, sd, b + "attn1.") + t
        t = _ldm_cross_attention(ln(t, "norm2"), ctx, sd, b + "attn2.") + t
        a, gate = F.linear(ln(t, "norm3"), sd[b + "ff.net.0.proj.weight"], sd[b + "ff.net.0.proj.bias"]).chunk(2, dim=-1)
        t = F.linear(a * F.gelu(gate), sd[b + "ff.net.2.weight"], sd[b + "ff.net.2.bias"]) + t
        d += 1
    t = t.permute(0, 2, 1).reshape(B, C, H, W)
    return F.conv2d(t, sd[p + "proj_out.weight"], sd[p + "proj_out.bias"]) + x


def _ldm_block(h: Tensor, emb: Tensor, ctx: Tensor, sd: SD, p: str) -> Tensor:
    """TimestepEmbedSequential.forward, ldm/openaimodel.py:83-92: children by index; the kind is read off the keys."""
    i = 0
    while True:
        q = f"{p}{i}."
        if q + "in_layers.0.weight" in sd:
            h = _ldm_res(h, emb, sd, q)
        elif q + "proj_in.weight" in sd:
            h = _ldm_transformer(h, ctx, sd, q)
        elif q + "op.weight" in sd:                                   # Downsample, :143-174
            h = F.conv2d(h, sd[q + "op.weight"], sd[q + "op.bias"], stride=2, padding=1)
        elif q + "conv.weight" in sd:                                 # Upsample, :93-124
            h = F.conv2d(F.interpolate(h, scale_factor=2, mode="nearest"), sd[q + "conv.weight"], sd[q + "conv.bias"], padding=1)
        elif q + "weight" in sd:                                      # the input conv
            h = F.conv2d(h, sd[q + "weight"], sd[q + "bias"], padding=1)
        else:
            return h
        i += 1


def ldm_forward(sd: SD, x: Tensor, pose: Tensor) -> Tensor:
    """`UNetModelPose.forward(x, pose)`, ldm/adapt_openaimodel.py:130-158: context = pose_mlp(pose)[:, None]; the timestep
    embedding is zeros unless `pose_mlp_timesteps.*` is present (injecting_condition_twice)."""
    emb_dim = sd["input_blocks.1.0.emb_layers.1.weight"].shape[1]
    if "pose_mlp_timesteps.0.weight" in sd:
        emb = F.linear(pose, sd["pose_mlp_timesteps.0.weight"], sd["pose_mlp_timesteps.0.bias"])
    else:
        emb = torch.zeros(x.shape[0], emb_dim)
    ctx = pose_mlp(pose, sd).unsqueeze(1)
    hs = []
    h = x
    n_in = 1 + max(int(k.split(".")[1]) for k in sd if k.startswith("input_blocks."))
    for i in range(n_in):
        h = _ldm_block(h, emb, ctx, sd, f"input_blocks.{i}.")
        hs.append(h)
    h = _ldm_block(h, emb, ctx, sd, "middle_block.")
    n_out = 1 + max(int(k.split(".")[1]) for k in sd if k.startswith("output_blocks."))
    for i in range(n_out):
        h = torch.cat([h, hs.pop()], dim=1)
        h = _ldm_block(h, emb, ctx, sd, f"output_blocks.{i}.")
    return F.conv2d(F.silu(F.group_norm(h, 32, sd["out.0.weight"], sd["out.0.bias"], 1e-5)), sd["out.2.weight"], sd["out.2.bias"], padding=1)
