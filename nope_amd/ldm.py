"""`UNetModelPose` -- drop-in for the reference's LDM cross-attention U-Net variant
(src/model/u_net/ldm/adapt_openaimodel.py:14-158 over ldm/openaimodel.py:428-760 and ldm/attention.py:149-277), the
variant whose pose conditioning is cross-attention against `context = pose_mlp(pose).unsqueeze(1)`; executed by
libnope_hip.so (`nope_ldm_*`, csrc/ldm_runtime.hip).

Same constructor arguments as the reference class (the ones configs/model/vae_cin_ldm.yaml passes), same attributes
(`encoder`, `channels`, `name`), same `state_dict()` keys and shapes -- an LDM checkpoint loads with `load_state_dict`
-- and the same call `u_net(x, pose) -> pred`, so it plugs into `nope_amd.PoseConditional` exactly as `nope_amd.UNet` does
(`forward_hypotheses` is the batched form `generate_templates` uses).  The module tree only holds parameters.

Supported configuration: `use_spatial_transformer=True`, `transformer_depth=1`, `num_head_channels=32`,
`conv_resample=True`, `use_scale_shift_norm` (FiLM ResBlocks) on or off, no `resblock_updown`, `pose_mlp_name`
"single_layer" / "two_layers", `injecting_condition_twice` on or off; anything else raises NotImplementedError.
"""
from __future__ import annotations

from typing import Optional

import torch
from torch import nn

from . import hip
from .u_net import _Params, _slot


def _res_params(cin, cout, emb_dim, film=False):
    m = _Params()
    m.in_layers = _slot(nn.GroupNorm(32, cin), None, nn.Conv2d(cin, cout, 3, padding=1))            # openaimodel.py:214-218
    m.emb_layers = _slot(None, nn.Linear(emb_dim, 2 * cout if film else cout))                        # :233-239 (FiLM: scale | shift)
    m.out_layers = _slot(nn.GroupNorm(32, cout), None, None, nn.Conv2d(cout, cout, 3, padding=1))     # :248-255
    if cin != cout:
        m.skip_connection = nn.Conv2d(cin, cout, 1)                                                   # :257-264
    return m


def _cross_attention_params(query_dim, context_dim):
    m = _Params()                                                                                     # attention.py:149-166
    m.to_q = nn.Linear(query_dim, query_dim, bias=False)
    m.to_k = nn.Linear(context_dim, query_dim, bias=False)
    m.to_v = nn.Linear(context_dim, query_dim, bias=False)
    m.to_out = _slot(nn.Linear(query_dim, query_dim), None)
    return m


def _transformer_params(ch, context_dim, depth=1):
    def block():
        blk = _Params()                                                                               # attention.py:192-212
        blk.attn1 = _cross_attention_params(ch, ch)
        ff = _Params()
        geglu = _Params()
        geglu.proj = nn.Linear(ch, 8 * ch)                                                            # GEGLU(dim, 4*dim): proj to 2 x inner
        ff.net = _slot(geglu, None, nn.Linear(4 * ch, ch))
        blk.ff = ff
        blk.attn2 = _cross_attention_params(ch, context_dim)
        blk.norm1, blk.norm2, blk.norm3 = nn.LayerNorm(ch), nn.LayerNorm(ch), nn.LayerNorm(ch)
        return blk
    m = _Params()                                                                                     # attention.py:232-262
    m.norm = nn.GroupNorm(32, ch, eps=1e-6)
    m.proj_in = nn.Conv2d(ch, ch, 1)
    m.transformer_blocks = nn.ModuleList([block() for _ in range(depth)])
    m.proj_out = nn.Conv2d(ch, ch, 1)
    return m


class UNetModelPose(nn.Module):
    def __init__(self, injecting_condition_twice, pose_mlp_name, rot_representation_dim, encoder, image_size, in_channels,
                 model_channels, out_channels, num_res_blocks, attention_resolutions, dropout=0, channel_mult=(1, 2, 4, 8),
                 conv_resample=True, dims=2, num_classes=None, use_checkpoint=False, use_fp16=False, num_heads=-1,
                 num_head_channels=-1, num_heads_upsample=-1, use_scale_shift_norm=False, resblock_updown=False,
                 use_new_attention_order=False, use_spatial_transformer=False, transformer_depth=1, context_dim=None,
                 n_embed=None, legacy=True, compute_dtype="f32", **kwargs):
        super().__init__()
        if not use_spatial_transformer or transformer_depth < 1 or context_dim is None:
            raise NotImplementedError("only use_spatial_transformer=True (configs/model/vae_cin_ldm.yaml), transformer_depth >= 1")
        self.transformer_depth = int(transformer_depth)
        if num_head_channels != 32 or resblock_updown or not conv_resample or dims != 2 \
                or num_classes is not None or n_embed is not None:
            raise NotImplementedError("unsupported UNetModel option (see module docstring)")
        if pose_mlp_name not in ("single_layer", "two_layers"):
            raise NotImplementedError(f"pose_mlp_name={pose_mlp_name!r}")
        self.encoder = encoder
        self.channels = encoder.latent_dim
        self.name = encoder.name
        self.image_size, self.in_channels, self.model_channels, self.out_channels = image_size, in_channels, model_channels, out_channels
        self.num_res_blocks, self.channel_mult = num_res_blocks, tuple(channel_mult)
        self.attention_resolutions = tuple(attention_resolutions)
        self.context_dim, self.rot_representation_dim = context_dim, rot_representation_dim
        self.injecting_condition_twice = bool(injecting_condition_twice)
        self.use_scale_shift_norm = film = bool(use_scale_shift_norm)
        self.compute_dtype = compute_dtype
        emb = model_channels * 4
        self.time_embed_dim = emb
        self.time_embed = _slot(nn.Linear(model_channels, emb), None, nn.Linear(emb, emb))           # present, never evaluated (timesteps skipped)
        self.input_blocks = nn.ModuleList([_slot(nn.Conv2d(in_channels, model_channels, 3, padding=1))])
        chans, ch, ds = [model_channels], model_channels, 1
        for level, mult in enumerate(self.channel_mult):                                              # openaimodel.py:523-612
            for _ in range(num_res_blocks):
                layers = [_res_params(ch, mult * model_channels, emb, film)]
                ch = mult * model_channels
                if ds in self.attention_resolutions:
                    layers.append(_transformer_params(ch, context_dim, transformer_depth))
                self.input_blocks.append(_slot(*layers))
                chans.append(ch)
            if level != len(self.channel_mult) - 1:
                down = _Params()
                down.op = nn.Conv2d(ch, ch, 3, stride=2, padding=1)
                self.input_blocks.append(_slot(down))
                chans.append(ch)
                ds *= 2
        self.middle_block = _slot(_res_params(ch, ch, emb, film), _transformer_params(ch, context_dim, transformer_depth), _res_params(ch, ch, emb, film))
        self.output_blocks = nn.ModuleList()
        for level, mult in list(enumerate(self.channel_mult))[::-1]:                                  # :651-731
            for i in range(num_res_blocks + 1):
                ich = chans.pop()
                layers = [_res_params(ch + ich, model_channels * mult, emb, film)]
                ch = model_channels * mult
                if ds in self.attention_resolutions:
                    layers.append(_transformer_params(ch, context_dim, transformer_depth))
                if level and i == num_res_blocks:
                    up = _Params()
                    up.conv = nn.Conv2d(ch, ch, 3, padding=1)
                    layers.append(up)
                    ds //= 2
                self.output_blocks.append(_slot(*layers))
        self.out = _slot(nn.GroupNorm(32, ch), None, nn.Conv2d(model_channels, out_channels, 3, padding=1))
        if pose_mlp_name == "single_layer":                                                            # adapt_openaimodel.py:105-116
            self.pose_mlp = _slot(nn.Linear(rot_representation_dim, context_dim))
            self._pose_layers = 1
        else:
            self.pose_mlp = _slot(nn.Linear(rot_representation_dim, context_dim), None, nn.Linear(context_dim, context_dim))
            self._pose_layers = 2
        if self.injecting_condition_twice:                                                             # :119-123
            self.pose_mlp_timesteps = _slot(nn.Linear(rot_representation_dim, emb))
        self._handle: Optional[hip.LdmHandle] = None
        self._handle_key = None
        self.register_load_state_dict_post_hook(lambda mod, _keys: mod.invalidate())

    def own_state_dict(self):
        return {k: v for k, v in self.state_dict().items() if not k.startswith("encoder.")}

    def invalidate(self):
        self._handle = None
        self.__dict__.pop("_own_params", None)      # the cached tensor list: parameters may have been re-assigned (load_state_dict(assign=True))
        inv = getattr(self.encoder, "invalidate", None)
        if callable(inv):
            inv()

    def _weights_version(self):
        # (storage address, version counter) per tensor, as UNet._weights_version: `.data` writes need invalidate()
        ps = self.__dict__.get("_own_params")
        if ps is None:
            ps = self.__dict__["_own_params"] = [p for n, p in self.named_parameters(recurse=True) if not n.startswith("encoder.")]
        return hash(tuple((p.data_ptr(), p._version) for p in ps))

    def _get_handle(self, device) -> "hip.LdmHandle":
        key = (str(device), self.compute_dtype, self._weights_version())
        if self._handle is None or self._handle_key != key:
            sd = {k: v.to(device) for k, v in self.own_state_dict().items() if not k.startswith("time_embed.")}
            levels = len(self.channel_mult)
            cfg = dict(in_channels=self.in_channels, model_channels=self.model_channels, out_channels=self.out_channels,
                       num_res_blocks=self.num_res_blocks, channel_mult=self.channel_mult,
                       attn_levels=tuple(int((1 << l) in self.attention_resolutions) for l in range(levels)),
                       num_head_channels=32, context_dim=self.context_dim, pose_dim=self.rot_representation_dim,
                       pose_mlp_layers=self._pose_layers, injecting_condition_twice=int(self.injecting_condition_twice),
                       use_scale_shift_norm=int(self.use_scale_shift_norm), transformer_depth=self.transformer_depth)
            self._handle = hip.LdmHandle(cfg, sd, hip.dtype_code(self.compute_dtype))
            self._handle_key = key
        return self._handle

    @torch.no_grad()
    def forward(self, x, pose):
        """adapt_openaimodel.py:130-158.  x (B,C,h,w), pose (B,rot_dim) -> (B,out_channels,h,w) f32."""
        return self._get_handle(x.device).forward(x, pose, x_rep=1)

    @torch.no_grad()
    def forward_hypotheses(self, x, poses, out=None, out_dtype="f32"):
        """x (B,C,h,w) reference latents, poses (B,N,rot_dim) -> (B,N,C,h,w): the body of the template loop model.py:212-222."""
        B, N = poses.shape[:2]
        flat = poses.reshape(B * N, poses.shape[-1])
        o = None if out is None else out.view(B * N, *out.shape[2:])
        y = self._get_handle(x.device).forward(x, flat, x_rep=N, out=o, out_dtype=hip.dtype_code(out_dtype))
        return y.view(B, N, *y.shape[1:])
