"""`UNet` -- drop-in for the reference's pose-conditioned U-Net
(src/model/u_net/denoising_diffusion_pytorch/u_net.py:26-198), executed by libnope_hip.so.

Same constructor kwargs, same attributes (`encoder`, `channels`, `name`, `out_dim`,
`rot_representation_dim`), same `state_dict()` keys and shapes -- a reference checkpoint
loads with `load_state_dict` (or the reference's shape-filtered `load_checkpoint`,
src/utils/weight.py:6-37) -- and the same call: `u_net(x, pose) -> pred`.

The module tree below only *holds parameters* under the reference's names; there is no
torch arithmetic in it.  `forward` hands the state dict to the C ABI once (weights are
repacked to the kernels' layout on device) and then launches the HIP schedule.
`forward_hypotheses` is the batched form used by `PoseConditional.generate_templates`:
N pose hypotheses per reference embedding in one launch sequence.
"""
from __future__ import annotations

from typing import Optional

import torch
from torch import nn

from . import hip

_HIDDEN = 4 * 32   # attention heads x dim_head (model_utils.py:368,394)


def _slot(*mods):
    """nn.Sequential whose indices reproduce the reference's key numbering; entries that
    carry no parameters in the reference (SiLU, Upsample, Rearrange) are placeholders."""
    return nn.Sequential(*[m if m is not None else nn.Identity() for m in mods])


class _Params(nn.Module):
    """A parameter container: never called."""

    def forward(self, *a, **k):  # pragma: no cover
        raise RuntimeError("parameter holder; computation happens in libnope_hip.so")


def _conv(cin, cout, k, bias=True):
    return nn.Conv2d(cin, cout, k, padding=k // 2, bias=bias)


def _resnet_params(cin, cout, emb_dim, groups):
    m = _Params()
    m.mlp = _slot(None, nn.Linear(emb_dim, cout))                 # model_utils.py:261-265
    for name, ci in (("block1", cin), ("block2", cout)):          # :267-268
        b = _Params()
        b.proj = _conv(ci, cout, 3)
        b.norm = nn.GroupNorm(groups, cout)
        setattr(m, name, b)
    if cin != cout:                                               # :269
        m.res_conv = _conv(cin, cout, 1)
    return m


def _attention_params(dim, linear):
    inner = _Params()
    inner.to_qkv = _conv(dim, 3 * _HIDDEN, 1, bias=False)         # :373,399
    if linear:
        inner.to_out = _slot(_conv(_HIDDEN, dim, 1), nn.GroupNorm(1, dim))   # :401
    else:
        inner.to_out = _conv(_HIDDEN, dim, 1)                     # :374
    pre = _Params()                                               # PreNorm :226-234
    pre.fn = inner
    pre.norm = nn.GroupNorm(1, dim)
    res = _Params()                                               # Residual :198-204
    res.fn = pre
    return res


class UNet(nn.Module):
    def __init__(self, u_net_dim, rot_representation_dim, encoder, pose_mlp_name, init_dim=None, out_dim=None,
                 use_hard_up_down=True, dim_mults=(1, 2, 4, 8), resnet_block_groups=8, compute_dtype="f32", **kwargs):
        super().__init__()
        if init_dim not in (None, u_net_dim):
            # (not a gap of this implementation: the reference's own forward fails for init_dim != u_net_dim -- final_res_block is built
            #  for 2 * u_net_dim input channels but receives 2 * init_dim, u_net.py:143-147,194; checked against the reference class)
            raise NotImplementedError("init_dim != u_net_dim (the reference's forward raises a channel mismatch for it as well)")
        self.encoder = encoder
        self.channels = encoder.latent_dim
        self.name = encoder.name
        self.out_dim = out_dim if out_dim is not None else self.channels
        self.rot_representation_dim = rot_representation_dim
        self.u_net_dim = u_net_dim
        self.use_hard_up_down = bool(use_hard_up_down)
        self.dim_mults = tuple(dim_mults)
        self.groups = resnet_block_groups
        self.compute_dtype = compute_dtype
        emb = 4 * u_net_dim
        if pose_mlp_name == "single_layer":
            self.pose_mlp = _slot(nn.Linear(rot_representation_dim, emb))
            self._pose_layers = 1
        elif pose_mlp_name == "two_layers":
            self.pose_mlp = _slot(nn.Linear(rot_representation_dim, emb), None, nn.Linear(emb, emb))
            self._pose_layers = 2
        elif pose_mlp_name == "posEncoding":          # u_net.py:73-76: parameter-free SinusoidalPosEmb(dim = emb / 6)
            assert emb % (2 * rot_representation_dim) == 0, "classes_dim must be divisible by 2 * rot_representation_dim"
            self.pose_mlp = _Params()
            self._pose_layers = 0
        else:
            raise NotImplementedError(f"pose_mlp_name={pose_mlp_name!r} (reference default is 'single_layer')")
        dims = [u_net_dim] + [u_net_dim * m for m in dim_mults]
        g = resnet_block_groups
        self.init_conv = _conv(self.channels, u_net_dim, 3)
        self.downs = nn.ModuleList()
        self.ups = nn.ModuleList()
        n = len(dim_mults)
        for l in range(n):
            cin, cout = dims[l], dims[l + 1]
            last = l == n - 1
            self.downs.append(nn.ModuleList([
                _resnet_params(cin, cin, emb, g), _resnet_params(cin, cin, emb, g), _attention_params(cin, True),
                _conv(cin, cout, 3) if last else (_slot(None, _conv(4 * cin, cout, 1)) if use_hard_up_down       # u_net.py:54-59
                                                  else nn.Conv2d(cin, cout, 4, 2, 1))]))                           # model_utils.py:129-136
        mid = dims[-1]
        self.mid_attn = _attention_params(mid, False)
        self.mid_block1 = _resnet_params(mid, mid, emb, g)
        self.mid_block2 = _resnet_params(mid, mid, emb, g)
        for l in range(n):
            r = n - 1 - l
            cin, cout = dims[r], dims[r + 1]
            last = l == n - 1
            self.ups.append(nn.ModuleList([
                _resnet_params(cout + cin, cout, emb, g), _resnet_params(cout + cin, cout, emb, g),
                _attention_params(cout, True),
                _conv(cout, cin, 3) if last else (_slot(None, _conv(cout, cin, 3)) if use_hard_up_down
                                                  else nn.ConvTranspose2d(cout, cin, 4, 2, 1))]))                   # model_utils.py:119-126
        self.final_res_block = _resnet_params(2 * u_net_dim, u_net_dim, emb, g)
        self.final_conv = _slot(_resnet_params(u_net_dim, u_net_dim, emb, g), _conv(u_net_dim, self.channels, 1))
        self._handle: Optional[hip.UNetHandle] = None
        self._handle_key = None
        # nn.Module.load_state_dict on a PARENT never calls a child's load_state_dict (it recurses through
        # _load_from_state_dict), but it does run every sub-module's post hooks: drop the repacked device copy there.
        self.register_load_state_dict_post_hook(lambda mod, _keys: mod.invalidate())

    # -- device handle ------------------------------------------------------------------
    def own_state_dict(self):
        """The U-Net's own tensors (no `encoder.*`), keyed as in the reference."""
        return {k: v for k, v in self.state_dict().items() if not k.startswith("encoder.")}

    def invalidate(self):
        """Drop the repacked device weights (also the encoder's); the next forward repacks them.  Called
        automatically after any load_state_dict that reaches this module and when a parameter was modified in place."""
        self._handle = None
        self.__dict__.pop("_own_params", None)      # the cached tensor list: parameters may have been re-assigned (load_state_dict(assign=True))
        inv = getattr(self.encoder, "invalidate", None)
        if callable(inv):
            inv()

    def _weights_version(self):
        """Key of the repacked device copy: (storage address, version counter) of every tensor of the U-Net.  In-place writes through
        the tensor itself (optimizer steps, `p.copy_`, `p.mul_`) bump `_version`, re-assigned storage moves `data_ptr`; writes
        through `p.data` have their own counter and are NOT seen -- call `invalidate()` after those (EMA swaps, hand-written
        checkpoint loaders); `load_state_dict` on this module or any parent invalidates by itself.  The parameter list is
        cached (the module tree is fixed after __init__), so a forward pays ~40 us for the ~420 tensors, not a named_parameters walk."""
        ps = self.__dict__.get("_own_params")
        if ps is None:
            ps = self.__dict__["_own_params"] = [p for n, p in self.named_parameters(recurse=True) if not n.startswith("encoder.")]
        return hash(tuple((p.data_ptr(), p._version) for p in ps))

    def _get_handle(self, device) -> hip.UNetHandle:
        key = (str(device), self.compute_dtype, self._weights_version())
        if self._handle is None or self._handle_key != key:
            sd = {k: v.to(device) for k, v in self.own_state_dict().items()}
            # (the reference stores `out_dim` but builds final_conv.1 with `channels` outputs, u_net.py:154-157)
            cfg = dict(u_net_dim=self.u_net_dim, channels=self.channels, out_dim=self.channels,
                       pose_dim=self.rot_representation_dim, dim_mults=self.dim_mults, groups=self.groups,
                       pose_mlp_layers=self._pose_layers, soft_up_down=int(not self.use_hard_up_down))
            self._handle = hip.UNetHandle(cfg, sd, hip.dtype_code(self.compute_dtype))
            self._handle_key = key
        return self._handle

    # -- reference call surface ---------------------------------------------------------
    @torch.no_grad()
    def forward(self, x, pose):
        """u_net.py:160-198.  x (B,C,h,w), pose (B,rot_dim) -> (B,C,h,w) f32."""
        return self._get_handle(x.device).forward(x, pose, x_rep=1)

    @torch.no_grad()
    def forward_hypotheses(self, x, poses, out=None, out_dtype="f32", defer_range_check=False):
        """x (B,C,h,w) reference embeddings, poses (B,N,rot_dim) -> (B,N,C,h,w):
        UNet(x[b], poses[b,n]) for every (b,n) -- the body of the template loop
        model.py:212-222 -- as one batched launch sequence."""
        B, N = poses.shape[:2]
        flat = poses.reshape(B * N, poses.shape[-1])
        o = None if out is None else out.view(B * N, *out.shape[2:])
        y = self._get_handle(x.device).forward(x, flat, x_rep=N, out=o, out_dtype=hip.dtype_code(out_dtype), defer_range_check=defer_range_check)
        return y.view(B, N, *y.shape[1:])

    def finish_range_check(self) -> bool:
        """f16x2: check (and if needed repeat) the forwards issued with defer_range_check; True when any was repeated (hip.UNetHandle)."""
        return self._handle.finish_range_check() if self._handle is not None else False
