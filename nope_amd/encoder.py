"""`FeatureExtractor` -- the "template" encoder on the input side of the hot path
(src/model/encoder/template.py:24-53; ResNet-50 trunk src/model/encoder/resnet.py:55-152).

ResNet-50 without max-pool / avg-pool / fc, strides (2,1,2,2,1) -> /8, then
ReLU -> 1x1(2048->256) -> ReLU -> 1x1(256->descriptor_size); optional channel L2-normalise.
Same constructor kwargs, attributes (`latent_dim`, `name`, `normalize`) and state-dict keys
as the reference (including its aliased `encoder.0.* / encoder.1.*` entries and the unused
`backbone.fc`), so `resnet50_template_pose.pth` loads unchanged.

Execution: device tensors go through the C ABI (`nope_encoder_*`,
csrc/encoder_runtime.hip): eval-mode BatchNorm folded into the conv weights at handle creation, every
1x1 / 3x3 / stride-2 conv on the implicit-GEMM MFMA kernel with bias + residual + ReLU in its epilogue,
conv1 as a small direct kernel -- no PyTorch/MIOpen arithmetic, and no fallback when the library is
missing.  The `nn.Module` tree below only holds the parameters under the reference's keys (there is no torch
forward in it; the CPU restatement used by the tests lives in oracle/nope_ref.py).
It runs once per query and once per reference image (the reference re-runs it N times, model.py:115).
"""
from __future__ import annotations

import math

import torch
from torch import nn

from . import hip

_LAYERS = (3, 4, 6, 3)
_STRIDES = (1, 2, 2, 1)      # resnet.py:102-105


class _Holder(nn.Module):
    """Parameter container: never called."""

    def forward(self, *a, **k):  # pragma: no cover
        raise RuntimeError("parameter holder; the encoder runs in libnope_hip.so (FeatureExtractor.encode_image)")


class _Bottleneck(_Holder):
    def __init__(self, cin, planes, stride, project):
        super().__init__()
        self.conv1 = nn.Conv2d(cin, planes, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(planes)
        self.conv2 = nn.Conv2d(planes, planes, 3, stride=stride, padding=1, bias=False)
        self.bn2 = nn.BatchNorm2d(planes)
        self.conv3 = nn.Conv2d(planes, planes * 4, 1, bias=False)
        self.bn3 = nn.BatchNorm2d(planes * 4)
        self.downsample = None
        if project:
            self.downsample = nn.Sequential(nn.Conv2d(cin, planes * 4, 1, stride=stride, bias=False),
                                            nn.BatchNorm2d(planes * 4))


class _Trunk(_Holder):
    def __init__(self, features=64):
        super().__init__()
        self.conv1 = nn.Conv2d(3, features, 7, stride=2, padding=3, bias=False)
        self.bn1 = nn.BatchNorm2d(features)
        cin = features
        for i, (n, s) in enumerate(zip(_LAYERS, _STRIDES), start=1):
            planes = features * 2 ** (i - 1)
            blocks = []
            for b in range(n):
                blocks.append(_Bottleneck(cin, planes, s if b == 0 else 1, b == 0))
                cin = planes * 4
            setattr(self, f"layer{i}", nn.Sequential(*blocks))
        self.fc = nn.Linear(cin, 1)          # present (unused) in the reference: resnet.py:107, num_classes=1
        for m in self.modules():             # resnet.py:110-116
            if isinstance(m, nn.Conv2d):
                n_ = m.kernel_size[0] * m.kernel_size[1] * m.out_channels
                m.weight.data.normal_(0, math.sqrt(2.0 / n_))


class FeatureExtractor(nn.Module):
    def __init__(self, descriptor_size, threshold=0.2, normalize=False, compute_dtype="f32", **kwargs):
        super().__init__()
        self.compute_dtype = compute_dtype      # "f32": parity mode, "bf16": throughput mode (as UNet.compute_dtype)
        self._handle = None
        self._handle_key = None
        self.latent_dim = descriptor_size
        self.normalize = normalize
        self.threshold = threshold
        self.name = "template"
        self.backbone = _Trunk()
        self.projector = nn.Sequential(nn.ReLU(), nn.Conv2d(2048, 256, 1, bias=False), nn.ReLU(),
                                       nn.Conv2d(256, descriptor_size, 1, bias=False))
        self.encoder = nn.Sequential(self.backbone, self.projector)   # aliased, as in template.py:40
        self.eval()
        # runs for a load_state_dict on this module or on any parent (UNet, PoseConditional, a Lightning module)
        self.register_load_state_dict_post_hook(lambda mod, _keys: mod.invalidate())

    def invalidate(self):
        """Drop the folded / repacked device weights; the next device call rebuilds them."""
        self._handle = None
        self.__dict__.pop("_own_tensors", None)      # the cached tensor list: parameters may have been re-assigned (load_state_dict(assign=True))

    def _weights_version(self):
        # (storage address, version counter) per parameter / BatchNorm buffer, as UNet._weights_version: `.data` writes need invalidate()
        ts = self.__dict__.get("_own_tensors")
        if ts is None:
            ts = self.__dict__["_own_tensors"] = list(self.backbone.parameters()) + list(self.backbone.buffers()) + list(self.projector.parameters())
        return hash(tuple((t.data_ptr(), t._version) for t in ts))

    def _get_handle(self, device) -> "hip.EncoderHandle":
        key = (str(device), self.compute_dtype, self._weights_version())
        if self._handle is None or self._handle_key != key:
            sd = {k: v.to(device) for k, v in self.state_dict().items() if k.startswith(("backbone.", "projector."))}
            self._handle = hip.EncoderHandle(self.latent_dim, sd, hip.dtype_code(self.compute_dtype),
                                             bn_eps=self.backbone.bn1.eps)
            self._handle_key = key
        return self._handle

    @torch.no_grad()
    def encode_image(self, image, mode=None):
        """template.py:47-53 through the C ABI.  `mode` is accepted and ignored, as in the reference."""
        hip.require_device(image)
        feat = self._get_handle(image.device).forward(image)
        if self.normalize:                                   # F.normalize(dim=1), eps 1e-12 (template.py:51-52)
            feat = feat / feat.norm(dim=1, keepdim=True).clamp_min(1e-12)
        return feat

    encode_image_hip = encode_image
