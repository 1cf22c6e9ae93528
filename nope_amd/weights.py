"""Deterministic synthetic weights:  w = f(seed, state_dict_key, shape).

The reference's checkpoints (NOPE U-Net, `resnet50_template_pose.pth`) are external
downloads (configs/model/template_base.yaml:9,14,25) and there is no network, so every
parity fixture and benchmark runs at a *defined* random initialisation.  Full-size
weights (305.8 M parameters) cannot be committed; instead each tensor is a pure function
of `(seed, key, shape)` drawn from torch's CPU generator, so the build container (where
the reference is imported to produce golden outputs) and the GPU box (same image)
materialise bit-identical tensors.  `sha256_of` lets fixtures record what they used.

Scales follow `torch.nn` defaults (uniform +-1/sqrt(fan_in)) so activations stay O(1);
norm affine parameters are perturbed away from (1, 0) so the affine code paths are
actually exercised.
"""
from __future__ import annotations

import hashlib
import math
import zlib
from typing import Dict, Iterable, Tuple

import torch


def _gen(seed: int, key: str) -> torch.Generator:
    g = torch.Generator(device="cpu")
    g.manual_seed((zlib.crc32(key.encode()) << 16) ^ (seed & 0xFFFF_FFFF))
    return g


def synth_tensor(seed: int, key: str, shape: Tuple[int, ...]) -> torch.Tensor:
    g = _gen(seed, key)
    u = torch.rand(shape, generator=g, dtype=torch.float32) * 2.0 - 1.0
    leaf = key.rsplit(".", 1)[-1]
    if leaf == "num_batches_tracked":
        return torch.zeros(shape, dtype=torch.int64)
    if leaf == "running_var":
        return 1.0 + 0.5 * u
    if leaf == "running_mean":
        return 0.1 * u
    if len(shape) == 1:
        is_norm = ".norm." in key or key.endswith("to_out.1.weight") or key.endswith("to_out.1.bias") \
            or ".bn" in key or "downsample.1." in key
        if is_norm:
            return (1.0 + 0.1 * u) if leaf == "weight" else 0.1 * u
        return 0.05 * u                                   # conv / linear bias
    fan_in = 1
    for d in shape[1:]:
        fan_in *= d
    return u / math.sqrt(fan_in)


def synth_state_dict(shapes: Iterable[Tuple[str, Tuple[int, ...]]], seed: int) -> Dict[str, torch.Tensor]:
    return {k: synth_tensor(seed, k, tuple(s)) for k, s in shapes}


@torch.no_grad()
def synth_init_(module: torch.nn.Module, seed: int, prefix: str = "") -> torch.nn.Module:
    """Overwrite every parameter/buffer of `module` in place with `synth_tensor`."""
    for k, v in module.state_dict().items():
        v.copy_(synth_tensor(seed, prefix + k, tuple(v.shape)).to(v.dtype))
    return module


def sha256_of(t: torch.Tensor) -> str:
    return hashlib.sha256(t.detach().cpu().contiguous().numpy().tobytes()).hexdigest()
