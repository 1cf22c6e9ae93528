import torch
from torch import nn


class StubEncoder(nn.Module):
    """What UNet/PoseConditional read from an encoder (u_net.py:44-46, model.py:107-108)."""

    def __init__(self, latent_dim=8):
        super().__init__()
        self.latent_dim = latent_dim
        self.name = "template"

    @torch.no_grad()
    def encode_image(self, image, mode=None):
        return image


def rel(a, b):
    return float((a.double() - b.double()).abs().max() / (b.double().abs().max() + 1e-30))


_MODELS = {}


def cached_model(compute_dtype="f32", bank_dtype="f32"):
    """The full-size synthetic model on cuda:0 in a compute mode, built once per test session (building + repacking 305.8 M parameters
    takes ~20 s): a fresh PoseConditional around the cached U-Net / encoder, so tests may set their own bank dtype / flags."""
    from nope_amd.harness import TEMPLATE_BASE, build_model
    from nope_amd.model import PoseConditional
    base = _MODELS.get(compute_dtype)
    if base is None:
        base = _MODELS[compute_dtype] = build_model(compute_dtype=compute_dtype, bank_dtype="f32", device="cuda")
    return PoseConditional(base.u_net, TEMPLATE_BASE["optim_config"], TEMPLATE_BASE["testing_config"], None, bank_dtype=bank_dtype).eval()
