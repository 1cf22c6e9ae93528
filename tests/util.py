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
