"""Import the *reference* (nv-nguyen/nope at /root/reference) in the build container.

Test infrastructure, container-only.  `/root/reference` does not exist on the GPU box,
so nothing at run time may import this module; it is used by `make_golden.py` (which
writes the committed fixtures) and by `tests/test_oracle_vs_reference.py` (skipped
when the reference tree is absent).

The reference imports a number of packages that are not installed here
(pytorch_lightning, diffusers, torchvision, pytorch3d, wandb, cv2, ...).  None of
them contributes arithmetic to the hot path: they are import-time
only, or `pl.LightningModule` used as a base class.  We register stub modules for
them before importing.
"""
import importlib.abc
import importlib.machinery
import os
import sys
import types

import torch
from torch import nn

REF_ROOT = "/root/reference"

_STUB_TOPLEVEL = (
    "pytorch_lightning", "diffusers", "einops_exts", "torchvision", "wandb", "imageio",
    "cv2", "trimesh", "pyrender", "pytorch3d", "ruamel_yaml", "ruamel", "moviepy", "omegaconf",
    "hydra", "pandas_stub_unused", "open3d", "blobfile", "mpi4py",
)


class _Anything:
    """Callable/attribute sink used for every name looked up on a stub module."""

    def __init__(self, name="stub"):
        self._name = name

    def __call__(self, *a, **k):
        return _Anything(self._name + "()")

    def __getattr__(self, item):
        if item.startswith("__"):
            raise AttributeError(item)
        return _Anything(self._name + "." + item)

    def __iter__(self):
        return iter(())

    def __mro_entries__(self, bases):
        return (object,)


class _StubModule(types.ModuleType):
    def __getattr__(self, item):
        if item.startswith("__"):
            raise AttributeError(item)
        return _Anything(self.__name__ + "." + item)


class _StubFinder(importlib.abc.MetaPathFinder, importlib.abc.Loader):
    def find_spec(self, fullname, path, target=None):
        top = fullname.split(".")[0]
        if top in _STUB_TOPLEVEL:
            return importlib.machinery.ModuleSpec(fullname, self, is_package=True)
        return None

    def create_module(self, spec):
        m = _StubModule(spec.name)
        m.__path__ = []
        return m

    def exec_module(self, module):
        if module.__name__ == "pytorch_lightning":
            class LightningModule(nn.Module):
                global_step = 0
                global_rank = 0

                @property
                def dtype(self):
                    return torch.float32

                def log(self, *a, **k):
                    pass

            module.LightningModule = LightningModule
            module.seed_everything = lambda *a, **k: None


_installed = False


def install():
    global _installed
    if _installed:
        return
    if not os.path.isdir(REF_ROOT):
        raise RuntimeError("reference tree not present (expected only in the build container)")
    sys.meta_path.insert(0, _StubFinder())
    sys.path.insert(0, REF_ROOT)
    _installed = True


def available():
    return os.path.isdir(os.path.join(REF_ROOT, "src", "model"))


def ref_unet_cls():
    install()
    from src.model.u_net.denoising_diffusion_pytorch.u_net import UNet
    return UNet


def ref_model_utils():
    install()
    from src.model.u_net.denoising_diffusion_pytorch import model_utils
    return model_utils


def ref_feature_extractor_cls():
    install()
    from src.model.encoder.template import FeatureExtractor
    return FeatureExtractor


def ref_pose_conditional_cls():
    install()
    from src.model.model import PoseConditional
    return PoseConditional


class StubEncoder(nn.Module):
    """Minimal encoder exposing what `UNet.__init__`/`PoseConditional` read
    (u_net.py:44-46; model.py:107-108): `.latent_dim`, `.name`, `encode_image`."""

    def __init__(self, latent_dim=8):
        super().__init__()
        self.latent_dim = latent_dim
        self.name = "template"

    @torch.no_grad()
    def encode_image(self, image, mode=None):
        return image
