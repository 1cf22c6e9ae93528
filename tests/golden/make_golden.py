"""Generate the committed golden fixtures from THE REFERENCE ITSELF.

Run in the build container only (needs /root/reference):

    python tests/golden/make_golden.py

The reference (nv-nguyen/nope) ships no tests or golden vectors, so every
fixture here is an input/output pair recorded from the reference's own PyTorch modules,
imported with non-arithmetic dependencies stubbed (tests/golden/_ref_import.py).  Weights are
`nope_amd.weights.synth_tensor(seed, key, shape)` -- a pure function, so full-size weights are
never stored; the fixtures keep only inputs, outputs and SHA-256 digests of a few weight
tensors (to detect generator drift).  Fixtures are data; no reference source is copied.
"""
from __future__ import annotations

import os
import sys
import tempfile
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)

import _ref_import as RI  # noqa: E402
from nope_amd.encoder import FeatureExtractor  # noqa: E402
from nope_amd.u_net import UNet  # noqa: E402
from nope_amd.weights import sha256_of, synth_init_, synth_tensor  # noqa: E402

SEED = 2022   # the reference's seed (train.py:14)


def save(name, **arrs):
    out = {}
    for k, v in arrs.items():
        if isinstance(v, torch.Tensor):
            v = v.detach().cpu().numpy()
        out[k] = v
    path = os.path.join(HERE, name)
    np.savez_compressed(path, **out)
    print(f"{name}: {os.path.getsize(path) / 1024:.1f} KiB")


def synth_module_(m, seed, prefix):
    with torch.no_grad():
        for k, v in m.state_dict().items():
            v.copy_(synth_tensor(seed, prefix + k, tuple(v.shape)))
    return m


@torch.no_grad()
def blocks():
    mu = RI.ref_model_utils()
    g = torch.Generator().manual_seed(SEED)
    rn = lambda *s: torch.randn(*s, generator=g)
    out = {}

    def record(tag, mod, *inputs):
        synth_module_(mod, SEED, tag + ".")
        y = mod(*inputs)
        for k, v in mod.state_dict().items():
            out[f"{tag}/w/{k}"] = v.clone()
        for i, t in enumerate(inputs):
            out[f"{tag}/in{i}"] = t
        out[f"{tag}/out"] = y

    x = rn(2, 16, 6, 6)
    emb = rn(2, 32)
    record("block", mu.Block(16, 24, groups=8), x)
    record("resnet_proj", mu.ResnetBlock(16, 24, time_emb_dim=32, groups=8), x, emb)
    record("resnet_id", mu.ResnetBlock(16, 16, time_emb_dim=32, groups=8), x, emb)
    record("linattn", mu.Residual(mu.PreNorm(16, mu.LinearAttention(16))), x)
    record("attn", mu.Residual(mu.PreNorm(16, mu.Attention(16))), rn(2, 16, 4, 4))
    record("down", mu.HardDownsample(16, 24), x)
    record("up", mu.HardUpsample(16, 8), x)
    save("blocks.npz", **out)


@torch.no_grad()
def unets():
    U = RI.ref_unet_cls()
    out = {}
    g = torch.Generator().manual_seed(SEED + 1)
    # (d16soft: use_hard_up_down=False -- Conv2d(4, 2, 1) / ConvTranspose2d(4, 2, 1) resampling, u_net.py:54-59; last, so that the
    #  draws of the earlier cases stay what they were)
    for tag, dim, hw, mlp, hard in (("d8", 8, 8, "single_layer", True), ("d16", 16, 16, "single_layer", True), ("d16two", 16, 8, "two_layers", True),
                                    ("d24pos", 24, 8, "posEncoding", True), ("d16soft", 16, 16, "single_layer", False)):
        mine = UNet(u_net_dim=dim, rot_representation_dim=6, encoder=RI.StubEncoder(8), pose_mlp_name=mlp, use_hard_up_down=hard)
        synth_init_(mine, SEED)
        ref = U(u_net_dim=dim, rot_representation_dim=6, encoder=RI.StubEncoder(8), pose_mlp_name=mlp, use_hard_up_down=hard)
        ref.load_state_dict(mine.state_dict(), strict=True)       # also proves key/shape parity
        x = torch.randn(3, 8, hw, hw, generator=g)
        pose = torch.randn(3, 6, generator=g)
        out[f"{tag}/x"] = x
        out[f"{tag}/pose"] = pose
        out[f"{tag}/out"] = ref(x, pose)
        out[f"{tag}/sha_init_conv"] = np.array(sha256_of(mine.state_dict()["init_conv.weight"]))
    save("unet_tiny.npz", **out)


def _ns(**k):
    return types.SimpleNamespace(**k)


@torch.no_grad()
def retrieval():
    PC = RI.ref_pose_conditional_cls()
    U = RI.ref_unet_cls()
    unet = U(u_net_dim=8, rot_representation_dim=6, encoder=RI.StubEncoder(8), pose_mlp_name="single_layer")
    tmp = tempfile.mkdtemp()
    pc = PC(unet, _ns(lr=5e-5, weight_decay=5e-4, warm_up_steps=500, use_inv_deltaR=True, loss_type="l1"),
            _ns(similarity_metric="l2"), tmp)
    g = torch.Generator().manual_seed(SEED + 2)
    out = {}
    for tag, (B, N, C, h) in {"a": (2, 26, 8, 32), "b": (3, 7, 8, 16), "c": (1, 5, 4, 8)}.items():
        q = torch.randn(B, C, h, h, generator=g)
        bank = torch.randn(B, N, C, h, h, generator=g)
        bank[B - 1, N // 2] = q[B - 1]            # planted exact match -> score -0.0, arg-top = that slot
        sim, idx = pc.retrieval(q, bank)          # StubEncoder.encode_image is the identity
        out[f"{tag}/q"], out[f"{tag}/bank"], out[f"{tag}/sim"], out[f"{tag}/idx"] = q, bank, sim, idx
    save("retrieval.npz", **out)


@torch.no_grad()
def encoder_and_pipeline():
    FE = RI.ref_feature_extractor_cls()
    U = RI.ref_unet_cls()
    PC = RI.ref_pose_conditional_cls()
    mine_enc = FeatureExtractor(descriptor_size=8, threshold=0.2, normalize=False)
    synth_init_(mine_enc, SEED, prefix="encoder.")
    ref_enc = FE(descriptor_size=8, threshold=0.2, normalize=False)
    ref_enc.load_state_dict(mine_enc.state_dict(), strict=True)
    ref_enc.eval()
    g = torch.Generator().manual_seed(SEED + 3)
    img = torch.rand(2, 3, 64, 64, generator=g) * 2 - 1
    save("encoder.npz", img=img, feat=ref_enc.encode_image(img),
         sha_conv1=np.array(sha256_of(mine_enc.state_dict()["backbone.conv1.weight"])))

    # ---- BASELINE config 1: single query, 64-template bank, 128x128, full-size model -------------
    mine = UNet(u_net_dim=192, rot_representation_dim=6, encoder=mine_enc, pose_mlp_name="single_layer")
    sd = {k: synth_tensor(SEED, k, tuple(v.shape)) for k, v in mine.own_state_dict().items()}
    ref = U(u_net_dim=192, rot_representation_dim=6, encoder=ref_enc, pose_mlp_name="single_layer")
    missing = ref.load_state_dict(sd, strict=False)
    assert all(k.startswith("encoder.") for k in missing.missing_keys) and not missing.unexpected_keys
    ref.eval()
    tmp = tempfile.mkdtemp()
    pc = PC(ref, _ns(lr=5e-5, weight_decay=5e-4, warm_up_steps=500, use_inv_deltaR=True, loss_type="l1"),
            _ns(similarity_metric="l2"), tmp)
    from nope_amd.harness import synthetic_batch
    batch = synthetic_batch(batch=1, n_templates=64, size=128, seed=SEED)
    import time
    t0 = time.time()
    bank, _, _ = pc.generate_templates(batch["reference"], batch["all_relativeR"], None, visualize=False)
    sim, idx = pc.retrieval(batch["query"], bank)
    loss = pc.forward(batch["query"], batch["reference"], batch["gt_relativeR"])
    print(f"config-1 reference run: {time.time() - t0:.1f}s; top-5 {idx.tolist()} gap {float(sim.sort(descending=True).values[0, 0] - sim.sort(descending=True).values[0, 1]):.4f}")
    save("pipeline_cfg1.npz", query=batch["query"], reference=batch["reference"], all_relativeR=batch["all_relativeR"],
         gt_relativeR=batch["gt_relativeR"], sim=sim, idx=idx, loss=loss, bank_head=bank[:, :4],
         query_feat=ref_enc.encode_image(batch["query"]), reference_feat=ref_enc.encode_image(batch["reference"]),
         sha_mid=np.array(sha256_of(sd["mid_block1.block1.proj.weight"])))

    # ---- full-size U-Net at the 32x32 latent of a 256x256 input ------------------------------------
    x = torch.randn(1, 8, 32, 32, generator=g)
    pose = torch.randn(2, 6, generator=g)
    y = ref(x.expand(2, -1, -1, -1), pose)
    save("unet_full_32.npz", x=x, pose=pose, out=y)


@torch.no_grad()
def ldm():
    """The LDM cross-attention variant (UNetModelPose): tiny configurations with synthesised weights -- every zero_module
    conv is overwritten, so the outputs are not identically zero as they are at the reference's own initialisation."""
    RI.install()
    from src.model.u_net.ldm.adapt_openaimodel import UNetModelPose as RefLdm
    from nope_amd.ldm import UNetModelPose
    out = {}
    g = torch.Generator().manual_seed(SEED + 7)
    for tag, kw, hw in (("m32", dict(model_channels=32, channel_mult=(1, 2), num_res_blocks=1, attention_resolutions=[1, 2], context_dim=24,
                                     pose_mlp_name="single_layer", injecting_condition_twice=False), 8),
                        ("m64two", dict(model_channels=64, channel_mult=(1, 2, 2), num_res_blocks=2, attention_resolutions=[2, 4], context_dim=40,
                                        pose_mlp_name="two_layers", injecting_condition_twice=True), 8),
                        # FiLM ResBlocks (use_scale_shift_norm, openaimodel.py:277-281): timestep embedding zeros / from the pose
                        ("m32film", dict(model_channels=32, channel_mult=(1, 2), num_res_blocks=1, attention_resolutions=[1, 2], context_dim=24,
                                         pose_mlp_name="single_layer", injecting_condition_twice=False, use_scale_shift_norm=True), 8),
                        ("m64film", dict(model_channels=64, channel_mult=(1, 2), num_res_blocks=1, attention_resolutions=[2], context_dim=40,
                                         pose_mlp_name="single_layer", injecting_condition_twice=True, use_scale_shift_norm=True), 8),
                        # two BasicTransformerBlocks per SpatialTransformer (transformer_depth = 2, attention.py:251-258)
                        ("m32d2", dict(model_channels=32, channel_mult=(1, 2), num_res_blocks=1, attention_resolutions=[1, 2], context_dim=24,
                                       pose_mlp_name="single_layer", injecting_condition_twice=False, transformer_depth=2), 8)):
        common = dict(rot_representation_dim=6, image_size=hw, in_channels=8, out_channels=8, num_head_channels=32,
                      use_spatial_transformer=True, transformer_depth=1)
        common.update(kw)
        mine = UNetModelPose(encoder=RI.StubEncoder(8), **common)
        synth_init_(mine, SEED)
        ref = RefLdm(encoder=RI.StubEncoder(8), **common)
        ref.load_state_dict(mine.state_dict(), strict=True)       # proves key / shape parity
        ref.eval()
        x = torch.randn(3, 8, hw, hw, generator=g)
        pose = torch.randn(3, 6, generator=g)
        y = ref(x, pose)
        assert float(y.abs().max()) > 1e-3
        out[f"{tag}/x"], out[f"{tag}/pose"], out[f"{tag}/out"] = x, pose, y
        out[f"{tag}/sha_in"] = np.array(sha256_of(mine.state_dict()["input_blocks.0.0.weight"]))
    save("ldm_tiny.npz", **out)


if __name__ == "__main__":
    torch.manual_seed(SEED)
    which = sys.argv[1:] or ["blocks", "unets", "retrieval", "pipeline"]
    if "blocks" in which:
        blocks()
    if "unets" in which:
        unets()
    if "retrieval" in which:
        retrieval()
    if "pipeline" in which:
        encoder_and_pipeline()
    if "ldm" in which:
        ldm()
