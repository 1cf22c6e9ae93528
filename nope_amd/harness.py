"""Evaluation harness -- counterpart of the reference's `test_shapeNet.py`, which the README
names (README.md:82) but the tree does not contain.

Builds the model from the values of configs/model/template_base.yaml, feeds batches shaped
like `ShapeNet.__getitem__` on the test split (src/dataloader/shapeNet.py:348-357, keyed
"shapeNet_<category>" as model.py:551-552 expects) and runs the hot path:
`generate_templates -> retrieval -> template_poses[0][nearest_idx]` (model.py:313,323,352),
then the geodesic angle / Acc@{15,30} (loss.py:76-115; non-symmetric branch only).  There is
no dataset and no checkpoint in the tree or on the box, so images, poses and weights are
synthetic and seeded; the numbers are plumbing checks, not accuracy claims.  It does not
replicate the reference's `vis_imgs` UnboundLocalError (model.py:367): predictions are saved
with `query_pose` and `similarity` only.

    python -m nope_amd.harness --batch 1 --templates 64 --size 128     # BASELINE config 1 shape
"""
from __future__ import annotations

import argparse
import json
import os
import time
from typing import Dict, Optional

import numpy as np
import torch

TEMPLATE_BASE = dict(          # configs/model/template_base.yaml
    u_net=dict(u_net_dim=192, rot_representation_dim=6, pose_mlp_name="single_layer",
               encoder=dict(descriptor_size=8, threshold=0.2, normalize=False)),
    optim_config=dict(loss_type="l1", lr=5e-5, weight_decay=0.0005, warm_up_steps=500, use_inv_deltaR=True),
    testing_config=dict(similarity_metric="l2"),
)


class StubEncoder(torch.nn.Module):
    """An encoder that hands embeddings through unchanged: exactly what UNet / PoseConditional read from an encoder (`latent_dim`, `name`,
    `encode_image`, u_net.py:44-46, model.py:107-108) and nothing else.  Scoring-only runs and operator-level checks build a U-Net
    around it when the images are already embeddings."""

    def __init__(self, latent_dim=8):
        super().__init__()
        self.latent_dim = latent_dim
        self.name = "template"

    @torch.no_grad()
    def encode_image(self, image, mode=None):
        return image


def random_rotations(n: int, gen: torch.Generator) -> torch.Tensor:
    """Haar-distributed rotations, float64 (n,3,3): QR of a Gaussian with sign fix."""
    a = torch.randn(n, 3, 3, generator=gen, dtype=torch.float64)
    q, r = torch.linalg.qr(a)
    q = q * torch.sign(torch.diagonal(r, dim1=-2, dim2=-1)).unsqueeze(-2)
    det = torch.linalg.det(q)
    q[:, :, 0] = q[:, :, 0] * det.unsqueeze(-1)
    return q


def rotation_6d(m: torch.Tensor) -> torch.Tensor:
    """6D representation = first two rows, flattened (src/poses/rotation_conversions.py:490-503)."""
    return m[..., :2, :].clone().reshape(*m.shape[:-2], 6)


def synthetic_batch(batch: int, n_templates: int, size: int, seed: int = 2022, device="cpu",
                    pose_level: Optional[int] = None, pose_root: Optional[str] = None) -> Dict[str, torch.Tensor]:
    """A test-split batch shaped like ShapeNet.__getitem__ (dataloader/shapeNet.py:348-357).  Template poses are
    Haar-random rotations, or -- `pose_level` 0..3 -- the upper-hemisphere icosphere grid the reference evaluates on
    (nope_amd.poses: 26 / 91 / 341 / 1321 viewpoints; `pose_root` = the reference's predefined_poses directory to
    read its own files); `n_templates` is ignored then."""
    g = torch.Generator().manual_seed(seed)
    query = torch.rand(batch, 3, size, size, generator=g) * 2 - 1
    reference = torch.rand(batch, 3, size, size, generator=g) * 2 - 1
    if pose_level is not None:
        from .poses import get_obj_poses_from_template_level
        grid = get_obj_poses_from_template_level(pose_level, "upper", root=pose_root)
        R_tpl = torch.from_numpy(grid[:, :3, :3].copy())
        n_templates = R_tpl.shape[0]
    else:
        R_tpl = random_rotations(n_templates, g)
    R_ref = random_rotations(batch, g)
    R_query = random_rotations(batch, g)
    rel = lambda a, b: a @ torch.linalg.inv(b)          # shapeNet.py:243-245
    all_rel = rotation_6d(rel(R_tpl[None], R_ref[:, None])).float()
    gt_rel = rotation_6d(rel(R_query, R_ref)).float()
    out = dict(query=query, reference=reference, gt_relativeR=gt_rel, all_relativeR=all_rel,
               symmetry=torch.zeros(batch, 1), query_pose=R_query,
               template_poses=R_tpl[None].expand(batch, -1, -1, -1).contiguous())
    return {k: v.to(device) for k, v in out.items()}


def build_model(seed: int = 2022, compute_dtype="f32", bank_dtype="f32", device="cuda", u_net_dim: Optional[int] = None,
                save_dir: Optional[str] = None, template_parallel: bool = False, max_hypotheses_per_launch: int = 512):
    from .encoder import FeatureExtractor
    from .model import PoseConditional
    from .u_net import UNet
    from .weights import synth_init_
    cfg = TEMPLATE_BASE
    enc = FeatureExtractor(**cfg["u_net"]["encoder"], compute_dtype=compute_dtype)
    synth_init_(enc, seed, prefix="encoder.")
    unet = UNet(u_net_dim=u_net_dim or cfg["u_net"]["u_net_dim"], rot_representation_dim=6, encoder=enc,
                pose_mlp_name=cfg["u_net"]["pose_mlp_name"], compute_dtype=compute_dtype)
    # U-Net tensors are keyed without the "encoder." prefix; the encoder was initialised above
    from .weights import synth_tensor
    with torch.no_grad():
        for k, v in unet.state_dict().items():
            if not k.startswith("encoder."):
                v.copy_(synth_tensor(seed, k, tuple(v.shape)))
    model = PoseConditional(unet, cfg["optim_config"], cfg["testing_config"], save_dir, bank_dtype=bank_dtype,
                            template_parallel=template_parallel, max_hypotheses_per_launch=max_hypotheses_per_launch)
    return model.to(device).eval()


from .metrics import GeodesicError  # noqa: E402


def geodesic_deg(predR: torch.Tensor, gtR: torch.Tensor) -> torch.Tensor:
    rel = predR.double() @ gtR.double().transpose(-1, -2)
    cos = (rel.diagonal(dim1=-2, dim2=-1).sum(-1) - 1.0) / 2.0
    return torch.rad2deg(torch.acos(cos.clamp(-1.0, 1.0)))


@torch.no_grad()
def eval_geodesic(model, batch: Dict[str, torch.Tensor], thresholds=(15, 30), save_path: Optional[str] = None):
    """The body of PoseConditional.eval_geodesic (model.py:268-376) without visualisation."""
    loss = model.forward(batch["query"], batch["reference"], batch["gt_relativeR"])
    similarity, nearest_idx, _ = model.generate_and_retrieve(batch["query"], batch["reference"], batch["all_relativeR"])
    sym = batch.get("symmetry", torch.zeros(nearest_idx.shape[0], 1, dtype=torch.long, device=nearest_idx.device))
    # pred_R = template_poses[0][nearest_idx] -> GeodesicError (model.py:352-358, loss.py:78-115): gather + angle + symmetry branches as
    # one device launch (nope_op_geodesic); the grid of the first sample serves every query, as model.py:352 indexes it
    _, metric = GeodesicError(list(thresholds)).from_indices(batch["template_poses"][:1], nearest_idx, batch["query_pose"], sym)
    res = {"loss": float(loss)}
    res.update({k: float(v) for k, v in metric.items()})
    if save_path:
        np.savez(save_path, query_pose=batch["query_pose"].cpu().numpy(), similarity=similarity.cpu().numpy())
    return similarity, nearest_idx, res


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__.split("\n")[0])
    ap.add_argument("--batch", type=int, default=1)
    ap.add_argument("--templates", type=int, default=64)
    ap.add_argument("--size", type=int, default=128)
    ap.add_argument("--dtype", default="f32", choices=["f32", "f16x2", "bf16x3", "f16", "bf16"])
    ap.add_argument("--bank-dtype", default="f32", choices=["f32", "bf16", "f16"])
    ap.add_argument("--seed", type=int, default=2022)
    ap.add_argument("--category", default="synthetic")
    ap.add_argument("--save-dir", default=None)
    ap.add_argument("--pose-level", type=int, default=None, choices=[0, 1, 2, 3],
                    help="use the upper-hemisphere icosphere grid (26/91/341/1321 templates) instead of --templates random poses")
    ap.add_argument("--pose-root", default=None, help="directory with the reference's predefined_poses/*.npy")
    a = ap.parse_args(argv)
    if not torch.cuda.is_available():
        raise SystemExit("nope_amd.harness needs an MI355X (no CPU fallback)")
    model = build_model(a.seed, a.dtype, a.bank_dtype, "cuda", save_dir=a.save_dir)
    batches = {f"shapeNet_{a.category}": synthetic_batch(a.batch, a.templates, a.size, a.seed, "cuda", pose_level=a.pose_level,
                                                                  pose_root=a.pose_root)}
    for name, batch in batches.items():                         # test_step, model.py:550-565
        t0 = time.time()
        save = os.path.join(a.save_dir, "predictions", f"pred_step0_rank{model.global_rank}") if a.save_dir else None
        sim, idx, res = eval_geodesic(model, batch, save_path=save)
        torch.cuda.synchronize()
        res.update(dataloader=name, seconds=time.time() - t0, nearest_idx=idx.tolist())
        print(json.dumps(res))


if __name__ == "__main__":
    main()
