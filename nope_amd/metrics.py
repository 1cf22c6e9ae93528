"""`GeodesicError` -- the metric applied right after the hot path (src/model/loss.py:14-115).

Device tensors go through the C ABI: ONE launch of `nope_op_geodesic` (csrc/kernels_metric.hip) evaluates the angle with the
reference's three symmetry branches for all (B, k) retrieved poses -- `geodesic_from_indices` gathers `template_poses[nearest_idx]`
(model.py:352-354) inside that launch.  Host tensors (the CPU tests, the fixtures recorded from the reference's own loss.py) take
the torch restatement below; both are pinned to the same hand-computed known answers of pytorch3d's published formula.

The reference delegates the angle to `pytorch3d.transforms.so3_relative_angle(pred, gt, eps=1e-2)`, an
un-vendored and unpinned dependency that is not installed here: **parity unpinned**.
This file restates pytorch3d's published algorithm (transforms/so3.py `so3_relative_angle` /
`so3_rotation_angle`, transforms/math.py `acos_linear_extrapolation`):

    angle(R1, R2) = acos_le((trace(R1 R2^T) - 1) / 2),   acos_le(x) = acos(x)                          |x| <= 1 - 1e-4
                                                                    = acos(b) - (x - b) / sqrt(1 - b^2)  beyond bound b
    and a ValueError when the trace leaves [-1 - eps, 3 + eps],

plus the reference's own symmetry handling (loss.py:14-75: 0 = none, 1 = 180 degrees about Y, 2 = circular) and the
top-1 / top-k result dictionaries (loss.py:85-115), including their float64 casts.  Tiny 3x3 arithmetic on
(B, k) poses: plain torch on whatever device the poses live on.
"""
from __future__ import annotations

import math
from typing import Dict, Sequence, Tuple

import torch
import torch.nn.functional as F

_COS_BOUND = 1e-4            # pytorch3d default `cos_bound`


def acos_linear_extrapolation(x: torch.Tensor, bound: float = 1.0 - _COS_BOUND) -> torch.Tensor:
    """acos inside [-bound, bound], first-order Taylor extrapolation outside (pytorch3d transforms/math.py)."""
    def ext(v, b):      # pytorch3d's `_acos_linear_approximation`: (x - x0) * dacos_dx(x0) + acos(x0), dacos_dx(x) = -1 / sqrt(1 - x^2)
        return (v - b) * (-1.0 / math.sqrt(1.0 - b * b)) + math.acos(b)
    out = torch.empty_like(x)
    hi, lo = x >= bound, x <= -bound
    mid = ~(hi | lo)
    out[mid] = torch.acos(x[mid])
    out[hi] = ext(x[hi], bound)
    out[lo] = ext(x[lo], -bound)
    return out


def so3_relative_angle(R1: torch.Tensor, R2: torch.Tensor, eps: float = 1e-4) -> torch.Tensor:
    """Angle (radians) of R1 R2^T for batches of 3x3 rotations (pytorch3d transforms/so3.py)."""
    if R1.shape[0] == 0:
        return R1.new_zeros((0,))
    R12 = torch.bmm(R1[:, :3, :3], R2[:, :3, :3].transpose(1, 2))
    tr = R12[:, 0, 0] + R12[:, 1, 1] + R12[:, 2, 2]
    if bool(((tr < -1.0 - eps) | (tr > 3.0 + eps)).any()):
        raise ValueError("A matrix has trace outside valid range [-1-eps,3+eps].")
    return acos_linear_extrapolation((tr - 1.0) * 0.5)


def _roty180(device, dtype):
    # load_rotation_transform("y", 180)[:3, :3].float()  (poses/utils.py:136-139): rounded through f32 as there
    c, s = math.cos(math.pi), math.sin(math.pi)
    m = torch.tensor([[c, 0.0, s], [0.0, 1.0, 0.0], [-s, 0.0, c]], dtype=torch.float64).float()
    return m.to(device=device, dtype=dtype)


def _opencv_to_opengl(R: torch.Tensor) -> torch.Tensor:
    # convert_openCV_to_openGL_torch, poses/utils.py:142-152 (same op sequence -- a batched product against the repeated
    # transform -- so that the unclamped acos of the circular branch sees the reference's bits: an exact match gives a cosine
    # within one ulp of 1 and acos returns 0 or NaN depending on that ulp; fixture metric_ref.npz holds both outcomes)
    t = torch.tensor([[1, 0, 0], [0, -1, 0], [0, 0, -1]], device=R.device, dtype=R.dtype).unsqueeze(0).repeat(R.shape[0], 1, 1)
    return torch.bmm(t, R[:, :3, :3])


def geodesic_from_indices(template_poses: torch.Tensor, nearest_idx: torch.Tensor, gt: torch.Tensor, symmetry: torch.Tensor) -> torch.Tensor:
    """model.py:352-357 in one device launch: error (radians, float64, (B, k)) of `template_poses[b, nearest_idx[b, j]]` against `gt[b]`
    with the reference's symmetry handling.  template_poses (B|1, N, 3, 3), nearest_idx (B, k) int64 (PoseConditional.retrieval's)."""
    from . import hip
    return hip.op_geodesic(template_poses, gt, symmetry, idx=nearest_idx)


def so3_relative_angle_with_symmetry(pred: torch.Tensor, gt: torch.Tensor, symmetry: torch.Tensor) -> torch.Tensor:
    """loss.py:14-75.  pred, gt (B,3,3); symmetry (B,) or (B,1) in {0,1,2}.  Radians."""
    if pred.is_cuda:                  # the HIP kernel: one thread per pose, float64
        from . import hip
        return hip.op_geodesic(pred[:, None].to(torch.float64), gt, symmetry)[:, 0].to(pred.dtype)
    sym = symmetry.reshape(-1).to(pred.device)
    non = sym == 0
    e_non = so3_relative_angle(pred[non], gt[non], eps=1e-2)
    if int(non.sum()) == pred.shape[0]:
        return e_non
    err = torch.zeros(pred.shape[0], device=pred.device, dtype=pred.dtype)
    err[non] = e_non
    two = sym == 1
    if bool(two.any()):
        e2 = so3_relative_angle(pred[two], gt[two], eps=1e-2)
        rot = torch.matmul(_roty180(pred.device, torch.float32), pred[two].float()).float()      # loss.py:36-43 (f32 product)
        e2r = so3_relative_angle(rot.to(torch.float64), gt[two].to(torch.float64), eps=1e-2)
        err[two] = torch.minimum(e2, e2r.to(err.dtype))
    if int(non.sum()) + int(two.sum()) == pred.shape[0]:
        return err
    cir = sym == 2
    p_gl = _opencv_to_opengl(pred[cir].clone()[:, :3, :3].inverse())       # object pose -> camera pose -> OpenGL axes (loss.py:56-63)
    g_gl = _opencv_to_opengl(gt[cir].clone()[:, :3, :3].inverse())
    err[cir] = torch.acos(F.cosine_similarity(p_gl[:, 2, :3], g_gl[:, 2, :3]))   # only the viewing (Z) axis matters
    return err


class GeodesicError(torch.nn.Module):
    """loss.py:78-115: returns (error of the top-1 prediction in degrees, dict of accuracy/median entries)."""

    def __init__(self, thresholds: Sequence[float] = (15,)):
        super().__init__()
        self.thresholds = list(thresholds)

    @torch.no_grad()
    def forward(self, predR: torch.Tensor, gtR: torch.Tensor, symmetry: torch.Tensor) -> Tuple[torch.Tensor, Dict[str, torch.Tensor]]:
        if predR.dim() == 3:                                         # top 1: (B,3,3)
            error = torch.rad2deg(so3_relative_angle_with_symmetry(predR.to(torch.float64), gtR.to(torch.float64), symmetry))
            res = {f"top1, accuracy_{t}": (error <= t).float().mean() * 100 for t in self.thresholds}
            res["top1, median"] = error.median()
            return error, res
        if predR.is_cuda:             # all k ranks in ONE launch instead of k passes over the symmetry branches
            from . import hip
            rad = hip.op_geodesic(predR.to(torch.float64), gtR, symmetry)
        else:
            rad = torch.stack([so3_relative_angle_with_symmetry(predR[:, k].to(torch.float64), gtR.to(torch.float64), symmetry)
                               for k in range(predR.shape[1])], 1)
        return self._topk_result(rad)

    @torch.no_grad()
    def from_indices(self, template_poses: torch.Tensor, nearest_idx: torch.Tensor, gtR: torch.Tensor, symmetry: torch.Tensor):
        """`GeodesicError(template_poses[nearest_idx], gtR, symmetry)` (model.py:352-357) with the gather inside the metric launch:
        template_poses (B|1, N, 3, 3) on the device, nearest_idx (B, k) as PoseConditional.retrieval returns it."""
        return self._topk_result(geodesic_from_indices(template_poses, nearest_idx, gtR, symmetry))

    def _topk_result(self, rad: torch.Tensor):
        """loss.py:97-115 from the (B, k) angles in radians (float64)."""
        res: Dict[str, torch.Tensor] = {}
        errors = torch.zeros(tuple(rad.shape), device=rad.device)     # f32, as loss.py:101
        for k in range(rad.shape[1]):
            errors[:, k] = torch.rad2deg(rad[:, k].to(errors.dtype))
            if k in (0, 2, 4):
                top = errors[:, :k + 1].min(dim=1).values
                for t in self.thresholds:
                    res[f"top{k + 1}, accuracy_{t}"] = (top <= t).float().mean() * 100
                    res[f"top{k + 1}, median"] = top.median()
        return errors[:, 0], res
