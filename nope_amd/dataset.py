"""Dataset-side producer of the hot path's inputs: the virtual-bounding-box crop of a rendered
frame (`crop_frame`, src/poses/utils.py:204-272), the loader's image transform (src/dataloader/shapeNet.py:64-69) and the
assembly of a test-split sample as `ShapeNet.process` / `__getitem__` build it (shapeNet.py:265-357).

The reference does the crop with OpenCV on the host (`cv2.getPerspectiveTransform` + `cv2.warpPerspective`, one image at a
time); here the 3x3 map is solved on the host in float64 (same four-point system) and the warp runs on the GPU
(`nope_op_warp_perspective`, csrc/kernels_misc.hip), fused with `/255, *2-1, HWC->CHW`, so a frame never leaves the
device between decode and encoder.  cv2 is not installed in this environment and the reference ships no image fixtures:
the crop is **parity-unpinned** against OpenCV's fixed-point bilinear interpolation (it differs by interpolation-weight
quantisation, ~1/32 pixel); geometry (corner correspondences) and interpolation are checked against closed forms and
`torch.nn.functional.grid_sample` in tests/.  File IO, zip repair and random reference sampling stay with the caller.
"""
from __future__ import annotations

from typing import Dict, Optional, Sequence

import numpy as np
import torch

from . import hip
from .poses import compute_relative_pose

SHAPENET_INTRINSIC = np.array([[525.0, 0, 256], [0, 525.0, 256], [0, 0, 1]])      # shapeNet.py:175


def get_perspective_transform(src: np.ndarray, dst: np.ndarray) -> np.ndarray:
    """3x3 M with M (src_i, 1) ~ (dst_i, 1) for four point pairs -- the system cv2.getPerspectiveTransform solves."""
    src, dst = np.asarray(src, np.float64), np.asarray(dst, np.float64)
    A, b = np.zeros((8, 8)), np.zeros(8)
    for i in range(4):
        x, y, u, v = src[i, 0], src[i, 1], dst[i, 0], dst[i, 1]
        A[i] = [x, y, 1, 0, 0, 0, -x * u, -y * u]
        A[i + 4] = [0, 0, 0, x, y, 1, -x * v, -y * v]
        b[i], b[i + 4] = u, v
    h = np.linalg.solve(A, b)
    return np.append(h, 1.0).reshape(3, 3)


def perspective(K: np.ndarray, obj_pose: np.ndarray, pts: np.ndarray) -> np.ndarray:
    """utils.py:50-57: project 3-D points, truncating to int32 as the reference does."""
    R, T = obj_pose[:3, :3], obj_pose[:3, 3]
    out = np.zeros((len(pts), 2))
    for i, p in enumerate(pts):
        rep = K @ (R @ p.reshape(3, 1) + T.reshape(3, 1))
        out[i, 0] = np.int32(rep[0, 0] / rep[2, 0])
        out[i, 1] = np.int32(rep[1, 0] / rep[2, 0])
    return out


def crop_transform(intrinsic: np.ndarray, openCV_pose: np.ndarray, image_size: int, keep_inplane: bool = False,
                   virtual_bbox_size: float = 0.3) -> np.ndarray:
    """The 3x3 map of `crop_frame` (utils.py:204-266): a square of side `virtual_bbox_size` facing the camera around the object
    origin, projected and mapped onto the (image_size x image_size) output."""
    origin = (openCV_pose @ np.array([0, 0, 0, 1.0]))[:3]
    if keep_inplane:
        upper = np.array([0.0, -origin[2], origin[1]])
        right = np.array([origin[1] ** 2 + origin[2] ** 2, -origin[0] * origin[1], -origin[0] * origin[2]])
    else:
        upV = np.array([0, 0, 6.0]) - origin
        upV = (openCV_pose @ np.array([upV[0], upV[1], upV[2], 1.0]))[:3]
        right = np.cross(origin, upV)
        upper = np.cross(right, origin)
    if np.linalg.norm(upper) == 0 and np.linalg.norm(right) == 0:
        upper, right = np.array([0.0, -1, 0]), np.array([1.0, 0, 0])
    upper = upper * (virtual_bbox_size / 2) / np.linalg.norm(upper)
    right = right * (virtual_bbox_size / 2) / np.linalg.norm(right)
    corners = np.stack([origin + upper - right, origin - upper - right, origin + upper + right, origin - upper + right])
    bbox2d = perspective(intrinsic, np.eye(4), corners).astype(np.int32).astype(np.float32)
    target = np.array([[0, 0], [0, 1], [1, 0], [1, 1]], np.float32) * image_size
    return get_perspective_transform(bbox2d, target)


def crop_frame(img, mask, intrinsic, openCV_pose, image_size, keep_inplane=False, virtual_bbox_size=0.3,
               normalize: bool = False, round_u8: bool = False):
    """utils.py:204-272 with the warp on the device.  img (H,W,C) uint8 / f32 tensor (or array) -> (C,S,S) f32 tensor; raw values,
    or the loader's `/255 * 2 - 1` when `normalize`.  `round_u8` (uint8 frames): the warped value is rounded and clamped to
    [0, 255] before the transform, as the reference's uint8 `cv2.warpPerspective` output is before `ToTensor` -- the sample then
    lies on the same k/255 grid as the reference loader's (within one grey level where OpenCV's 1/32-pixel fixed-point weights
    round the other way: unpinned, cv2 is not installed).  Returns (img, mask) when a mask is given."""
    M = crop_transform(np.asarray(intrinsic, np.float64), np.asarray(openCV_pose, np.float64), image_size, keep_inplane, virtual_bbox_size)
    Minv = np.linalg.inv(M)
    def warp(a, norm):
        t = a if isinstance(a, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(np.asarray(a)))
        if t.dim() == 2:
            t = t[..., None]
        if not t.is_cuda and torch.cuda.is_available():
            t = t.cuda()
        sc, sh = ((2.0 / 255.0, -1.0) if t.dtype == torch.uint8 else (2.0, -1.0)) if norm else (1.0, 0.0)
        return hip.op_warp_perspective(t, Minv, image_size, sc, sh, round_u8=round_u8 and t.dtype == torch.uint8)
    out = warp(img, normalize)
    return (out, warp(mask, False)) if mask is not None else out


def process_test_sample(query_img, reference_img, template_imgs: Sequence, query_pose: np.ndarray, ref_pose: np.ndarray,
                        template_img_poses: Sequence[np.ndarray], testing_template_poses: np.ndarray, img_size: int = 256,
                        symmetry: int = 0) -> Dict[str, torch.Tensor]:
    """`ShapeNet.process` + `__getitem__` on the test split (shapeNet.py:265-357) for already decoded frames: crops (virtual bbox
    of size 1, shapeNet.py:171-184), image transform, relative poses; keys and shapes as the reference's sample dict."""
    crop = lambda im, pose: crop_frame(im, None, SHAPENET_INTRINSIC, pose, img_size, virtual_bbox_size=1, normalize=True, round_u8=True)
    rel, _ = compute_relative_pose(query_pose, ref_pose)
    all_rel = torch.stack([compute_relative_pose(testing_template_poses[i], ref_pose)[0] for i in range(len(template_imgs))])
    return {
        "query": crop(query_img, query_pose), "reference": crop(reference_img, ref_pose),
        "gt_relativeR": rel, "all_relativeR": all_rel,
        "gt_templates": torch.stack([crop(im, p) for im, p in zip(template_imgs, template_img_poses)]),
        "symmetry": torch.tensor([float(symmetry)]),
        "query_pose": torch.from_numpy(np.asarray(query_pose))[:3, :3],
        "template_poses": torch.from_numpy(np.asarray(testing_template_poses))[:, :3, :3],
    }
