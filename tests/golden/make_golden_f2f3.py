"""Fixtures that pin SURVEY.md section 8 rows f2 (metric) and f3 (pose grids / relative poses / crop geometry) to the
reference's OWN code, run in the build container.        python tests/golden/make_golden_f2f3.py

What runs here is /root/reference/src/model/loss.py, src/poses/utils.py and src/dataloader/shapeNet.py themselves, imported
with the stubs of _ref_import.py plus two bindings for the third-party calls they make:
  * `pytorch3d.transforms.so3_relative_angle` (pytorch3d is not installed, not vendored, not pinned: SURVEY 8 c4) is bound to
    `nope_amd.metrics.so3_relative_angle`, the restatement of pytorch3d's published algorithm.  Everything AROUND that call --
    the three symmetry branches, the f32 round trip of roty180, the float64 casts, the top-k dictionaries (loss.py:14-115) --
    is the reference's code, so `metric_ref.npz` pins those; the angle function itself stays unpinned.
  * `pytorch3d.transforms.matrix_to_rotation_6d` is bound to the reference's vendored copy (src/poses/rotation_conversions.py).
  * `cv2` is a recording stub: `getPerspectiveTransform` stores the four source / target points `crop_frame` hands it
    (utils.py:204-260 is numpy up to that call), `warpPerspective` returns zeros.  OpenCV's interpolation stays unpinned.
Only inputs and outputs are written (tests/golden/metric_ref.npz, poses_ref.npz); no reference source travels."""
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)
import _ref_import  # noqa: E402

_ref_import.install()
from nope_amd import metrics as M  # noqa: E402
from nope_amd.harness import random_rotations  # noqa: E402

# ---- bindings (before the reference modules are imported) ----------------------------------------------------------
p3d = types.ModuleType("pytorch3d")
p3d.__path__ = []
p3dt = types.ModuleType("pytorch3d.transforms")
p3dt.so3_relative_angle = M.so3_relative_angle
from src.poses import rotation_conversions as RC  # noqa: E402  (the reference's vendored conversions)
p3dt.matrix_to_rotation_6d = RC.matrix_to_rotation_6d
p3dt.matrix_to_euler_angles = RC.matrix_to_euler_angles
p3dt.matrix_to_quaternion = RC.matrix_to_quaternion
p3d.transforms = p3dt
sys.modules["pytorch3d"], sys.modules["pytorch3d.transforms"] = p3d, p3dt

CV_CALLS = []
cv2 = types.ModuleType("cv2")


def _gpt(src, dst):
    CV_CALLS.append((np.array(src, dtype=np.float64), np.array(dst, dtype=np.float64)))
    return np.eye(3)


cv2.getPerspectiveTransform = _gpt
cv2.warpPerspective = lambda img, M_, size: np.zeros((size[1], size[0]) + tuple(np.asarray(img).shape[2:]), dtype=np.uint8)
sys.modules["cv2"] = cv2

from src.model import loss as RL  # noqa: E402
from src.poses import utils as RU  # noqa: E402
from src.dataloader.shapeNet import ShapeNet  # noqa: E402


def metric_fixture():
    g = torch.Generator().manual_seed(2024)
    out = {}
    B = 36
    pred = random_rotations(B * 5, g).view(B, 5, 3, 3)
    gt = random_rotations(B, g)
    # make a few predictions (nearly) exact so that the small-angle end of acos is exercised
    pred[0, 0], pred[1, 0] = gt[0], gt[1] @ random_rotations(1, g)[0].matrix_power(0)
    cases = {"mixed": (torch.arange(B) % 3).float(), "none": torch.zeros(B), "two": torch.ones(B), "circle": torch.full((B,), 2.0),
             "none_two": (torch.arange(B) % 2).float()}
    out["pred"], out["gt"] = pred.numpy().copy(), gt.numpy().copy()
    ref = RL.GeodesicError([15, 30])
    # the circular branch takes an UNCLAMPED acos of a cosine similarity (loss.py:66-73): on an exact match the cosine lands within
    # one ulp of 1 and the reference itself returns 0 or NaN from run to run (vectorised reductions on differently aligned
    # buffers), so the all-circular case gets predictions without exact matches; exact matches stay in the other branches
    pred_c = pred.clone()
    pred_c[0, 0], pred_c[1, 0] = pred[0, 1], pred[1, 1]
    out["circle/pred"] = pred_c.numpy().copy()
    for tag, sym in cases.items():
        out[f"{tag}/symmetry"] = sym.numpy()
        for form, p in (("topk", pred_c if tag == "circle" else pred), ("top1", (pred_c if tag == "circle" else pred)[:, 0])):
            err, res = ref(p, gt, sym)
            out[f"{tag}/{form}/error"] = err.numpy()
            keys = sorted(res)
            out[f"{tag}/{form}/keys"] = np.array(keys)
            out[f"{tag}/{form}/values"] = np.array([float(res[k]) for k in keys], dtype=np.float64)
    # the helper on its own (radians, f64)
    e = RL.so3_relative_angle_with_symmetry(pred[:, 1].double(), gt.double(), cases["mixed"])
    out["mixed/helper_rad"] = e.numpy()
    out["roty180"] = RL.roty180.numpy()
    np.savez_compressed(os.path.join(HERE, "metric_ref.npz"), **out)
    print("metric_ref.npz:", len(out), "arrays")


def poses_fixture():
    out = {}
    for level in (0, 1):
        for dist in ("upper", "all"):
            idx, poses = RU.get_obj_poses_from_template_level(level, dist, return_index=True)
            out[f"L{level}/{dist}/index"] = idx
            if level == 0:
                out[f"L{level}/{dist}/obj_poses"] = poses
                out[f"L{level}/{dist}/cam_poses"] = RU.get_obj_poses_from_template_level(level, dist, return_cam=True)
    for dist in ("upper", "all"):
        out[f"idx_level0_in_level2/{dist}"] = RU.load_index_level0_in_level2(dist)
        m = RU.load_mapping_id_templates_to_idx_pose_distribution(0, dist)
        out[f"mapping_L0/{dist}"] = np.array(sorted(m.items()), dtype=np.int64)
    for l in (2, 3):
        idx, _ = RU.get_obj_poses_from_template_level(l, "upper", return_index=True)
        out[f"L{l}/upper/count"] = np.array([len(idx)])
    # perspective (utils.py:50-57) and the crop geometry of crop_frame (utils.py:204-260)
    rng = np.random.default_rng(7)
    K = np.array([[525.0, 0, 256], [0, 525.0, 256], [0, 0, 1]])
    obj_poses = RU.get_obj_poses_from_template_level(0, "upper")
    pts = rng.normal(size=(6, 3)) * 0.1
    out["perspective/K"], out["perspective/pts"] = K, pts
    out["perspective/pose"] = obj_poses[3]
    out["perspective/out"] = RU.perspective(K, obj_poses[3], pts)
    img = np.zeros((512, 512, 3), dtype=np.uint8)
    crop_in, crop_src, crop_dst = [], [], []
    for i, (pose_i, inplane, vb, size) in enumerate([(0, False, 0.3, 256), (5, False, 1.0, 256), (11, True, 0.3, 128), (17, True, 1.0, 64),
                                                     (25, False, 1.0, 128)]):
        pose = obj_poses[pose_i].copy()
        pose[:3, 3] += rng.normal(size=3) * 0.02          # off-centre object: the four points are not symmetric
        CV_CALLS.clear()
        RU.crop_frame(img, None, K, pose, size, keep_inplane=inplane, virtual_bbox_size=vb)
        (src, dst), = CV_CALLS
        crop_in.append(np.concatenate([pose.reshape(-1), [float(inplane), vb, size]]))
        crop_src.append(src)
        crop_dst.append(dst)
    out["crop/in"], out["crop/src"], out["crop/dst"] = np.array(crop_in), np.array(crop_src), np.array(crop_dst)
    # ShapeNet.compute_relative_pose (shapeNet.py:243-251) through the class's own methods
    this = types.SimpleNamespace(rot_representation="rotation6d")
    this.convert_rotation_representation = lambda r: ShapeNet.convert_rotation_representation(this, r)
    q, r = obj_poses[2], obj_poses[9]
    rel, rel_inv = ShapeNet.compute_relative_pose(this, q, r)
    out["relpose/query"], out["relpose/ref"] = q, r
    out["relpose/rel"], out["relpose/rel_inv"] = rel.numpy(), rel_inv.numpy()
    allrel = [ShapeNet.compute_relative_pose(this, obj_poses[i], r)[0].numpy() for i in range(len(obj_poses))]
    out["relpose/all"] = np.array(allrel)
    np.savez_compressed(os.path.join(HERE, "poses_ref.npz"), **out)
    print("poses_ref.npz:", len(out), "arrays")


def nearest_fixture():
    """NearestTemplateFinder (utils.py:318-356) of the reference on random query poses, with the grid it searched (so that the restatement can be
    run on the same grid without the reference's files): nearest_ref.npz."""
    from scipy.spatial.transform import Rotation
    rng = np.random.default_rng(11)
    out = {}
    for tag, level, dist in (("L0_upper", 0, "upper"), ("L1_all", 1, "all")):
        f = RU.NearestTemplateFinder(level, dist, return_inplane=True)
        M_ = 24
        q = np.tile(np.eye(4), (M_, 1, 1))
        q[:, :3, :3] = Rotation.random(M_, random_state=int(rng.integers(1 << 30))).as_matrix()
        q[:, :3, 3] = rng.normal(size=(M_, 3)) * 0.3
        q[0] = f.obj_template_poses[3]                      # an exact grid pose: in-plane 0, index of that pose
        q[1, :3, :3] = Rotation.from_euler("z", 37.0, degrees=True).as_matrix() @ f.obj_template_poses[5][:3, :3]     # ... turned in the image plane
        idx, inp = f.search_nearest_template(q)
        out[f"{tag}/avail_index"], out[f"{tag}/obj_template_poses"] = f.avail_index, f.obj_template_poses
        out[f"{tag}/query"], out[f"{tag}/index"], out[f"{tag}/inplane"] = q, idx, inp
        f2 = RU.NearestTemplateFinder(level, dist, return_inplane=False)
        out[f"{tag}/index_only"] = f2.search_nearest_template(q)
    np.savez_compressed(os.path.join(HERE, "nearest_ref.npz"), **out)
    print("nearest_ref.npz:", len(out), "arrays")


if __name__ == "__main__":
    if "--nearest-only" not in sys.argv:
        metric_fixture()
        poses_fixture()
    nearest_fixture()
