"""Copies the reference's level-0 viewpoint grid (data, 42 poses) into tests/golden/pose_grid_level0.npz so that the CPU
suite can check nope_amd.poses against it without /root/reference.   python tests/golden/make_pose_fixture.py"""
import os

import numpy as np

SRC = "/root/reference/src/poses/predefined_poses"
HERE = os.path.dirname(os.path.abspath(__file__))
out = {k: np.load(os.path.join(SRC, f)) for k, f in [
    ("sphere_poses_level0", "sphere_poses_level0.npy"), ("obj_poses_level0", "obj_poses_level0.npy"),
    ("idx_upper_level0_in_level2", "idx_upper_level0_in_level2.npy"), ("idx_all_level0_in_level2", "idx_all_level0_in_level2.npy")]}
# level 1-3: camera positions only would be 40 KB+; keep their counts
for l in (1, 2, 3):
    c = np.load(os.path.join(SRC, f"sphere_poses_level{l}.npy"))
    out[f"count_level{l}"] = np.array([len(c), int((c[:, 2, 3] >= 0).sum())])
np.savez_compressed(os.path.join(HERE, "pose_grid_level0.npz"), **out)
print({k: v.shape for k, v in out.items()})
