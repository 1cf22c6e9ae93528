"""Viewpoint grids and relative poses -- the caller-side producer of `all_relativeR` / `template_poses`
(src/poses/utils.py:72-112, src/dataloader/shapeNet.py:243-256,300-306).

The reference reads its grids from `src/poses/predefined_poses/{obj,sphere}_poses_level{0..3}.npy` (icospheres with 42 /
162 / 642 / 2562 cameras on the unit sphere, OpenCV camera looking at the object; `obj_pose = inverse(cam_pose)` with
the translation halved).  Those files are data of the reference checkout and are not shipped here:
`get_obj_poses_from_template_level(..., root=<that directory>)` reads them with the reference's selection semantics
("upper" = cameras with z >= 0: 26 / 91 / 341 / 1321), and without `root` the same grids are synthesised -- a
subdivided icosahedron with two vertices on the z axis gives the same camera POSITIONS (tests/test_host_logic.py
checks this against the level-0 file); the in-plane camera roll is this file's own look-at convention, which no
arithmetic on the path depends on (relative rotations between a template and the reference view are what the
U-Net consumes).
"""
from __future__ import annotations

import functools
import os
from typing import Optional, Tuple

import numpy as np
import torch


# ---- icosphere ---------------------------------------------------------------------------------------------
def _icosahedron() -> Tuple[np.ndarray, np.ndarray]:
    """12 vertices with the poles on +-z, 20 faces."""
    a = np.arctan(0.5)
    v = [(0.0, 0.0, 1.0)]
    v += [(np.cos(a) * np.cos(2 * np.pi * k / 5), np.cos(a) * np.sin(2 * np.pi * k / 5), np.sin(a)) for k in range(5)]
    v += [(np.cos(a) * np.cos(2 * np.pi * (k + 0.5) / 5), np.cos(a) * np.sin(2 * np.pi * (k + 0.5) / 5), -np.sin(a)) for k in range(5)]
    v += [(0.0, 0.0, -1.0)]
    f = []
    for k in range(5):
        u0, u1 = 1 + k, 1 + (k + 1) % 5
        l0, l1 = 6 + k, 6 + (k + 1) % 5
        f += [(0, u0, u1), (u0, l0, u1), (u1, l0, l1), (11, l1, l0)]
    return np.array(v, dtype=np.float64), np.array(f, dtype=np.int64)


def icosphere_vertices(subdivisions: int) -> np.ndarray:
    """Unit vectors of an icosahedron subdivided `subdivisions` times: 12, 42, 162, 642, 2562, ...  Vertices of level s
    keep their index at level s+1 (new midpoints are appended), so coarser grids are prefixes of finer ones."""
    v, f = _icosahedron()
    verts = [tuple(x) for x in v]
    for _ in range(subdivisions):
        cache, nf = {}, []

        def mid(i, j):
            key = (i, j) if i < j else (j, i)
            if key not in cache:
                m = (np.array(verts[i]) + np.array(verts[j])) * 0.5
                z_exact = verts[i][2] == -verts[j][2]            # equator midpoints stay exactly on z = 0
                m = m / np.linalg.norm(m)
                if z_exact:
                    m[2] = 0.0
                verts.append(tuple(m))
                cache[key] = len(verts) - 1
            return cache[key]
        for a, b, c in f:
            ab, bc, ca = mid(a, b), mid(b, c), mid(c, a)
            nf += [(a, ab, ca), (b, bc, ab), (c, ca, bc), (ab, bc, ca)]
        f = np.array(nf, dtype=np.int64)
    return np.array(verts, dtype=np.float64)


def look_at_cam_pose(position: np.ndarray) -> np.ndarray:
    """4x4 camera-to-world pose (OpenCV axes: +z forward, +y down) of a camera at `position` looking at the origin."""
    z = -position / np.linalg.norm(position)
    up = np.array([0.0, 0.0, 1.0])
    x = np.cross(z, up)
    if np.linalg.norm(x) < 1e-9:                   # camera on the z axis: any roll; pick world -x as image-right for +z... (fixed)
        x = np.array([-1.0, 0.0, 0.0]) if position[2] < 0 else np.array([1.0, 0.0, 0.0])
    x = x / np.linalg.norm(x)
    y = np.cross(z, x)
    T = np.eye(4)
    T[:3, 0], T[:3, 1], T[:3, 2], T[:3, 3] = x, y, z, position
    return T


@functools.lru_cache(maxsize=8)
def _grid_cached(level: int):
    return _synthesize_grid(level)


def synthesize_grid(level: int) -> Tuple[np.ndarray, np.ndarray]:
    """(cam_poses, obj_poses), each (n,4,4) float64, n = 42 / 162 / 642 / 2562 for level 0..3."""
    cams, objs = _grid_cached(level)
    return cams.copy(), objs.copy()


def _synthesize_grid(level: int) -> Tuple[np.ndarray, np.ndarray]:
    pos = icosphere_vertices(level + 1)
    cams = np.stack([look_at_cam_pose(p) for p in pos])
    objs = np.linalg.inv(cams)
    objs[:, :3, 3] *= 0.5                           # the reference renders at half the unit distance
    return cams, objs


# ---- utils.py:72-112 -----------------------------------------------------------------------------------------
def get_obj_poses_from_template_level(level: int, pose_distribution: str, return_cam: bool = False, return_index: bool = False,
                                      root: Optional[str] = None):
    """utils.py:72-100.  `root`: directory holding the reference's predefined_poses/*.npy; None -> synthesised grid."""
    if root is not None:
        cams = np.load(os.path.join(root, f"sphere_poses_level{level}.npy"))
        objs = np.load(os.path.join(root, f"obj_poses_level{level}.npy"))
    else:
        cams, objs = synthesize_grid(level)
    poses = cams if return_cam else objs
    if pose_distribution == "all":
        keep = np.ones(len(poses), dtype=bool)
    elif pose_distribution == "upper":
        keep = cams[:, 2, 3] >= 0
    else:
        raise ValueError(f"unknown pose_distribution {pose_distribution!r}")     # (the reference silently returns None)
    if return_index:
        return np.arange(len(poses))[keep], poses[keep]
    return poses[keep]


def load_index_level0_in_level2(pose_distribution: str, root: Optional[str] = None) -> np.ndarray:
    """utils.py:103-110: indices (into the level-2 grid restricted to `pose_distribution`) of the level-0 viewpoints."""
    if root is not None:
        return np.load(os.path.join(root, f"idx_{pose_distribution}_level0_in_level2.npy"))
    c0 = get_obj_poses_from_template_level(0, pose_distribution, return_cam=True)[:, :3, 3]
    c2 = get_obj_poses_from_template_level(2, pose_distribution, return_cam=True)[:, :3, 3]
    d = np.linalg.norm(c0[:, None] - c2[None], axis=-1)
    idx = d.argmin(axis=1)
    assert float(d[np.arange(len(c0)), idx].max()) < 1e-9      # coarse vertices are vertices of the finer grid
    return idx


def load_mapping_id_templates_to_idx_pose_distribution(level: int, pose_distribution: str, root: Optional[str] = None):
    """utils.py:113-125."""
    index_range, _ = get_obj_poses_from_template_level(level, pose_distribution, return_index=True, root=root)
    return {int(t): i for i, t in enumerate(index_range)}


# ---- shapeNet.py:243-251, rotation_conversions.py:490-503 --------------------------------------------------------
def matrix_to_rotation_6d(m: torch.Tensor) -> torch.Tensor:
    return m[..., :2, :].clone().reshape(*m.shape[:-2], 6)


def compute_relative_pose(query_pose, ref_pose) -> Tuple[torch.Tensor, torch.Tensor]:
    """shapeNet.py:243-251: 6D of R_q R_ref^-1 and of its inverse, float32.  Poses are 3x3 or 4x4 (numpy or torch)."""
    q = np.asarray(query_pose, dtype=np.float64)
    r = np.asarray(ref_pose, dtype=np.float64)
    rel = q[:3, :3] @ np.linalg.inv(r)[:3, :3]
    inv = r[:3, :3] @ np.linalg.inv(q)[:3, :3]
    return (matrix_to_rotation_6d(torch.tensor(rel, dtype=torch.float32)),
            matrix_to_rotation_6d(torch.tensor(inv, dtype=torch.float32)))


def all_relative_poses(template_poses, ref_pose) -> torch.Tensor:
    """shapeNet.py:300-306 for a whole grid at once: (N,6) float32."""
    t = np.asarray(template_poses, dtype=np.float64)[:, :3, :3]
    r = np.linalg.inv(np.asarray(ref_pose, dtype=np.float64))[:3, :3]
    return matrix_to_rotation_6d(torch.tensor(t @ r, dtype=torch.float32))


# ---- utils.py:14-20, 44-47, 290-356: nearest template on the viewing sphere (+ in-plane rotation) ---------------------------------
_CV2GL = np.diag([1.0, -1.0, -1.0, 1.0])


def opencv2opengl(pose: np.ndarray) -> np.ndarray:
    """utils.py:14-20: flip the y and z ROWS of one 4x4 pose or a stack of them."""
    return np.matmul(_CV2GL, np.asarray(pose))


def geodesic_numpy(R1: np.ndarray, R2: np.ndarray) -> float:
    """utils.py:44-47: rotation angle between two 3x3 rotations, degrees."""
    c = (np.trace(np.asarray(R2) @ np.asarray(R1).T) - 1.0) / 2.0
    return float(np.degrees(np.arccos(np.clip(c, -1.0, 1.0))))


def inplane_of_rotation(delta: np.ndarray) -> float:
    """utils.py:290-292: first angle (degrees) of scipy's INTRINSIC-free "zyx" Euler decomposition of a rotation matrix."""
    from scipy.spatial.transform import Rotation
    return float(Rotation.from_matrix(np.asarray(delta)).as_euler("zyx", degrees=True)[0])


def inplane_rotation(inplane_deg: float) -> np.ndarray:
    """utils.py:295-297: rotation about z by MINUS the in-plane angle."""
    a = np.radians(-inplane_deg)
    c, s = np.cos(a), np.sin(a)
    return np.array([[c, -s, 0.0], [s, c, 0.0], [0.0, 0.0, 1.0]])


def compute_inplane(rot_query: np.ndarray, rot_template: np.ndarray) -> float:
    """utils.py:306-315: in-plane angle that turns the template's rotation into the query's (the reference also prints a warning when the
    recovered rotation is >= 15 degrees off: `inplane_residual` returns that angle instead)."""
    return inplane_of_rotation(np.asarray(rot_template) @ np.asarray(rot_query).T)


def inplane_residual(rot_query: np.ndarray, rot_template: np.ndarray, inplane_deg: float) -> float:
    return geodesic_numpy(inplane_rotation(inplane_deg) @ np.asarray(rot_template), rot_query)


class NearestTemplateFinder:
    """utils.py:318-356.  For every query object pose (M,4,4) the template of the grid (`level_templates`, `pose_distribution`) whose
    viewpoint lies nearest on the sphere -- compared through row 2 of the OpenGL-convention poses, as the reference does -- returned as
    its index in the FULL grid (`avail_index`), optionally with the in-plane angle between query and template.  `normalize_query_translation`
    is accepted and stored (the reference never reads it either).  `root`: the reference's predefined_poses directory (None: synthesised grid)."""


    def __init__(self, level_templates, pose_distribution, return_inplane, normalize_query_translation=True, root: Optional[str] = None, grid=None):
        self.level_templates = level_templates
        self.normalize_query_translation = normalize_query_translation
        self.pose_distribution = pose_distribution
        self.return_inplane = return_inplane
        if grid is not None:               # (avail_index, obj_template_poses) handed in: a grid loaded elsewhere (tests: the reference's own level-0 file)
            self.avail_index, self.obj_template_poses = np.asarray(grid[0]), np.asarray(grid[1])
        else:
            self.avail_index, self.obj_template_poses = get_obj_poses_from_template_level(level_templates, pose_distribution, return_cam=False,
                                                                                          return_index=True, root=root)
        self.obj_template_openGL_poses = opencv2opengl(self.obj_template_poses)

    def search_nearest_template(self, obj_query_pose):
        q = np.asarray(obj_query_pose)
        q_loc = opencv2opengl(q)[:, 2, :3]                                     # (M,3)
        t_loc = self.obj_template_openGL_poses[:, 2, :3]                       # (N,3)
        d = np.sqrt(((q_loc[:, None, :] - t_loc[None, :, :]) ** 2).sum(-1))    # scipy.spatial.distance.cdist (euclidean)
        best = np.argmin(d, axis=-1)                                           # first minimum, as numpy
        if not self.return_inplane:
            return self.avail_index[best]
        nearest = self.obj_template_poses[best]
        inplanes = np.zeros(len(q))
        for i in range(len(q)):
            inplanes[i] = compute_inplane(q[i, :3, :3], nearest[i, :3, :3])
        return self.avail_index[best], inplanes
