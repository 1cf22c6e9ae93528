"""CPU: host-side logic, C-ABI surface, weight determinism, sharding helpers."""
import ctypes
import os
import re
import subprocess

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _header_functions():
    src = open(os.path.join(ROOT, "include", "nope_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(nope_[a-z0-9_]+)\s*\(", src)))


def test_capi_exports_every_declared_symbol():
    """libnope_hip.so (gfx950 build) loads and exports exactly what include/nope_hip.h declares.
    No compute call is made here (no GPU in this suite)."""
    from nope_amd import hip
    from nope_amd.csrc import build
    lib = build.build()
    declared = _header_functions()
    assert set(declared) == set(hip.EXPORTED_SYMBOLS)
    out = subprocess.run(["nm", "-D", "--defined-only", lib], capture_output=True, text=True, check=True).stdout
    exported = set(re.findall(r"\bT (nope_[a-z0-9_]+)", out))
    assert set(declared) <= exported, set(declared) - exported
    dll = ctypes.CDLL(lib)
    assert dll.nope_abi_version() == hip.ABI_VERSION
    dll.nope_strerror.restype = ctypes.c_char_p
    assert dll.nope_strerror(-3) == b"workspace too small"


def test_code_object_is_gfx950():
    from nope_amd.csrc import build
    lib = build.build()
    data = open(lib, "rb").read()
    assert b"gfx950" in data and b"gfx942" not in data and b"sm_" not in data


def test_missing_library_error_message(tmp_path):
    from nope_amd import hip
    with pytest.raises(hip.NopeError, match="no CPU fallback"):
        hip.NopeLib(str(tmp_path / "nope.so"))


def test_package_never_imports_oracle_or_emulator():
    for dirpath, _, files in os.walk(os.path.join(ROOT, "nope_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".h")):
                s = open(os.path.join(dirpath, f)).read()
                assert "import oracle" not in s and "from oracle" not in s, f
                assert "hipemu" not in s.replace("tests/hipemu", ""), f


def test_synth_weights_are_a_pure_function():
    from nope_amd.weights import sha256_of, synth_tensor
    a = synth_tensor(2022, "downs.0.0.block1.proj.weight", (192, 192, 3, 3))
    b = synth_tensor(2022, "downs.0.0.block1.proj.weight", (192, 192, 3, 3))
    c = synth_tensor(2023, "downs.0.0.block1.proj.weight", (192, 192, 3, 3))
    assert torch.equal(a, b) and not torch.equal(a, c)
    assert float(a.abs().max()) <= 1 / (192 * 9) ** 0.5 + 1e-7
    assert sha256_of(a) == sha256_of(b)


def test_state_dict_contract():
    """Key names/shapes the reference checkpoints use (SURVEY.md §8 'State-dict keys')."""
    from nope_amd.u_net import UNet
    from tests.util import StubEncoder
    m = UNet(u_net_dim=192, rot_representation_dim=6, encoder=StubEncoder(8), pose_mlp_name="single_layer", pretrained_path=None)
    sd = m.state_dict()
    assert sum(v.numel() for v in sd.values()) == 305_771_144
    want = {
        "pose_mlp.0.weight": (768, 6), "init_conv.weight": (192, 8, 3, 3),
        "downs.0.0.mlp.1.weight": (192, 768), "downs.0.0.block1.proj.weight": (192, 192, 3, 3),
        "downs.0.2.fn.norm.weight": (192,), "downs.0.2.fn.fn.to_qkv.weight": (384, 192, 1, 1),
        "downs.0.2.fn.fn.to_out.1.bias": (192,), "downs.0.3.1.weight": (192, 768, 1, 1),
        "downs.3.3.weight": (1536, 768, 3, 3), "mid_attn.fn.fn.to_out.weight": (1536, 128, 1, 1),
        "ups.0.0.block1.proj.weight": (1536, 2304, 3, 3), "ups.0.0.res_conv.weight": (1536, 2304, 1, 1),
        "ups.0.3.1.weight": (768, 1536, 3, 3), "ups.3.3.weight": (192, 192, 3, 3),
        "final_res_block.res_conv.weight": (192, 384, 1, 1), "final_conv.0.mlp.1.weight": (192, 768),
        "final_conv.1.weight": (8, 192, 1, 1),
    }
    for k, s in want.items():
        assert tuple(sd[k].shape) == s, k
    assert m.channels == 8 and m.out_dim == 8 and m.name == "template" and m.rot_representation_dim == 6


def test_state_dict_contract_of_the_non_default_variants():
    """use_hard_up_down=False (u_net.py:54-59: Conv2d(4, 2, 1) / ConvTranspose2d(4, 2, 1) in slot 3 of a level) and the LDM variant
    with FiLM ResBlocks (openaimodel.py:233-239: emb_layers.1 twice as wide): key names and shapes.  (That the reference classes
    load these state dicts strictly is checked where the fixtures are recorded, tests/golden/make_golden.py.)"""
    from nope_amd.ldm import UNetModelPose
    from nope_amd.u_net import UNet
    from tests.util import StubEncoder
    sd = UNet(u_net_dim=16, rot_representation_dim=6, encoder=StubEncoder(8), pose_mlp_name="single_layer", use_hard_up_down=False).state_dict()
    assert tuple(sd["downs.0.3.weight"].shape) == (16, 16, 4, 4) and tuple(sd["downs.1.3.weight"].shape) == (32, 16, 4, 4)
    assert tuple(sd["ups.0.3.weight"].shape) == (128, 64, 4, 4)        # ConvTranspose2d: [in, out, kh, kw]
    assert tuple(sd["downs.3.3.weight"].shape) == (128, 64, 3, 3) and "downs.0.3.1.weight" not in sd
    kw = dict(injecting_condition_twice=True, pose_mlp_name="single_layer", rot_representation_dim=6, image_size=8, in_channels=8,
              model_channels=32, out_channels=8, num_res_blocks=1, attention_resolutions=[1, 2], channel_mult=(1, 2),
              num_head_channels=32, use_spatial_transformer=True, transformer_depth=1, context_dim=24)
    plain = UNetModelPose(encoder=StubEncoder(8), **kw).state_dict()
    film = UNetModelPose(encoder=StubEncoder(8), use_scale_shift_norm=True, **kw).state_dict()
    assert set(plain) == set(film)
    assert tuple(plain["input_blocks.1.0.emb_layers.1.weight"].shape) == (32, 128) and tuple(film["input_blocks.1.0.emb_layers.1.weight"].shape) == (64, 128)
    assert tuple(film["pose_mlp_timesteps.0.weight"].shape) == (128, 6)
    assert tuple(film["input_blocks.1.1.transformer_blocks.0.attn2.to_v.weight"].shape) == (32, 24)


def test_shard_range_partitions():
    from nope_amd.dist import shard_range
    for n in (0, 1, 5, 64, 341, 512, 8192):
        for ws in (1, 2, 3, 8):
            spans = [shard_range(n, r, ws) for r in range(ws)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            assert max(h - l for l, h in spans) - min(h - l for l, h in spans) <= 1


def test_harness_batch_shapes():
    from nope_amd.harness import geodesic_deg, random_rotations, rotation_6d, synthetic_batch
    b = synthetic_batch(2, 7, 64, seed=1)
    assert b["query"].shape == (2, 3, 64, 64) and b["all_relativeR"].shape == (2, 7, 6)
    assert b["template_poses"].shape == (2, 7, 3, 3) and b["query_pose"].dtype == torch.float64
    assert float(b["query"].min()) >= -1 and float(b["query"].max()) <= 1
    R_ = random_rotations(5, torch.Generator().manual_seed(0))
    assert torch.allclose(R_ @ R_.transpose(-1, -2), torch.eye(3, dtype=torch.float64).expand(5, 3, 3), atol=1e-12)
    assert torch.allclose(torch.linalg.det(R_), torch.ones(5, dtype=torch.float64))
    assert rotation_6d(R_).shape == (5, 6) and torch.equal(rotation_6d(R_)[:, :3], R_[:, 0])
    assert float(geodesic_deg(R_, R_).abs().max()) < 1e-5


def test_unsupported_metric_returns_none():
    """model.py:256,266: any metric other than "l2" silently returns None."""
    from nope_amd.model import PoseConditional
    from nope_amd.u_net import UNet
    from tests.util import StubEncoder
    u = UNet(u_net_dim=8, rot_representation_dim=6, encoder=StubEncoder(8), pose_mlp_name="single_layer")
    m = PoseConditional(u, None, {"similarity_metric": "cosine"}, None)
    assert m.retrieval(torch.zeros(1, 8, 8, 8), torch.zeros(1, 2, 8, 8, 8)) is None


def test_geodesic_error_metric():
    """nope_amd.metrics (loss.py:14-115; pytorch3d restated, parity unpinned): known angles, the linear extrapolation
    near 0 / 180 degrees, the three symmetry branches, top-k dictionary keys and dtypes."""
    import math
    from nope_amd.harness import random_rotations
    from nope_amd.metrics import GeodesicError, acos_linear_extrapolation, so3_relative_angle, so3_relative_angle_with_symmetry

    def rot(axis, deg):
        a = math.radians(deg)
        c, s = math.cos(a), math.sin(a)
        m = {"x": [[1, 0, 0], [0, c, -s], [0, s, c]], "y": [[c, 0, s], [0, 1, 0], [-s, 0, c]], "z": [[c, -s, 0], [s, c, 0], [0, 0, 1]]}[axis]
        return torch.tensor(m, dtype=torch.float64)

    eye = torch.eye(3, dtype=torch.float64)
    for deg in (5.0, 30.0, 90.0, 170.0):
        ang = so3_relative_angle(rot("x", deg)[None], eye[None])
        assert abs(math.degrees(float(ang)) - deg) < 1e-9
    # identical rotations: cos = 1 is beyond the bound -> extrapolated value acos(b) - (1 - b) / sqrt(1 - b^2), not 0
    b = 1.0 - 1e-4
    z = float(so3_relative_angle(eye[None], eye[None]))
    assert abs(z - (math.acos(b) - (1 - b) / math.sqrt(1 - b * b))) < 1e-12 and 0 < z < 0.01
    x = torch.tensor([-1.0, -0.5, 0.0, 0.5, 1.0], dtype=torch.float64)
    y = acos_linear_extrapolation(x)
    assert torch.allclose(y[1:4], torch.acos(x[1:4])) and float(y[0]) > float(y[1]) > float(y[3]) > float(y[4])
    with pytest.raises(ValueError):
        so3_relative_angle(3.0 * eye[None], eye[None], eps=1e-2)
    # symmetry 1: a prediction that is the ground truth turned by 180 degrees about Y scores (almost) zero
    R = random_rotations(6, torch.Generator().manual_seed(4)).double()
    flipped = rot("y", 180)[None] @ R
    sym = torch.tensor([0, 1, 1, 0, 1, 0]).view(-1, 1)
    e = so3_relative_angle_with_symmetry(flipped, R, sym)
    assert bool((e[sym.view(-1) == 1] < 0.02).all()) and bool((e[sym.view(-1) == 0] > 3.0).all())
    # symmetry 2 (circular): turning the object about its own symmetry (Z) axis does not change the error
    spun = R @ rot("z", 77)[None]
    e2 = so3_relative_angle_with_symmetry(spun, R, torch.full((6, 1), 2))
    # (the reference takes acos of an unclamped cosine similarity, loss.py:71-73: exactly equal axes can give NaN)
    assert float(torch.nan_to_num(e2, nan=0.0).abs().max()) < 1e-6
    e3 = so3_relative_angle_with_symmetry(R @ rot("x", 40)[None], R, torch.full((6, 1), 2))
    assert bool(((torch.rad2deg(e3) - 40.0).abs() < 1e-6).all())
    # module: top-1 and top-k forms
    err, res = GeodesicError([15, 30])(flipped, R, sym)
    assert err.dtype == torch.float64 and set(res) == {"top1, accuracy_15", "top1, accuracy_30", "top1, median"}
    predk = torch.stack([flipped, R, R, R, R], 1)
    err, res = GeodesicError([15])(predk, R, torch.zeros(6, 1))
    assert err.dtype == torch.float32 and float(res["top1, accuracy_15"]) == 0.0 and float(res["top3, accuracy_15"]) == 100.0
    assert set(res) == {f"top{k}, {m}" for k in (1, 3, 5) for m in ("accuracy_15", "median")}


def test_geodesic_known_answers_host():
    """Row f2, the angle itself: the torch restatement against hand-computed answers of pytorch3d's published formula (tests/util.py);
    the device kernel is held to the same table in tests/test_gpu_configs.py."""
    from nope_amd.metrics import GeodesicError, so3_relative_angle_with_symmetry
    from tests.util import geodesic_known_answers
    for name, p, g, sym, want in geodesic_known_answers():
        if want == "raises":
            with pytest.raises(ValueError):
                so3_relative_angle_with_symmetry(p[None], g[None], torch.tensor([sym]))
            continue
        got = float(so3_relative_angle_with_symmetry(p[None], g[None], torch.tensor([sym]))[0])
        assert abs(got - want) < (2e-6 if sym == 1 else 1e-12), (name, got, want)
    # the module on the same table: degrees, top-1 form
    rows = [k for k in geodesic_known_answers() if k[4] != "raises"]
    P, G = torch.stack([k[1] for k in rows]), torch.stack([k[2] for k in rows])
    err, _ = GeodesicError([15])(P, G, torch.tensor([k[3] for k in rows]).view(-1, 1))
    want = torch.rad2deg(torch.tensor([k[4] for k in rows], dtype=torch.float64))
    assert float((err - want).abs().max()) < 1e-4


def _check_geodesic_kernel(hip, dev):
    """`nope_op_geodesic` (csrc/kernels_metric.hip) against the hand-computed answers of pytorch3d's published formula (identity -> the
    extrapolated 7.07e-3, NOT 0; 180 degrees about each axis -> pi - 7.07e-3; the extrapolation branches either side of the bound; a
    trace outside [-1 - eps, 3 + eps] -> ValueError; the two symmetric branches), then the gather form (`template_poses[nearest_idx]`
    inside the launch, model.py:352-354) against the torch restatement on the host."""
    from nope_amd.harness import random_rotations
    from nope_amd.metrics import GeodesicError, geodesic_from_indices, so3_relative_angle_with_symmetry
    from tests.util import geodesic_known_answers
    rows = [k for k in geodesic_known_answers() if k[4] != "raises"]
    P, G = torch.stack([k[1] for k in rows]).to(dev), torch.stack([k[2] for k in rows]).to(dev)
    sym = torch.tensor([k[3] for k in rows]).to(dev)
    got = hip.op_geodesic(P[:, None], G, sym)[:, 0].cpu()
    for (name, _, _, s_, want), e in zip(rows, got.tolist()):
        assert abs(e - want) < (2e-6 if s_ == 1 else 1e-12), (name, e, want)
    for name, p, g_, s_, want in geodesic_known_answers():
        if want == "raises":
            with pytest.raises(ValueError):
                hip.op_geodesic(p[None, None].to(dev), g_[None].to(dev), torch.tensor([s_]).to(dev))
    # gather form, every symmetry, shared and per-query grids, against the host restatement
    g = torch.Generator().manual_seed(11)
    B, N, k = 7, 26, 5
    grid = random_rotations(N, g)
    gt = random_rotations(B, g)
    idx = torch.stack([torch.randperm(N, generator=g)[:k] for _ in range(B)])
    symm = torch.tensor([0, 1, 2, 0, 1, 2, 0])
    want = torch.stack([so3_relative_angle_with_symmetry(grid[idx[:, j]], gt, symm) for j in range(k)], 1)
    got = geodesic_from_indices(grid[None].to(dev), idx.to(dev), gt.to(dev), symm.to(dev)).cpu()
    assert float((got - want).abs().max()) < 1e-6
    per_query = grid[None].repeat(B, 1, 1, 1).contiguous()
    assert torch.equal(geodesic_from_indices(per_query.to(dev), idx.to(dev), gt.to(dev), symm.to(dev)).cpu(), got)
    e1, r1 = GeodesicError([15, 30]).from_indices(grid[None].to(dev), idx.to(dev), gt.to(dev), symm.to(dev))
    e2, r2 = GeodesicError([15, 30])(grid[idx], gt, symm)
    assert float((e1.cpu() - e2).abs().max()) < 1e-4 and sorted(r1) == sorted(r2)
    assert all(abs(float(r1[k_]) - float(r2[k_])) < 1e-3 for k_ in r1)
    with pytest.raises(hip.NopeError):           # an index outside the grid is an error, not a read past the buffer
        geodesic_from_indices(grid[None].to(dev), torch.full((B, k), N, dtype=torch.int64).to(dev), gt.to(dev), symm.to(dev))


def test_geodesic_kernel_known_answers_emu(emu):
    """The device kernel's source (csrc/kernels_metric.hip) under the CPU interpreter, same table as on the GPU."""
    _check_geodesic_kernel(emu, "cpu")


def test_geodesic_binding_rejects_mismatched_shapes(emu):
    """hip.op_geodesic: a symmetry tensor or an index tensor whose leading size is not the batch would make the kernel read past it."""
    import pytest
    hip = emu
    poses = torch.eye(3, dtype=torch.float64).expand(2, 4, 3, 3).contiguous()
    gt = torch.eye(3, dtype=torch.float64).expand(2, 3, 3).contiguous()
    assert hip.op_geodesic(poses, gt, torch.zeros(2, 1), torch.zeros(2, 3, dtype=torch.int64)).shape == (2, 3)
    with pytest.raises(hip.NopeError):
        hip.op_geodesic(poses, gt, torch.zeros(1), None)
    with pytest.raises(hip.NopeError):
        hip.op_geodesic(poses, gt, None, torch.zeros(1, 3, dtype=torch.int64))
    with pytest.raises(hip.NopeError):
        hip.op_geodesic(poses, gt, None, torch.zeros(6, dtype=torch.int64))


def test_pose_grids_and_relative_poses(tmp_path):
    """nope_amd.poses (utils.py:72-125, shapeNet.py:243-251): synthesised icosphere grids have the reference's camera
    positions (level-0 fixture, as a set) and its upper-hemisphere counts at every level; reading the reference's files
    from a directory keeps its selection semantics; coarse grids index into fine ones; relative-pose identities."""
    import os
    import numpy as np
    from nope_amd import poses as P
    fx = np.load(os.path.join(os.path.dirname(__file__), "golden", "pose_grid_level0.npz"))
    cams, objs = P.synthesize_grid(0)
    ref_pos = fx["sphere_poses_level0"][:, :3, 3]
    d = np.linalg.norm(cams[:, None, :3, 3] - ref_pos[None], axis=-1)
    assert d.min(axis=1).max() < 2e-5 and len(set(d.argmin(axis=1))) == 42          # same 42 viewpoints
    for c, o in zip(cams, objs):                                                    # proper look-at poses
        assert np.allclose(c[:3, :3] @ c[:3, :3].T, np.eye(3), atol=1e-12) and abs(np.linalg.det(c[:3, :3]) - 1) < 1e-12
        assert np.allclose(c[:3, :3].T @ (-c[:3, 3]), [0, 0, 1], atol=1e-12)          # +z looks at the object
        assert np.allclose(o[:3, :3], c[:3, :3].T) and np.allclose(np.linalg.norm(o[:3, 3]), 0.5)
    assert np.allclose(np.linalg.norm(fx["obj_poses_level0"][:, :3, 3], axis=1), 0.5, atol=1e-6)
    for level, (n, up) in {0: (42, 26), 1: tuple(fx["count_level1"]), 2: tuple(fx["count_level2"]), 3: tuple(fx["count_level3"])}.items():
        assert len(P.get_obj_poses_from_template_level(level, "all")) == n
        idx, sel = P.get_obj_poses_from_template_level(level, "upper", return_index=True)
        assert len(sel) == up and len(idx) == up
    # the reference's own files, read from a directory
    for k in ("sphere_poses_level0", "obj_poses_level0", "idx_upper_level0_in_level2"):
        np.save(tmp_path / f"{k}.npy", fx[k])
    idx, sel = P.get_obj_poses_from_template_level(0, "upper", return_index=True, root=str(tmp_path))
    want = fx["sphere_poses_level0"][:, 2, 3] >= 0
    assert np.array_equal(idx, np.arange(42)[want]) and np.array_equal(sel, fx["obj_poses_level0"][want])
    assert np.array_equal(P.load_index_level0_in_level2("upper", root=str(tmp_path)), fx["idx_upper_level0_in_level2"])
    assert P.load_mapping_id_templates_to_idx_pose_distribution(0, "upper", root=str(tmp_path))[int(idx[3])] == 3
    # synthesised: level-0 viewpoints are found inside level 2
    i02 = P.load_index_level0_in_level2("upper")
    c0 = P.get_obj_poses_from_template_level(0, "upper", return_cam=True)
    c2 = P.get_obj_poses_from_template_level(2, "upper", return_cam=True)
    assert len(i02) == 26 and np.allclose(c2[i02][:, :3, 3], c0[:, :3, 3])
    # relative poses
    R = objs[:, :3, :3]
    rel, inv = P.compute_relative_pose(objs[7], objs[3])
    assert rel.dtype == torch.float32 and rel.shape == (6,)
    assert np.allclose(rel.numpy(), (R[7] @ R[3].T)[:2].reshape(6), atol=1e-6) and np.allclose(inv.numpy(), (R[3] @ R[7].T)[:2].reshape(6), atol=1e-6)
    allr = P.all_relative_poses(objs, objs[3])
    assert allr.shape == (42, 6) and torch.allclose(allr[7], rel, atol=1e-6)
    assert torch.allclose(allr[3], torch.tensor([1.0, 0, 0, 0, 1, 0]), atol=1e-6)


def test_handles_are_dropped_by_parent_load_state_dict(monkeypatch):
    """`PoseConditional.load_state_dict` (or any parent's) never calls a child's `load_state_dict`, so the
    cached device handles of the U-Net and the encoder must be invalidated from post hooks; in-place parameter
    writes are caught through the tensors' version counters."""
    from nope_amd import hip
    from nope_amd.encoder import FeatureExtractor
    from nope_amd.model import PoseConditional
    from nope_amd.u_net import UNet
    made = {"unet": 0, "enc": 0}

    class FakeU:
        def __init__(self, cfg, sd, dt): made["unet"] += 1
    class FakeE:
        def __init__(self, d, sd, dt, bn_eps=1e-5): made["enc"] += 1
    monkeypatch.setattr(hip, "UNetHandle", FakeU)
    monkeypatch.setattr(hip, "EncoderHandle", FakeE)
    enc = FeatureExtractor(8)
    u = UNet(u_net_dim=8, rot_representation_dim=6, encoder=enc, pose_mlp_name="single_layer")
    m = PoseConditional(u, None, {"similarity_metric": "l2"}, None)
    dev = torch.device("cpu")
    h1, e1 = u._get_handle(dev), enc._get_handle(dev)
    assert u._get_handle(dev) is h1 and enc._get_handle(dev) is e1 and made == {"unet": 1, "enc": 1}
    m.load_state_dict(m.state_dict())                       # parent load: both handles must go
    assert u._get_handle(dev) is not h1 and enc._get_handle(dev) is not e1 and made == {"unet": 2, "enc": 2}
    h2, e2 = u._get_handle(dev), enc._get_handle(dev)
    u.load_state_dict(u.state_dict())                       # has encoder.* keys: the encoder handle goes too
    assert u._get_handle(dev) is not h2 and enc._get_handle(dev) is not e2
    h3, e3 = u._get_handle(dev), enc._get_handle(dev)
    with torch.no_grad():
        u.init_conv.weight.mul_(2.0)                        # in-place write without any load
        enc.backbone.bn1.running_mean.add_(1.0)
    assert u._get_handle(dev) is not h3 and enc._get_handle(dev) is not e3
    h4 = u._get_handle(dev)
    enc._get_handle(dev)
    u.invalidate()                                          # cascades to the encoder
    assert enc._handle is None and u._get_handle(dev) is not h4


def _check_metric_against_reference(dev):
    """nope_amd.metrics.GeodesicError against tests/golden/metric_ref.npz -- outputs of the REFERENCE's loss.py:14-115 run in the
    build container with pytorch3d's angle bound to the restated function (tests/golden/make_golden_f2f3.py): the three
    symmetry branches, roty180's f32 round trip, the float64 casts and the result dictionaries are the reference's own."""
    import numpy as np
    from nope_amd import metrics as M
    z = np.load(os.path.join(ROOT, "tests", "golden", "metric_ref.npz"))
    pred, gt = torch.from_numpy(z["pred"]).to(dev), torch.from_numpy(z["gt"]).to(dev)
    assert np.array_equal(M._roty180("cpu", torch.float32).numpy(), z["roty180"])
    for tag in ("mixed", "none", "two", "circle", "none_two"):
        sym = torch.from_numpy(z[f"{tag}/symmetry"]).to(dev)
        pr = torch.from_numpy(z["circle/pred"]).to(dev) if tag == "circle" else pred      # (no exact matches in the unclamped-acos branch)
        for form, p in (("topk", pr), ("top1", pr[:, 0])):
            err, res = M.GeodesicError([15, 30])(p, gt, sym)
            want = torch.from_numpy(z[f"{tag}/{form}/error"])
            assert err.dtype == want.dtype and err.device.type == torch.device(dev).type
            assert torch.allclose(err.cpu(), want, rtol=0, atol=1e-4 if dev != "cpu" else 1e-9, equal_nan=True), (tag, form)
            keys = sorted(res)
            assert keys == list(z[f"{tag}/{form}/keys"])
            for k, v in zip(keys, z[f"{tag}/{form}/values"]):
                got = float(res[k])
                assert (v != v and got != got) or abs(got - v) <= (1e-3 if dev != "cpu" else 1e-9), (tag, form, k, got, v)
    e = M.so3_relative_angle_with_symmetry(pred[:, 1].double(), gt.double(), torch.from_numpy(z["mixed/symmetry"]).to(dev))
    assert torch.allclose(e.cpu(), torch.from_numpy(z["mixed/helper_rad"]), rtol=0, atol=1e-6 if dev != "cpu" else 1e-12, equal_nan=True)


def test_geodesic_error_against_reference_run():
    _check_metric_against_reference("cpu")


def test_poses_and_crop_geometry_against_reference_run(tmp_path):
    """nope_amd.poses / nope_amd.dataset against tests/golden/poses_ref.npz: outputs of the reference's own
    get_obj_poses_from_template_level / load_mapping / perspective / crop_frame (up to the cv2 call: the four source and target
    points it hands to cv2.getPerspectiveTransform) / ShapeNet.compute_relative_pose (utils.py:50-57,72-125,204-260;
    shapeNet.py:243-251).  The reference's grid FILES do not travel: the level-0 grid recorded in the fixture is written to a
    directory and read back through `root=`, which must reproduce the reference's selection bit for bit; the synthesised
    grids have the same viewpoints in another order (test_pose_grids_and_relative_poses)."""
    import numpy as np
    from nope_amd import dataset as D
    from nope_amd import poses as P
    w = np.load(os.path.join(ROOT, "tests", "golden", "poses_ref.npz"))
    np.save(tmp_path / "sphere_poses_level0.npy", w["L0/all/cam_poses"])
    np.save(tmp_path / "obj_poses_level0.npy", w["L0/all/obj_poses"])
    for dist in ("upper", "all"):
        idx, poses = P.get_obj_poses_from_template_level(0, dist, return_index=True, root=str(tmp_path))
        assert np.array_equal(idx, w[f"L0/{dist}/index"]) and np.array_equal(poses, w[f"L0/{dist}/obj_poses"])
        assert np.array_equal(P.get_obj_poses_from_template_level(0, dist, return_cam=True, root=str(tmp_path)), w[f"L0/{dist}/cam_poses"])
        m = P.load_mapping_id_templates_to_idx_pose_distribution(0, dist, root=str(tmp_path))
        assert np.array_equal(np.array(sorted(m.items()), dtype=np.int64), w[f"mapping_L0/{dist}"])
    for level in (0, 1):          # synthesised grids: same number of selected viewpoints as the reference's files
        for dist in ("upper", "all"):
            assert len(P.get_obj_poses_from_template_level(level, dist)) == len(w[f"L{level}/{dist}/index"])
    for level in (2, 3):
        assert len(P.get_obj_poses_from_template_level(level, "upper")) == int(w[f"L{level}/upper/count"][0])
    assert len(P.load_index_level0_in_level2("upper")) == len(w["idx_level0_in_level2/upper"])
    assert np.array_equal(D.perspective(w["perspective/K"], w["perspective/pose"], w["perspective/pts"]), w["perspective/out"])
    for row, src, dst in zip(w["crop/in"], w["crop/src"], w["crop/dst"]):
        pose, inplane, vb, size = row[:16].reshape(4, 4), bool(row[16]), float(row[17]), int(row[18])
        # the four image points the reference projects (int32-truncated) are the ones crop_transform maps onto the output square
        origin = (pose @ np.array([0, 0, 0, 1.0]))[:3]
        Mx = D.crop_transform(w["perspective/K"], pose, size, inplane, vb)
        h = np.concatenate([src, np.ones((4, 1))], 1) @ Mx.T
        assert np.abs(h[:, :2] / h[:, 2:] - dst).max() < 1e-9 and origin[2] > 0
    rel, inv = P.compute_relative_pose(w["relpose/query"], w["relpose/ref"])
    assert np.array_equal(rel.numpy(), w["relpose/rel"]) and np.array_equal(inv.numpy(), w["relpose/rel_inv"])
    allr = P.all_relative_poses(w["L0/upper/obj_poses"], w["relpose/ref"])
    assert np.abs(allr.numpy() - w["relpose/all"]).max() < 1e-6


def test_abi_version_is_checked_at_load(monkeypatch):
    """nope_unet_config grew a field and the enums new meanings; a binding written against another header version must
    fail at load time instead of passing shorter structs (hip.NopeLib compares nope_abi_version() with its own ABI_VERSION)."""
    from nope_amd import hip
    from nope_amd.csrc import build
    lib = build.build()
    assert hip.NopeLib(lib).dll.nope_abi_version() == hip.ABI_VERSION
    src = open(os.path.join(ROOT, "include", "nope_hip.h")).read()
    assert int(re.search(r"#define NOPE_ABI_VERSION (\d+)", src).group(1)) == hip.ABI_VERSION
    monkeypatch.setattr(hip, "ABI_VERSION", hip.ABI_VERSION + 1)
    with pytest.raises(hip.NopeError, match="ABI version"):
        hip.NopeLib(lib)


def test_compute_mode_codes():
    from nope_amd import hip
    src = open(os.path.join(ROOT, "include", "nope_hip.h")).read()
    for name, code in (("NOPE_F32", hip.F32), ("NOPE_BF16", hip.BF16), ("NOPE_F16", hip.F16), ("NOPE_BF16X3", hip.BF16X3)):
        assert re.search(rf"\b{name} = {code}\b", src), name
    assert [hip.dtype_code(s) for s in ("f32", "bf16", "f16", "bf16x3")] == [0, 1, 2, 3]
    assert hip.torch_dtype(hip.BF16X3) == torch.float32 and hip.storage_code(hip.BF16X3) == hip.F32 and hip.storage_code(hip.F16) == hip.F16
    with pytest.raises(hip.NopeError):
        hip.dtype_code("fp8")


def test_nearest_template_finder_vs_reference_run():
    """nope_amd.poses.NearestTemplateFinder (utils.py:318-356) against the reference's own class run in the build container
    (tests/golden/make_golden_f2f3.py -> nearest_ref.npz, which carries the grid it searched): same template index for every query, same
    in-plane angle; an exact grid pose finds itself with in-plane 0, the same pose turned by 37 degrees in the image plane finds it with -37
    ... whatever sign the reference's Euler convention gives (recorded, not assumed); the synthesised grid finds the same camera POSITIONS."""
    import os
    import numpy as np
    from nope_amd import poses as P
    fx = np.load(os.path.join(os.path.dirname(__file__), "golden", "nearest_ref.npz"))
    for tag, level, dist in (("L0_upper", 0, "upper"), ("L1_all", 1, "all")):
        grid = (fx[f"{tag}/avail_index"], fx[f"{tag}/obj_template_poses"])
        f = P.NearestTemplateFinder(level, dist, return_inplane=True, grid=grid)
        idx, inp = f.search_nearest_template(fx[f"{tag}/query"])
        assert np.array_equal(idx, fx[f"{tag}/index"])
        assert np.allclose(inp, fx[f"{tag}/inplane"], atol=1e-9)
        assert np.array_equal(P.NearestTemplateFinder(level, dist, return_inplane=False, grid=grid).search_nearest_template(fx[f"{tag}/query"]),
                              fx[f"{tag}/index_only"])
        assert idx[0] == grid[0][3] and abs(inp[0]) < 1e-9 and idx[1] == grid[0][5] and abs(abs(inp[1]) - 37.0) < 1e-6
        # the in-plane angle really recovers the query's rotation for the two constructed cases
        for i, t in ((0, 3), (1, 5)):
            assert P.inplane_residual(fx[f"{tag}/query"][i, :3, :3], grid[1][t][:3, :3], inp[i]) < 0.05      # degrees (the grid files are float32: acos near 1)
        # the synthesised grid (no reference files): the nearest VIEWPOINT is the same point of the sphere
        fs = P.NearestTemplateFinder(level, dist, return_inplane=False)
        got = fs.search_nearest_template(fx[f"{tag}/query"])
        cam_ref = np.linalg.inv(grid[1])[:, :3, 3]
        cam_syn = np.linalg.inv(fs.obj_template_poses)[:, :3, 3]
        pos_ref = cam_ref[np.searchsorted(grid[0], fx[f"{tag}/index"])]
        pos_syn = cam_syn[np.searchsorted(fs.avail_index, got)]
        unit = lambda v: v / np.linalg.norm(v, axis=-1, keepdims=True)
        assert float(np.abs(unit(pos_ref) - unit(pos_syn)).max()) < 1e-4


def test_tuning_switches_are_cached_and_reloaded(emu, capfd):
    """The library reads a NOPE_* switch once per call site (nope_tuning_reload, include/nope_hip.h); the binding reloads when it sees the
    environment change between two calls.  Seen through NOPE_CONV_TRACE's per-launch line: the same 3x3 conv goes to the tap-resident kernel
    under NOPE_CONV_PP=13 and to the per-tap ping-pong kernel under 29, in one process; a change the binding is NOT told about (putenv behind
    its back, no call through hip.lib()) is not seen until nope_tuning_reload()."""
    import ctypes
    import os
    hip = emu
    g = torch.Generator().manual_seed(2)
    x, w = torch.randn(1, 32, 4, 4, generator=g), torch.randn(8, 32, 3, 3, generator=g) / 17
    os.environ["NOPE_CONV_TRACE"] = "1"
    try:
        seen = []
        for pp in ("13", "29", "13"):
            os.environ["NOPE_CONV_PP"] = pp
            capfd.readouterr()
            hip.op_conv(0, hip.to_nhwc(x, 0), w, None)
            seen.append("halo256" in capfd.readouterr().err)
        assert seen == [True, False, True], seen
        # behind the binding's back: the cached value stays until the reload entry point is called
        libc = ctypes.CDLL(None)
        libc.setenv(b"NOPE_CONV_PP", b"29", 1)
        xs, pw = hip.to_nhwc(x, 0), hip.pack_conv_weight(w, 0)[0]
        out = torch.empty(1, 4, 4, 8)
        call = lambda: hip.lib().dll.nope_op_conv(0, xs.data_ptr(), 32, 1, None, 0, 1, 4, 4, 0, 9, pw.data_ptr(), None, None, out.data_ptr(), 8, 1, 0, 0, 0, None)
        dll = hip.lib().dll              # (os.environ still says 13: the binding sees no change and does not reload)
        capfd.readouterr()
        assert dll.nope_op_conv(0, xs.data_ptr(), 32, 1, None, 0, 1, 4, 4, 0, 9, pw.data_ptr(), None, None, out.data_ptr(), 8, 1, 0, 0, 0, None) == 0
        assert "halo256" in capfd.readouterr().err
        dll.nope_tuning_reload()
        assert dll.nope_op_conv(0, xs.data_ptr(), 32, 1, None, 0, 1, 4, 4, 0, 9, pw.data_ptr(), None, None, out.data_ptr(), 8, 1, 0, 0, 0, None) == 0
        assert "halo256" not in capfd.readouterr().err
    finally:
        os.environ.pop("NOPE_CONV_TRACE", None)
        os.environ.pop("NOPE_CONV_PP", None)
        hip.lib().dll.nope_tuning_reload()


def test_bench_power_ceiling_record(tmp_path, monkeypatch):
    """bench.py's `roofline.power_ceiling`: the probe's JSON line + the kernel's fractions of it; None (not an exception) when the probe
    is missing or cannot run (no GPU here)."""
    import json
    import stat
    import bench
    rf = {"achieved": 500.0, "achieved_pass_equivalents": 1000.0}      # (algorithmic rate; x MFMA passes per product: what the probe ratios use)
    assert bench.power_ceiling(rf, "f16x2") is None or isinstance(bench.power_ceiling(rf, "f16x2"), dict)      # (the real probe: no device in this container)
    fake_root = tmp_path / "repo"
    (fake_root / "tools" / "probes").mkdir(parents=True)
    monkeypatch.setattr(bench, "ROOT", str(fake_root))
    assert bench.power_ceiling(rf, "f16x2") is None                                                            # no probe at all
    rec = {"mfma_from_registers": {"f16": {"tflops": 1600, "sclk_mhz": 1600}, "bf16": {"tflops": 1800, "sclk_mhz": 1800},
                                   "f16x2": {"tflops": 2000, "sclk_mhz": 1900}}, "pingpong_skeleton_f16": {"tflops": 1250, "sclk_mhz": 1500}}
    exe = fake_root / "tools" / "probes" / "overlap_probe"
    exe.write_text("#!/bin/sh\necho some banner\necho '" + json.dumps(rec) + "'\n")
    exe.chmod(exe.stat().st_mode | stat.S_IEXEC)
    out = bench.power_ceiling(rf, "f16x2")
    assert out["instruction_mix"] == "f16x2" and abs(out["frac_of_registers_only"] - 0.5) < 1e-12 and abs(out["frac_of_skeleton_f16"] - 0.8) < 1e-12
    assert bench.power_ceiling(rf, "bf16x3")["instruction_mix"] == "bf16"
    assert "frac_of_registers_only" not in bench.power_ceiling(rf, "f32")                                      # exact-f32 MFMA: no such record
    exe.write_text("#!/bin/sh\necho not json\n")
    assert bench.power_ceiling(rf, "f16x2") is None
