"""GPU: the BASELINE.json configurations that round 1 left without a test (configs[2], [3], [4]), `PoseConditional.sample`
and the evaluation harness end to end.

  configs[2]  batch = 32 queries x 512 templates, one GPU, per compute mode       test_config2_batch32_x_512[f16|bf16x3|bf16]
  configs[3]  batch = 32 x 4096 templates sharded 8-way, WHOLE (8 ranks)          test_whole_configs_3_and_4_eight_ranks[4096-bf16-bf16]
  configs[4]  fp16 embeddings + 8192-template bank, WHOLE (8 ranks)               test_whole_configs_3_and_4_eight_ranks[8192-f16-f16]
The sharded configurations run with all their ranks sharing the one GPU of the test box (gloo carries the score all-gather: RCCL
wants one device per rank; the 8-GPU RCCL run is the driver's) and are compared with the unsharded call: same launches, same
bits."""
import os
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import nope_ref as R
from tests.util import MODE_BOUNDS, NORTH_STAR_SCORE_TOL, TOLERANCE_MODES, cached_model, rel

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def model_f32(gpu):
    from tests.util import cached_model
    return cached_model("f32", "f32")


@pytest.mark.parametrize("cdt", ["f16", "f16x2", "bf16x3", "bf16"])
def test_config2_batch32_x_512(model_f32, cdt):
    """BASELINE configs[2] end to end: 32 queries x 512 templates (16384 pose hypotheses) through encoder, U-Net, scoring and
    top-5, per compute mode.  (i) EVERY score and the top-5 against the reference's own recorded run of this batch (cfg2_scores.npz);
    (ii) against the f32 parity mode (itself pinned to the reference at configs[0] / configs[1]):
        f16    -- the benchmark's default 16-bit mode: the best template equals the f32 mode's for ALL 32 queries;
        f16x2, bf16x3 -- the fast modes inside north_star's tolerance: scores within 1e-4 relative, top-5 bit-exact for all 32 queries;
        bf16   -- 8 significand bits: reported; it does NOT meet the top-1 bar (31 of 32 on this input: the smallest f32 top-1 gap,
                  1.3e-4 of the score scale, is far below its 6e-3 score error), which is why it is not the default mode; its winner
                  is always inside the f32 top-5."""
    from nope_amd.harness import build_model, synthetic_batch
    bank_dt = cdt if cdt in ("f16", "bf16") else "f32"
    m = cached_model(cdt, bank_dt)
    b = synthetic_batch(32, 512, 256, seed=77, device="cuda")
    sim, idx, bank = m.generate_and_retrieve(b["query"], b["reference"], b["all_relativeR"])
    torch.cuda.synchronize()
    assert sim.shape == (32, 512) and idx.shape == (32, 5) and bank.shape == (32, 512, 8, 32, 32) and bool(torch.isfinite(sim).all())
    # (i) ALL 16384 scores against the reference's own run of this batch (tests/golden/cfg2_scores.npz: the imported reference's encoder,
    #     UNet.forward for every template, PoseConditional.retrieval -- make_golden_cfg12.py), and its top-5
    import numpy as np
    from nope_amd.weights import sha256_of
    g = np.load(os.path.join(ROOT, "tests", "golden", "cfg2_scores.npz"))
    assert tuple(int(v) for v in g["batch"]) == (32, 512, 256, 77)
    assert sha256_of(b["query"]) == str(g["sha_query"]) and sha256_of(b["all_relativeR"]) == str(g["sha_poses"])
    sim_ref, idx_ref = torch.from_numpy(g["sim"]).cuda(), torch.from_numpy(g["idx"]).cuda()
    e_ref = float((sim - sim_ref).abs().max() / sim_ref.abs().max())
    e_map = rel(bank[:4, :2].float().cpu(), torch.from_numpy(g["bank_first"]))
    same1_ref = int((idx[:, 0] == idx_ref[:, 0]).sum())
    same5_ref = int((idx == idx_ref).all(dim=1).sum())
    print(f"configs[2] {cdt} vs the REFERENCE's recorded run: all 16384 scores rel err {e_ref:.3e}; embedding maps (8 recorded) {e_map:.3e}; "
          f"top-1 equal for {same1_ref}/32 queries, top-5 (ordered) for {same5_ref}/32")
    tol_map, tol_score = MODE_BOUNDS[cdt]
    assert e_map < tol_map and e_ref < tol_score
    if cdt in TOLERANCE_MODES:
        assert e_ref < NORTH_STAR_SCORE_TOL and same1_ref == 32 and same5_ref == 32
    elif cdt == "f16":
        assert same1_ref == 32
    # (ii) against the f32 parity mode
    sim32, idx32, _ = model_f32.generate_and_retrieve(b["query"], b["reference"], b["all_relativeR"])
    err = float((sim - sim32).abs().max()) / float(sim32.abs().max())
    same1 = int((idx[:, 0] == idx32[:, 0]).sum())
    same5 = int((idx == idx32).all(dim=1).sum())
    top2 = sim32.topk(2, dim=1).values
    print(f"configs[2] {cdt} vs f32: score rel err {err:.2e}; top-1 equal for {same1}/32 queries, top-5 (ordered) equal for {same5}/32; smallest "
          f"f32 top-1 gap {float((top2[:, 0] - top2[:, 1]).min()) / float(sim32.abs().max()):.2e} of the score scale")
    assert err < MODE_BOUNDS[cdt][1]
    if cdt == "f16":
        assert same1 == 32
    elif cdt in TOLERANCE_MODES:
        assert same1 == 32 and same5 == 32 and err < NORTH_STAR_SCORE_TOL
    else:
        assert same1 >= 28
    for q in range(32):                     # wherever a mode's winner differs, it is among the f32 top-5
        assert int(idx[q, 0]) in idx32[q].tolist()


def test_sample_vs_oracle(model_f32):
    """PoseConditional.sample (model.py:113-124): u_net(encode_image(reference), relativeR), no decoder for the template encoder."""
    g = torch.Generator().manual_seed(31)
    ref = torch.rand(2, 3, 64, 64, generator=g) * 2 - 1
    pose = torch.randn(2, 6, generator=g)
    pred, rgb = model_f32.sample(ref.cuda(), pose.cuda())
    assert rgb is None and pred.shape == (2, 8, 8, 8)
    enc_sd = {k: v.detach().cpu() for k, v in model_f32.u_net.encoder.state_dict().items()}
    sd = {k: v.detach().cpu() for k, v in model_f32.u_net.own_state_dict().items()}
    want = R.unet_forward(sd, R.encode_image(enc_sd, ref), pose)
    e = rel(pred.cpu(), want)
    print("sample() f32 rel err", e)
    assert e < MODE_BOUNDS["f32"][0]


def test_harness_eval_geodesic_config1(model_f32, golden, tmp_path):
    """The harness body (counterpart of the missing test_shapeNet.py -> PoseConditional.eval_geodesic, model.py:268-376) on
    BASELINE configs[0]'s batch: top-5 indices and loss equal the values recorded from the reference; the accuracy
    dictionary is what the retrieved template poses imply; the saved prediction file holds query_pose + similarity."""
    import numpy as np
    from nope_amd.harness import eval_geodesic, geodesic_deg, main, synthetic_batch
    g = golden("pipeline_cfg1.npz")
    batch = synthetic_batch(1, 64, 128, seed=2022, device="cuda")
    assert torch.equal(batch["query"].cpu(), g["query"]) and torch.equal(batch["all_relativeR"].cpu(), g["all_relativeR"])
    save = str(tmp_path / "pred_step0_rank0")
    sim, idx, res = eval_geodesic(model_f32, batch, save_path=save)
    assert torch.equal(idx.cpu(), g["idx"]) and rel(sim.cpu(), g["sim"]) < MODE_BOUNDS["f32"][1]
    assert abs(res["loss"] - float(g["loss"])) < 1e-4 * abs(float(g["loss"]))
    assert set(res) == {"loss"} | {f"top{k}, {m}" for k in (1, 3, 5) for m in ("accuracy_15", "accuracy_30", "median")}
    err = geodesic_deg(batch["template_poses"][0][idx[0]].cpu(), batch["query_pose"].cpu().expand(5, -1, -1))   # (5,) degrees
    for k in (1, 3, 5):
        best = float(err[:k].min())
        assert res[f"top{k}, accuracy_15"] == (100.0 if best <= 15 else 0.0) and res[f"top{k}, accuracy_30"] == (100.0 if best <= 30 else 0.0)
        assert abs(res[f"top{k}, median"] - best) < 2e-2
    z = np.load(save + ".npz")
    assert z["similarity"].shape == (1, 64) and z["query_pose"].shape == (1, 3, 3)
    # the command-line entry point (python -m nope_amd.harness), small shape
    main(["--batch", "2", "--templates", "6", "--size", "64", "--save-dir", str(tmp_path / "run")])
    assert os.path.isdir(tmp_path / "run" / "predictions")


def _whole_config_worker(rank, ws, port, n_total, cdt, bank_dtype, ret):
    """One of `ws` gloo ranks sharing GPU 0: the whole BASELINE configuration (32 queries x n_total templates), not its per-rank
    shape -- sharded template generation, scoring into the collective's send buffer, all-gather, top-5."""
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=ws)
    try:
        from nope_amd import dist as nd
        from nope_amd.harness import build_model, synthetic_batch
        torch.cuda.set_device(0)
        torch.set_num_threads(2)                              # eight ranks share the host's cores too
        B = 32
        b = synthetic_batch(B, n_total, 256, seed=93, device="cuda")
        m = build_model(compute_dtype=cdt, bank_dtype=bank_dtype, device="cuda", template_parallel=True)
        tdt = {"bf16": torch.bfloat16, "f16": torch.float16}[bank_dtype]
        lo, hi = nd.shard_range(n_total, rank, ws)
        plant = n_total - 7                                   # a slot of the LAST rank's shard
        # The ranks take turns on the one GPU for the heavy part (8 processes time-slicing a device pay a context switch per
        # kernel: 6 minutes instead of one); on 8 GPUs they would run side by side.  Nothing is exchanged in this phase.
        for turn in range(ws):
            if turn == rank:
                # the query embedding rounded to the bank's storage type, so that a planted copy of it scores exactly -0.0
                q_feat = m.u_net.encoder.encode_image(b["query"], mode="mode").to(tdt).float()
                bank, _, _ = m.generate_templates(b["reference"], b["all_relativeR"], None)
                if lo <= plant < hi:
                    bank[:, plant - lo] = q_feat.to(tdt)      # (in place: the bank keeps its shard placement)
                torch.cuda.synchronize()
            dist.barrier()
        sim, idx = m.retrieval_from_feat(q_feat, bank)            # scoring into the send buffer + the all-gather + top-5: all ranks at once
        torch.cuda.synchronize()
        out = {"shape": tuple(bank.shape), "sim": sim.cpu(), "idx": idx.cpu(), "range": (lo, hi)}
        if rank == 0:
            # the unsharded call on one rank: same per-launch hypothesis batches (one reference image x 512 poses), so the same bits
            m.template_parallel = False
            bank1, _, _ = m.generate_templates(b["reference"], b["all_relativeR"], None)
            bank1[:, plant] = q_feat.to(tdt)
            sim1, idx1 = m.retrieval_from_feat(q_feat, bank1)
            torch.cuda.synchronize()
            out["sim1"], out["idx1"] = sim1.cpu(), idx1.cpu()
            # oracle spot check: (b, n) pairs spread over the shards of six different ranks
            per = n_total // ws
            pairs = [(0, 1), (0, per + 3), (13, 2 * per + 5), (13, 4 * per + 7), (31, 5 * per + 11), (31, 7 * per + 1)]
            enc_sd = {k: v.detach().cpu() for k, v in m.u_net.encoder.state_dict().items()}
            sd = {k: v.detach().cpu() for k, v in m.u_net.own_state_dict().items()}
            worst = 0.0
            for bb in sorted({p[0] for p in pairs}):
                ns = [n for (x, n) in pairs if x == bb]
                ref_feat = R.encode_image(enc_sd, b["reference"][bb:bb + 1].cpu())
                want = R.generate_templates(sd, ref_feat, b["all_relativeR"][bb:bb + 1, ns].cpu())
                s_want = R.similarity_scores(q_feat[bb:bb + 1].cpu(), want)
                worst = max(worst, float(((sim1[bb, ns].cpu() - s_want[0]).abs() / s_want[0].abs()).max()))
            out["oracle_pairs"], out["oracle_err"] = pairs, worst
        ret[rank] = out
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("n_total,cdt,bank_dtype", [(4096, "bf16", "bf16"), (8192, "f16", "f16")])
def test_whole_configs_3_and_4_eight_ranks(gpu, n_total, cdt, bank_dtype):
    """BASELINE configs[3] (batch 32 x 4096 templates sharded 8-way, bf16) and configs[4] (fp16 embeddings, 8192-template bank,
    8 ranks) WHOLE: eight gloo ranks share this GPU (RCCL needs one device per rank; the 8-GPU run is the driver's).
    (i) every rank ends up with the same gathered (32, N) similarity and top-5, bit-identical to the unsharded call;
    (ii) six (b, n) scores from the shards of six different ranks match the CPU oracle; (iii) an exact copy of each query's
    embedding planted in the LAST rank's shard scores -0.0 and wins on every rank."""
    ws = 8
    mgr = mp.Manager()
    ret = mgr.dict()
    port = 29900 + os.getpid() % 1000
    mp.spawn(_whole_config_worker, args=(ws, port, n_total, cdt, bank_dtype, ret), nprocs=ws, join=True)
    r0 = ret[0]
    plant = n_total - 7
    for r in range(ws):
        o = ret[r]
        assert o["shape"] == (32, n_total // ws, 8, 32, 32) and o["range"] == (r * n_total // ws, (r + 1) * n_total // ws)
        assert o["sim"].shape == (32, n_total) and torch.equal(o["sim"], r0["sim"]) and torch.equal(o["idx"], r0["idx"])
    assert torch.equal(r0["sim"], r0["sim1"]) and torch.equal(r0["idx"], r0["idx1"])                       # (i)
    print(f"{n_total} templates x 32 queries over 8 ranks ({cdt} compute, {bank_dtype} bank): oracle spot check on {r0['oracle_pairs']}: "
          f"score rel err {r0['oracle_err']:.3e}")
    assert r0["oracle_err"] < 2 * MODE_BOUNDS[cdt][1]          # (ii) per-score relative error: 2 x the mode's bound (observed 4.6e-3 bf16 / 6.6e-4 f16)
    assert bool((r0["idx"][:, 0] == plant).all()) and bool((r0["sim"][:, plant] == 0).all())               # (iii)
    assert bool(torch.signbit(r0["sim"][:, plant]).all())                                                  # ... exactly -0.0
    assert bool(torch.isfinite(r0["sim"]).all())


def test_geodesic_metric_vs_reference_run_on_device(gpu):
    """Row f2 on the poses' device against outputs of the reference's own loss.py (tests/golden/metric_ref.npz,
    make_golden_f2f3.py): all three symmetry branches, top-1 and top-k forms, CUDA float64 arithmetic."""
    from tests.test_host_logic import _check_metric_against_reference
    _check_metric_against_reference("cuda")


def test_crop_warp_on_device_with_reference_geometry(gpu):
    """Row f3, image side: the four source points the REFERENCE's crop_frame hands to cv2.getPerspectiveTransform (recorded in
    tests/golden/poses_ref.npz) define the map; the device warp of a coordinate-ramp image must return, at every output pixel
    whose source lies inside the frame, the source coordinates that map prescribes (bilinear interpolation of a linear
    function is exact).  OpenCV's fixed-point interpolation itself stays unpinned (cv2 is not installed here)."""
    import numpy as np
    from nope_amd import dataset as D
    w = np.load(os.path.join(ROOT, "tests", "golden", "poses_ref.npz"))
    H = W = 512
    ys, xs = torch.meshgrid(torch.arange(H, dtype=torch.float32), torch.arange(W, dtype=torch.float32), indexing="ij")
    ramp = torch.stack([xs, ys], -1).cuda()                                    # (H, W, 2): channel 0 = x, channel 1 = y
    for row, src, dst in zip(w["crop/in"], w["crop/src"], w["crop/dst"]):
        pose, inplane, vb, size = row[:16].reshape(4, 4), bool(row[16]), float(row[17]), int(row[18])
        M = D.get_perspective_transform(src, dst)
        out = D.crop_frame(ramp, None, w["perspective/K"], pose, size, keep_inplane=inplane, virtual_bbox_size=vb).cpu().numpy()
        Minv = np.linalg.inv(M)
        uu, vv = np.meshgrid(np.arange(size), np.arange(size))
        h = np.stack([uu, vv, np.ones_like(uu)], -1).astype(np.float64) @ Minv.T
        sx, sy = h[..., 0] / h[..., 2], h[..., 1] / h[..., 2]
        inside = (sx >= 0) & (sx <= W - 1) & (sy >= 0) & (sy <= H - 1)
        assert inside.mean() > 0.2
        assert np.abs(out[0][inside] - sx[inside]).max() < 2e-2 and np.abs(out[1][inside] - sy[inside]).max() < 2e-2


def test_unet_graph_replay_matches_direct(gpu):
    """hipGraph replay of small forwards is opt-in (nope_unet_graph_limit): on a NON-default stream a graph must really be
    instantiated and replayed, and its output must equal the direct launches' bit for bit -- in f32, bf16x3 and f16, into a different
    output tensor, and after the workspace was regrown (a new capture)."""
    from nope_amd.u_net import UNet
    from nope_amd.weights import synth_init_
    from tests.util import StubEncoder
    g = torch.Generator().manual_seed(31)
    x = torch.randn(1, 8, 16, 16, generator=g).cuda()
    poses = torch.randn(1, 9, 6, generator=g).cuda()
    for cdt in ("f32", "bf16x3", "f16"):
        u = UNet(u_net_dim=64, rot_representation_dim=6, encoder=StubEncoder(8), pose_mlp_name="single_layer", compute_dtype=cdt)
        synth_init_(u, 2022)
        u = u.cuda()
        direct5 = u.forward_hypotheses(x, poses[:, :5].contiguous())
        direct9 = u.forward_hypotheses(x, poses)
        h = u._get_handle(x.device)
        assert h.graph_replays() == 0                       # off by default
        h.graph_limit(1 << 20)
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            y1 = u.forward_hypotheses(x, poses[:, :5].contiguous())
            y2 = u.forward_hypotheses(x, poses[:, :5].contiguous(), out=torch.empty_like(y1))      # another output pointer, same graph
            y3 = u.forward_hypotheses(x, poses)                                                     # bigger batch: the workspace regrows
            y4 = u.forward_hypotheses(x, poses[:, :5].contiguous())                                 # ... so this shape is captured afresh
        side.synchronize()
        assert h.graph_replays() == 4, (cdt, h.graph_replays())
        assert torch.equal(y1, direct5) and torch.equal(y2, direct5) and torch.equal(y4, direct5) and torch.equal(y3, direct9), cdt
        h.graph_limit(0)
        assert torch.equal(u.forward_hypotheses(x, poses[:, :5].contiguous()), direct5) and h.graph_replays() == 4


def test_encoder_graph_replay_matches_direct(gpu):
    """The encoder's hipGraph replay (on by default: 54 launches of 5-15 us) against direct launches (NOPE_ENC_GRAPH=0), on the default and on a
    side stream (the stream `generate_and_retrieve` encodes the query on): equal bits."""
    from nope_amd.encoder import FeatureExtractor
    from nope_amd.weights import synth_init_
    e = FeatureExtractor(8, 0.2, False, compute_dtype="f16")
    synth_init_(e, 2022, prefix="encoder.")
    e = e.cuda()
    img = torch.rand(2, 3, 128, 128, generator=torch.Generator().manual_seed(4)).cuda() * 2 - 1
    side = torch.cuda.Stream()
    outs = {}
    for tag, env in (("graph", None), ("direct", "0")):
        if env is not None:
            os.environ["NOPE_ENC_GRAPH"] = env
        a = e.encode_image(img)
        b = e.encode_image(img)              # second call: the replay proper
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            c = e.encode_image(img)
        side.synchronize()
        torch.cuda.synchronize()
        os.environ.pop("NOPE_ENC_GRAPH", None)
        assert torch.equal(a, b) and torch.equal(a, c), tag
        outs[tag] = a
    assert torch.equal(outs["graph"], outs["direct"])


def test_pipelined_encoders_equal_bits(gpu):
    """PoseConditional.pipeline_encoders (bench.py's headline schedule): both encoder passes of a call on the side stream without waiting for
    what the current stream has queued, so consecutive calls overlap on the device.  Three different (query, reference) pairs issued back to
    back, twice: every call returns the bits of the unpipelined call on the same inputs (a call that picked up its neighbour's embeddings or
    a workspace still in use would not)."""
    from nope_amd.harness import build_model, synthetic_batch
    m = build_model(seed=2022, compute_dtype="bf16x3", device="cuda", u_net_dim=64)
    batches = [synthetic_batch(1, 40, 128, seed=s, device="cuda") for s in (11, 12, 13)]
    torch.cuda.synchronize()                       # the inputs are complete: the flag's precondition
    want = []
    m.pipeline_encoders = False
    for b in batches:
        sim, idx, bank = m.generate_and_retrieve(b["query"], b["reference"], b["all_relativeR"])
        want.append((sim.clone(), idx.clone(), bank.clone()))
    torch.cuda.synchronize()
    assert not torch.equal(want[0][0], want[1][0])
    m.pipeline_encoders = True
    got = []
    for rep in range(2):
        for b in batches:
            got.append(m.generate_and_retrieve(b["query"], b["reference"], b["all_relativeR"]))      # no synchronisation in between
    torch.cuda.synchronize()
    for i, (sim, idx, bank) in enumerate(got):
        w = want[i % 3]
        assert torch.equal(sim, w[0]) and torch.equal(idx, w[1]) and torch.equal(bank, w[2]), i


def test_two_stream_split_close_to_single_batch(gpu):
    """`two_stream_below` (off by default): the two half batches on two HIP streams agree with the single batch to rounding -- not bit
    for bit, the halves have other GEMM row counts and so other launch plans -- and the bank is complete when the call returns."""
    from nope_amd.model import PoseConditional
    from nope_amd.u_net import UNet
    from nope_amd.weights import synth_init_
    from tests.util import StubEncoder
    g = torch.Generator().manual_seed(32)
    feat = torch.randn(1, 8, 16, 16, generator=g).cuda()
    poses = torch.randn(1, 27, 6, generator=g).cuda()
    u = UNet(u_net_dim=64, rot_representation_dim=6, encoder=StubEncoder(8), pose_mlp_name="single_layer", compute_dtype="f32")
    synth_init_(u, 2022)
    one = PoseConditional(u, None, {"similarity_metric": "l2"}, None).cuda()
    two = PoseConditional(u, None, {"similarity_metric": "l2"}, None, two_stream_below=64)
    b1 = one.generate_templates_from_feat(feat, poses)
    b2 = two.generate_templates_from_feat(feat, poses)
    torch.cuda.synchronize()
    assert b2.shape == b1.shape and rel(b2, b1) < 1e-5


def test_geodesic_kernel_known_answers(gpu):
    """Row f2 on the device: `nope_op_geodesic` against hand-computed answers of pytorch3d's published formula, the ValueError, the
    gather form against the host restatement (tests/test_host_logic.py::_check_geodesic_kernel)."""
    from tests.test_host_logic import _check_geodesic_kernel
    _check_geodesic_kernel(gpu, "cuda")


def test_geodesic_metric_on_device(gpu):
    """GeodesicError with all three symmetry branches (loss.py:14-115) evaluated on CUDA tensors -- the
    step right behind the hot path, on the poses' device -- equals the host evaluation (float64 branch arithmetic)."""
    from nope_amd.harness import random_rotations
    from nope_amd.metrics import GeodesicError
    g = torch.Generator().manual_seed(5)
    B = 48
    pred = random_rotations(B * 5, g).view(B, 5, 3, 3)
    gt = random_rotations(B, g)
    sym = torch.arange(B).view(B, 1) % 3                       # 0: none, 1: 180 degrees about Y, 2: circular
    for p in (pred, pred[:, 0]):
        e_cpu, r_cpu = GeodesicError([15, 30])(p, gt, sym)
        e_gpu, r_gpu = GeodesicError([15, 30])(p.cuda(), gt.cuda(), sym.cuda())
        assert e_gpu.is_cuda and torch.allclose(e_gpu.cpu().double(), e_cpu.double(), atol=1e-4, equal_nan=True)
        assert set(r_cpu) == set(r_gpu)
        for k in r_cpu:
            assert abs(float(r_cpu[k]) - float(r_gpu[k])) < 1e-3, k


def test_ldm_variant_full_size(gpu):
    """The LDM cross-attention variant at the size configs/model/vae_cin_ldm.yaml ships (model_channels 256, channel_mult
    (1,2,4), two ResBlocks per level, SpatialTransformers at all three resolutions, context_dim 512; latent channels 8 instead of
    4 so the bf16 mode applies) at a 32x32 latent: f32 mode against the CPU restatement on 2 pose hypotheses, bf16 mode on a
    batch of 32 against the f32 mode."""
    import time
    from nope_amd.ldm import UNetModelPose
    from nope_amd.weights import synth_init_
    from tests.util import StubEncoder
    kw = dict(injecting_condition_twice=False, pose_mlp_name="single_layer", rot_representation_dim=6, image_size=32, in_channels=8,
              model_channels=256, out_channels=8, num_res_blocks=2, attention_resolutions=[4, 2, 1], channel_mult=(1, 2, 4),
              num_head_channels=32, use_spatial_transformer=True, transformer_depth=1, context_dim=512)
    m = UNetModelPose(encoder=StubEncoder(8), compute_dtype="f32", **kw)
    synth_init_(m, 2022)
    nparam = sum(v.numel() for k, v in m.own_state_dict().items())
    sd = {k: v.clone() for k, v in m.own_state_dict().items()}
    g = torch.Generator().manual_seed(3)
    x, poses = torch.randn(1, 8, 32, 32, generator=g), torch.randn(1, 32, 6, generator=g)
    m = m.cuda()
    y32 = m.forward_hypotheses(x.cuda(), poses.cuda())
    torch.cuda.synchronize()
    want = R.ldm_forward(sd, x.expand(2, -1, -1, -1), poses[0, :2])
    e = rel(y32[0, :2].cpu(), want)
    mb = UNetModelPose(encoder=StubEncoder(8), compute_dtype="bf16", **kw)
    mb.load_state_dict(m.state_dict())
    mb = mb.cuda()
    yb = mb.forward_hypotheses(x.cuda(), poses.cuda())
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    yb = mb.forward_hypotheses(x.cuda(), poses.cuda())
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    eb = rel(yb.float(), y32)
    print(f"LDM variant, {nparam / 1e6:.1f} M parameters, 32x32 latent: f32 vs oracle {e:.2e}; bf16 vs f32 {eb:.2e}; bf16 {32 / dt:.0f} hypotheses/s (batch of 32)")
    assert e < 1e-4 and eb < 8e-2
    del mb
    # the split-precision modes at 128 hypotheses (the launches that take the ping-pong kernels): f16x2 = the 3x3 convs on the two-pass
    # tile with range tracking (x2_range.h), everything else as bf16x3; both inside 1e-4 of the f32 mode; then the same input x 1e4
    # (un-normalised activations reach the first convolutions): range_mode "repeat" re-centres the shifts and stays accurate
    poses128 = torch.randn(1, 128, 6, generator=g)
    y32 = m.forward_hypotheses(x.cuda(), poses128.cuda())
    y32s = m.forward_hypotheses(x.cuda() * 1e4, poses128.cuda())
    for cdt in ("bf16x3", "f16x2"):
        mm = UNetModelPose(encoder=StubEncoder(8), compute_dtype=cdt, **kw)
        mm.load_state_dict(m.state_dict())
        mm = mm.cuda()
        os.environ["NOPE_X2_RANGE_CHECK"] = "2"        # (handles are created at the first forward: range_mode "repeat" for the f16x2 one)
        try:
            y = mm.forward_hypotheses(x.cuda(), poses128.cuda())
        finally:
            os.environ.pop("NOPE_X2_RANGE_CHECK")
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        y = mm.forward_hypotheses(x.cuda(), poses128.cuda())
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        ex = rel(y.float(), y32)
        h = mm._handle
        ys = mm.forward_hypotheses(x.cuda() * 1e4, poses128.cuda())
        exs = rel(ys.float(), y32s)
        print(f"LDM variant {cdt}, 128 hypotheses: {ex:.2e} vs f32, {dt * 1e3:.1f} ms per forward; input x 1e4: {exs:.2e}"
              + (f", repeated {[(ev['attempt'], ev['code'], ev['layers_out_of_range']) for ev in h.range_events]}" if cdt == "f16x2" else ""))
        assert ex < 1e-4 and exs < 1e-4, (cdt, ex, exs)
        del mm
