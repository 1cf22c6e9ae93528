"""Record the REFERENCE's own scores for BASELINE configs[1] (1 query x 512 templates, 256x256) and configs[2] (32 queries x 512
templates) -- every one of the 512 / 16 384 pose hypotheses, not a spot check.

Run in the build container only (needs /root/reference; ~1 h of CPU on 4 threads, resumable):

    python tests/golden/make_golden_cfg12.py [cfg1] [cfg2]

What runs is the imported reference (tests/golden/_ref_import.py), module by module:
  * `FeatureExtractor.encode_image` (src/model/encoder/template.py:47-53) on the query and reference images, ONCE per image -- the
    reference's `sample` re-encodes the reference image for every template (model.py:115), a pure function of the image, so the
    hoisted call returns the same tensor;
  * `UNet.forward` (u_net.py:160-198) for every template index over the whole batch, exactly the loop of
    `generate_templates` (model.py:218-230: `pred_feat_templates[:, idx_template] = u_net(reference_feat, all_relativeR[:, idx_template])`);
  * `PoseConditional.retrieval` (model.py:254-266) on the bank, through a PoseConditional whose encoder hands the already-encoded
    query through (the StubEncoder of `retrieval.npz`).
The fixture keeps the (B, 512) scores, the top-5 indices and SHA-256 digests of two weight tensors and of the batch -- inputs are
regenerated from the seed by `nope_amd.harness.synthetic_batch` (a pure function), weights by `nope_amd.weights.synth_tensor`.
"""
from __future__ import annotations

import os
import sys
import tempfile
import time
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)

import _ref_import as RI  # noqa: E402
from nope_amd.encoder import FeatureExtractor  # noqa: E402
from nope_amd.harness import synthetic_batch  # noqa: E402
from nope_amd.u_net import UNet  # noqa: E402
from nope_amd.weights import sha256_of, synth_init_, synth_tensor  # noqa: E402

SEED = 2022
CASES = {"cfg1": dict(batch=1, templates=512, size=256, seed=2022),     # bench.py / test_pipeline_config2_vs_oracle
         "cfg2": dict(batch=32, templates=512, size=256, seed=77)}      # tests/test_gpu_configs.py::test_config2_batch32_x_512


def _ns(**k):
    return types.SimpleNamespace(**k)


@torch.no_grad()
def build_reference():
    FE, U, PC = RI.ref_feature_extractor_cls(), RI.ref_unet_cls(), RI.ref_pose_conditional_cls()
    mine_enc = FeatureExtractor(descriptor_size=8, threshold=0.2, normalize=False)
    synth_init_(mine_enc, SEED, prefix="encoder.")
    ref_enc = FE(descriptor_size=8, threshold=0.2, normalize=False)
    ref_enc.load_state_dict(mine_enc.state_dict(), strict=True)
    ref_enc.eval()
    mine = UNet(u_net_dim=192, rot_representation_dim=6, encoder=mine_enc, pose_mlp_name="single_layer")
    sd = {k: synth_tensor(SEED, k, tuple(v.shape)) for k, v in mine.own_state_dict().items()}
    ref = U(u_net_dim=192, rot_representation_dim=6, encoder=RI.StubEncoder(8), pose_mlp_name="single_layer")
    missing = ref.load_state_dict(sd, strict=False)
    assert not missing.missing_keys and not missing.unexpected_keys, missing
    ref.eval()
    pc = PC(ref, _ns(lr=5e-5, weight_decay=5e-4, warm_up_steps=500, use_inv_deltaR=True, loss_type="l1"),
            _ns(similarity_metric="l2"), tempfile.mkdtemp())
    return ref_enc, ref, pc, sd


@torch.no_grad()
def run_case(tag, ref_enc, ref, pc, sd, part_dir):
    c = CASES[tag]
    b = synthetic_batch(c["batch"], c["templates"], c["size"], seed=c["seed"])
    B, N = c["batch"], c["templates"]
    t0 = time.time()
    ref_feat = torch.cat([ref_enc.encode_image(b["reference"][i:i + 8], mode="mode") for i in range(0, B, 8)])
    q_feat = torch.cat([ref_enc.encode_image(b["query"][i:i + 8], mode="mode") for i in range(0, B, 8)])
    print(f"{tag}: encoder {time.time() - t0:.1f}s", flush=True)
    part = os.path.join(part_dir, f"{tag}_bank.npy")
    done_path = os.path.join(part_dir, f"{tag}_done.txt")
    bank = np.lib.format.open_memmap(part, mode="r+" if os.path.exists(part) else "w+", dtype=np.float32, shape=(B, N, 8, 32, 32))
    done = int(open(done_path).read()) if os.path.exists(done_path) else 0
    for n in range(done, N):
        bank[:, n] = ref(ref_feat, b["all_relativeR"][:, n, :]).numpy()          # model.py:223-227 without the re-encode
        if (n + 1) % 8 == 0 or n + 1 == N:
            bank.flush()
            open(done_path, "w").write(str(n + 1))
            el = time.time() - t0
            print(f"{tag}: template {n + 1}/{N}  {el:.0f}s elapsed", flush=True)
    sim, idx = pc.retrieval(q_feat, torch.from_numpy(np.asarray(bank)))           # model.py:254-266 (StubEncoder: the query is already encoded)
    digest = lambda t: np.array(sha256_of(t))
    np.savez_compressed(os.path.join(HERE, f"{tag}_scores.npz"), sim=sim.numpy(), idx=idx.numpy(),
                        batch=np.array([c["batch"], c["templates"], c["size"], c["seed"]]),
                        sha_query=digest(b["query"]), sha_poses=digest(b["all_relativeR"]),
                        sha_mid=digest(sd["mid_block1.block1.proj.weight"]),
                        sha_enc_conv1=digest(ref_enc.state_dict()["backbone.conv1.weight"]),
                        bank_first=np.asarray(bank[:4, :2]).copy(), query_feat=q_feat.numpy()[:2], reference_feat=ref_feat.numpy()[:2])
    top2 = sim.topk(2, dim=1).values
    print(f"{tag}: done in {time.time() - t0:.0f}s; top-5 of query 0 {idx[0].tolist()}; smallest top-1 gap "
          f"{float((top2[:, 0] - top2[:, 1]).min() / sim.abs().max()):.2e} of the score scale", flush=True)


if __name__ == "__main__":
    torch.set_num_threads(int(os.environ.get("NOPE_GOLDEN_THREADS", "4")))
    torch.manual_seed(SEED)
    which = [a for a in sys.argv[1:] if a in CASES] or list(CASES)
    part_dir = os.environ.get("NOPE_GOLDEN_PARTS", "/tmp/nope_golden_parts")
    os.makedirs(part_dir, exist_ok=True)
    mods = build_reference()
    for tag in which:
        run_case(tag, *mods, part_dir)
