"""CPU: pin the oracle (oracle/nope_ref.py) against outputs recorded from the reference itself
(tests/golden/*.npz, written by tests/golden/make_golden.py).  fp32 on the same CPU kernels,
so agreement is expected to be (near) bit-exact; tolerances are written out."""
import pytest
import torch

from oracle import nope_ref as R

TOL = 2e-6   # relative to max |reference|


def rel(a, b):
    return float((a.double() - b.double()).abs().max() / (b.double().abs().max() + 1e-30))


def sub(d, tag):
    return {k[len(tag) + 3:]: v for k, v in d.items() if k.startswith(tag + "/w/")}


def test_blocks(golden):
    g = golden("blocks.npz")
    x = g["block/in0"]
    assert rel(R.block(x, sub(g, "block"), "", 8), g["block/out"]) < TOL
    for tag in ("resnet_proj", "resnet_id"):
        y = R.resnet_block(g[f"{tag}/in0"], g[f"{tag}/in1"], sub(g, tag), "", 8)
        assert rel(y, g[f"{tag}/out"]) < TOL
    assert rel(R.residual_prenorm(g["linattn/in0"], sub(g, "linattn"), "", R.linear_attention), g["linattn/out"]) < TOL
    assert rel(R.residual_prenorm(g["attn/in0"], sub(g, "attn"), "", R.attention), g["attn/out"]) < TOL
    assert rel(R.hard_downsample(g["down/in0"], sub(g, "down"), ""), g["down/out"]) < TOL
    assert rel(R.hard_upsample(g["up/in0"], sub(g, "up"), ""), g["up/out"]) < TOL


@pytest.mark.parametrize("tag,dim,mlp", [("d8", 8, "single_layer"), ("d16", 16, "single_layer"), ("d16two", 16, "two_layers"),
                                         ("d24pos", 24, "posEncoding"), ("d16soft", 16, "single_layer")])
def test_tiny_unets(golden, tag, dim, mlp):
    from nope_amd.u_net import UNet
    from nope_amd.weights import sha256_of, synth_init_
    from tests.util import StubEncoder
    g = golden("unet_tiny.npz")
    m = UNet(u_net_dim=dim, rot_representation_dim=6, encoder=StubEncoder(8), pose_mlp_name=mlp, use_hard_up_down=tag != "d16soft")
    synth_init_(m, 2022)
    sd = m.own_state_dict()
    assert sha256_of(sd["init_conv.weight"]) == str(g[f"{tag}/sha_init_conv"])      # generator drift check
    assert rel(R.unet_forward(sd, g[f"{tag}/x"], g[f"{tag}/pose"]), g[f"{tag}/out"]) < TOL


def test_retrieval(golden):
    g = golden("retrieval.npz")
    for tag in "abc":
        sim, idx = R.retrieval(g[f"{tag}/q"], g[f"{tag}/bank"])
        assert rel(sim, g[f"{tag}/sim"]) < TOL
        assert torch.equal(idx, g[f"{tag}/idx"])                  # fixtures are tie-free
        B, N = sim.shape
        assert sim[B - 1, N // 2] == 0 and idx[B - 1, 0] == N // 2   # planted exact match (KAT)


def test_encoder(golden, monkeypatch):
    from nope_amd import hip
    from nope_amd.encoder import FeatureExtractor
    from nope_amd.weights import sha256_of, synth_init_
    g = golden("encoder.npz")
    enc = FeatureExtractor(descriptor_size=8)
    synth_init_(enc, 2022, prefix="encoder.")
    sd = enc.state_dict()
    assert sha256_of(sd["backbone.conv1.weight"]) == str(g["sha_conv1"])
    assert rel(R.encode_image(sd, g["img"]), g["feat"]) < TOL
    import types
    stand_in = types.SimpleNamespace(dll=types.SimpleNamespace(nope_tuning_reload=lambda: None))
    monkeypatch.setattr(hip, "_lib", stand_in)                     # "a loaded product library" (no interpreter injected; passes in isolation:
    monkeypatch.setattr(hip, "_tuning_seen", None)                 #  hip.lib() calls nope_tuning_reload on whatever is loaded)
    with pytest.raises(hip.NopeError, match="no CPU path"):        # the product module never computes on host tensors
        enc.encode_image(g["img"])


def test_configs1_and_2_against_reference_recorded_scores(golden):
    """BASELINE configs[1] / [2] at full size (256x256, 512 templates): the reference's own recorded run (make_golden_cfg12.py: ALL
    512 / 16384 scores) pins the restatement on a bounded sample -- the encoder on the first recorded images, the U-Net on the first two
    hypotheses of the first recorded queries, their scores -- the whole banks are the GPU tests' job."""
    from nope_amd.harness import build_model, synthetic_batch
    from nope_amd.weights import sha256_of
    model = build_model(device="cpu")
    sd = model.u_net.own_state_dict()
    enc_sd = model.u_net.encoder.state_dict()
    for name, nq in (("cfg1_scores.npz", 1), ("cfg2_scores.npz", 2)):
        g = golden(name)
        B, N, size, seed = (int(v) for v in g["batch"])
        b = synthetic_batch(B, N, size, seed=seed)
        assert sha256_of(b["query"]) == str(g["sha_query"]) and sha256_of(b["all_relativeR"]) == str(g["sha_poses"])
        assert sha256_of(sd["mid_block1.block1.proj.weight"]) == str(g["sha_mid"])
        ref_feat = R.encode_image(enc_sd, b["reference"][:nq])
        q_feat = R.encode_image(enc_sd, b["query"][:nq])
        assert rel(ref_feat, g["reference_feat"][:nq]) < TOL and rel(q_feat, g["query_feat"][:nq]) < TOL
        bank = R.generate_templates(sd, ref_feat, b["all_relativeR"][:nq, :2])
        assert rel(bank, g["bank_first"][:nq]) < TOL
        s = R.similarity_scores(q_feat, bank)
        assert float(((s - g["sim"][:nq, :2]).abs() / g["sim"][:nq, :2].abs()).max()) < 5e-6
        assert g["sim"].shape == (B, N) and bool((g["sim"].topk(5, dim=1).indices == g["idx"]).all())


def test_pipeline_config1(golden):
    """BASELINE config 1 (single query, 64 templates, 128x128, full-size model): check a
    4-template prefix of the bank, the scores computed from the recorded bank prefix, and the loss
    definition -- the full 64-template U-Net pass is exercised on the GPU."""
    from nope_amd.harness import build_model
    g = golden("pipeline_cfg1.npz")
    model = build_model(device="cpu")
    sd = model.u_net.own_state_dict()
    from nope_amd.weights import sha256_of
    assert sha256_of(sd["mid_block1.block1.proj.weight"]) == str(g["sha_mid"])
    enc_sd = model.u_net.encoder.state_dict()
    qf = R.encode_image(enc_sd, g["query"])
    rf = R.encode_image(enc_sd, g["reference"])
    assert rel(qf, g["query_feat"]) < TOL and rel(rf, g["reference_feat"]) < TOL
    bank = R.generate_templates(sd, rf, g["all_relativeR"][:, :4])
    assert rel(bank, g["bank_head"]) < 1e-5
    s = R.similarity_scores(qf, bank)
    assert rel(s, g["sim"][:, :4]) < 1e-5
    assert torch.equal(R.topk_desc_lowest_index(g["sim"], 5), g["idx"])


LDM_CASES = {"m32": dict(model_channels=32, channel_mult=(1, 2), num_res_blocks=1, attention_resolutions=[1, 2], context_dim=24,
                         pose_mlp_name="single_layer", injecting_condition_twice=False),
             "m64two": dict(model_channels=64, channel_mult=(1, 2, 2), num_res_blocks=2, attention_resolutions=[2, 4], context_dim=40,
                            pose_mlp_name="two_layers", injecting_condition_twice=True),
             "m32film": dict(model_channels=32, channel_mult=(1, 2), num_res_blocks=1, attention_resolutions=[1, 2], context_dim=24,
                             pose_mlp_name="single_layer", injecting_condition_twice=False, use_scale_shift_norm=True),
             "m64film": dict(model_channels=64, channel_mult=(1, 2), num_res_blocks=1, attention_resolutions=[2], context_dim=40,
                             pose_mlp_name="single_layer", injecting_condition_twice=True, use_scale_shift_norm=True),
             "m32d2": dict(model_channels=32, channel_mult=(1, 2), num_res_blocks=1, attention_resolutions=[1, 2], context_dim=24,
                           pose_mlp_name="single_layer", injecting_condition_twice=False, transformer_depth=2)}


def build_ldm(tag, compute_dtype="f32"):
    from nope_amd.ldm import UNetModelPose
    from nope_amd.weights import synth_init_
    from tests.util import StubEncoder
    kw = dict(transformer_depth=1)
    kw.update(LDM_CASES[tag])
    m = UNetModelPose(encoder=StubEncoder(8), rot_representation_dim=6, image_size=8, in_channels=8, out_channels=8, num_head_channels=32,
                      use_spatial_transformer=True, compute_dtype=compute_dtype, **kw)
    synth_init_(m, 2022)
    return m


@pytest.mark.parametrize("tag", ["m32", "m64two", "m32film", "m64film", "m32d2"])
def test_ldm_variant(golden, tag):
    """The LDM cross-attention variant (UNetModelPose, adapt_openaimodel.py:130-158): the oracle's restatement against outputs
    recorded from the reference class (tests/golden/make_golden.py ldm), same synthesised weights."""
    from nope_amd.weights import sha256_of
    g = golden("ldm_tiny.npz")
    m = build_ldm(tag)
    sd = m.own_state_dict()
    assert sha256_of(sd["input_blocks.0.0.weight"]) == str(g[f"{tag}/sha_in"])
    assert rel(R.ldm_forward(sd, g[f"{tag}/x"], g[f"{tag}/pose"]), g[f"{tag}/out"]) < TOL
