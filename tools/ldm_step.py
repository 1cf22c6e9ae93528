"""Tuning aid: the LDM cross-attention variant at its shipped size, bf16, a batch of pose hypotheses at a 32x32 latent
(run under rocprofv3 --kernel-trace --stats for the per-kernel split; prints hypotheses/s)."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

from nope_amd.ldm import UNetModelPose
from nope_amd.weights import synth_init_
from tests.util import StubEncoder


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 128
    dtype = sys.argv[sys.argv.index("--dtype") + 1] if "--dtype" in sys.argv else "bf16"
    kw = dict(injecting_condition_twice=False, pose_mlp_name="single_layer", rot_representation_dim=6, image_size=32, in_channels=8,
              model_channels=256, out_channels=8, num_res_blocks=2, attention_resolutions=[4, 2, 1], channel_mult=(1, 2, 4),
              num_head_channels=32, use_spatial_transformer=True, transformer_depth=1, context_dim=512)
    m = UNetModelPose(encoder=StubEncoder(8), compute_dtype=dtype, **kw)
    synth_init_(m, 2022)
    m = m.cuda()
    g = torch.Generator().manual_seed(3)
    x, poses = torch.randn(1, 8, 32, 32, generator=g).cuda(), torch.randn(1, n, 6, generator=g).cuda()
    m.forward_hypotheses(x, poses)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    reps = 3
    for _ in range(reps):
        m.forward_hypotheses(x, poses)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    print(f"LDM {dtype}, {n} hypotheses at 32x32: {dt * 1e3:.1f} ms per forward = {n / dt:.0f} hypotheses/s")


if __name__ == "__main__":
    main()
