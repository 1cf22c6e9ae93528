"""Round 6: (i) the lean wide epilogue of the f32-storage modes (epilogue_wide, LEANM = 1; LEAN instantiations of the 128 x 192, streaming,
per-tap and tap-resident kernels) -- bit-identical to the generic row loop it replaces; (ii) the streaming 1x1 kernel
(kernels_gemm_stream.hip; off by default, see plan_conv) -- bit-identical to the 128 x 192 kernel.  Reference operators: the 1x1 convs of
LinearAttention / Attention / ResnetBlock.res_conv (model_utils.py:269,373-374,399-401) and every 3x3 conv of Block (:240).

CPU: the kernel sources run under tests/hipemu in its adversarial settings (LDS-DMA landing as late as the counted vmcnt waits allow, waves
scheduled as far apart as the barriers permit).  GPU: the same cases on the device."""
import os

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_stream_kernel_and_lean_epilogue_under_adversarial_interpreter(emu_jobs):
    """tests/conftest.py: EMU_JOBS.  stream_light: the streaming kernel's op-level cases; lean_3 / lean_4: the lean epilogue in bf16x3 and f16x2 against
    the generic row loop; stream_unet (NOPE_EMU_FULL=1 -- the streaming kernel is off by default, measured slower; the GPU test
    below keeps it): fused PreNorm, residual, GroupNorm statistics in a whole forward, bit-identical."""
    emu_jobs.collect(["stream_light", "lean_3", "lean_4", "stream_unet"])


def test_plan_keeps_ragged_launches_off_the_lean_path():
    """Host logic: `lean` needs EVERY wave tile whole along M and made of whole 32-column passes along N (the kernel skips the passes behind the last
    channel, conv_gemm_common.h) -- the launcher's condition, restated (kernels_gemm.hip: launch_conv)."""
    src = open(os.path.join(ROOT, "nope_amd", "csrc", "kernels_gemm.hip")).read()
    cond = src[src.index("p.lean = ("):src.index("? 1 : 0;", src.index("p.lean = ("))]
    for must in ("M % bm == 0", "a.Cout % 32 == 0", "!p.posmajor", "!phased", "p.splits == 1", "0xffffffffull", "% 64 == 0"):
        assert must in cond, must
    common = open(os.path.join(ROOT, "nope_amd", "csrc", "conv_gemm_common.h")).read()
    assert "if (nw + pass * 32 >= p.Cout) continue;" in common      # ... and the pass skip the relaxed column condition relies on


@pytest.mark.gpu
def test_stream_kernel_and_lean_epilogue_gpu(gpu):
    from tests import lean_emu_case, stream_emu_case
    assert lean_emu_case.run(gpu, "cuda") < 1.0
    assert stream_emu_case.run(gpu, "cuda", dts=(3, 1, 2)) < 1.0
    e, same = stream_emu_case.run_unet(gpu, "cuda", 64, "bf16x3", n_hyp=4, hw=16)
    assert e < 1e-4 and same, (e, same)


@pytest.mark.gpu
def test_lean_epilogue_whole_step_bit_identical(gpu):
    """A 64-hypothesis U-Net forward at the full width (every launch shape of the benchmark's step at 1/8 of its rows) in bf16x3 and f16x2:
    NOPE_EPILOGUE_LEAN=0 and =1 give the same bank, bit for bit."""
    import torch
    from nope_amd.u_net import UNet
    from nope_amd.weights import synth_init_
    from tests.util import StubEncoder
    for cdt in ("bf16x3", "f16x2"):
        u = UNet(u_net_dim=192, rot_representation_dim=6, encoder=StubEncoder(8), pose_mlp_name="single_layer", compute_dtype=cdt)
        synth_init_(u, 2022)
        u = u.to("cuda")
        g = torch.Generator().manual_seed(5)
        x, pose = torch.randn(1, 8, 32, 32, generator=g).cuda(), torch.randn(1, 64, 6, generator=g).cuda()
        outs = []
        for lean in ("0", "1"):
            os.environ["NOPE_EPILOGUE_LEAN"] = lean
            outs.append(u.forward_hypotheses(x, pose).clone())
        os.environ.pop("NOPE_EPILOGUE_LEAN")
        assert bool(torch.isfinite(outs[0]).all()) and torch.equal(outs[0], outs[1]), cdt
