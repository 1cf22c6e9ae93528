import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden():
    import numpy as np
    import torch

    def load(name):
        z = np.load(os.path.join(GOLDEN, name), allow_pickle=False)
        return {k: (torch.from_numpy(z[k]) if z[k].dtype.kind in "fiu" else z[k]) for k in z.files}
    return load


@pytest.fixture(scope="session")
def emu():
    """Inject the CPU interpreter build of the kernel sources (tests/hipemu) into nope_amd.hip.
    Test-only: the package itself can only ever load the gfx950 library."""
    sys.path.insert(0, os.path.join(ROOT, "tests", "hipemu"))
    import build_emu
    if not build_emu.available():
        pytest.skip("host clang++ for hipemu not available")
    from nope_amd import hip
    lib = hip.NopeLib(build_emu.build())
    hip._set_library_for_testing(lib)
    yield hip
    hip._set_library_for_testing(None)


# ---- the interpreter runs of the tiled conv kernels (tests/*_emu_case.py as subprocesses, each under one adversarial setting of tests/hipemu).
# They are independent and CPU-bound, and each test's slowest process used to leave the other cores idle: all of them are started TOGETHER the
# first time one is asked for, and every test collects its own.  NOPE_EMU_FULL=1 adds the whole-U-Net schedules that otherwise run on the GPU only.
_L1, _S2, _L3, _L2 = ({"HIPEMU_DMA": "late", "HIPEMU_SHUFFLE": "1"}, {"HIPEMU_SHUFFLE": "2"}, {"HIPEMU_DMA": "late", "HIPEMU_SHUFFLE": "3"},
                      {"HIPEMU_DMA": "late", "HIPEMU_SHUFFLE": "2"})
EMU_JOBS = {   # name: (script, interpreter setting, arguments, marker of success, full-suite only)
    # test_conv_pingpong.py (compute modes: 1 = bf16, 3 = bf16x3 split precision on f32 data, 0 = f32, 2 = f16)
    "pp_light_13": ("pp_emu_case.py", _L1, ["--light", "--dts", "1,3"], "pp_emu_case OK", False),
    "pp_light_02": ("pp_emu_case.py", _S2, ["--light", "--dts", "0,2"], "pp_emu_case OK", False),
    "pp_unet16": ("pp_emu_case.py", {"HIPEMU_SHUFFLE": "1"}, ["--unet16"], "pp_emu_case OK", True),
    "x2_light": ("x2_emu_case.py", _L1, ["--light"], "x2 ok", False),
    "x2_unet": ("x2_emu_case.py", _S2, ["--unet"], "x2 unet ok", False),
    "up2p_light": ("up2p_emu_case.py", _L1, ["--light"], "up2p ok", False),
    # test_conv_small.py (tiles: 0 = 64 x 64 / 3 stages, 1 = 128 x 128, 2 = 64 x 64 / 6 stages, 3 = 64 x 64 by two K groups)
    "small_10": ("small_emu_case.py", _L1, ["--dts", "1,0", "--tiles", "0,1,3"], "small_emu_case OK", False),
    "small_32": ("small_emu_case.py", _S2, ["--dts", "3,2", "--tiles", "2,3", "--light"], "small_emu_case OK", False),
    "small_unet16": ("small_emu_case.py", _L3, ["--unet16"], "small_emu_case OK", False),
    "small_unet16split": ("small_emu_case.py", _L1, ["--unet16split"], "small_emu_case OK", True),
    "small_x2": ("small_emu_case.py", _L2, ["--x2"], "small_emu_case OK", False),
    # test_conv_stream.py
    "stream_light": ("stream_emu_case.py", _L1, ["--dts", "3", "--light"], "stream_emu_case OK", False),
    "stream_unet": ("stream_emu_case.py", _L3, ["--unet"], "stream_emu_case OK", True),
    "lean_3": ("lean_emu_case.py", _L1, ["--dts", "3"], "lean_emu_case OK", False),
    "lean_4": ("lean_emu_case.py", _S2, ["--dts", "4", "--light"], "lean_emu_case OK", False),
}


class _EmuJobs:
    def __init__(self):
        import subprocess
        self.full = os.environ.get("NOPE_EMU_FULL") == "1"
        self.procs = {}
        for name, (script, setting, args, _, full_only) in EMU_JOBS.items():
            if full_only and not self.full:
                continue
            env = dict(os.environ, HIPEMU_THREADS="8" if name == "x2_unet" else "2", **setting)      # (x2_unet: the longest run by far: it gets the cores the others leave)
            self.procs[name] = subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", script)] + args, env=env,
                                                stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)

    def collect(self, names, timeout=3000):
        """Wait for the named runs (those of the full suite are skipped outside it) and assert that each one succeeded."""
        for name in names:
            pr = self.procs.get(name)
            if pr is None:
                assert EMU_JOBS[name][4] and not self.full, name
                continue
            out, _ = pr.communicate(timeout=timeout)
            assert pr.returncode == 0 and EMU_JOBS[name][3] in out, (name, EMU_JOBS[name][:3], out[-2000:])


_WANT_EMU_JOBS = False


def pytest_collection_modifyitems(config, items):
    """The tests that collect interpreter runs go LAST: the runs are started when the session starts (below) and work through their CPU-minutes
    next to the rest of the suite instead of in front of it."""
    tail = [it for it in items if "emu_jobs" in getattr(it, "fixturenames", ())]
    if tail:
        items[:] = [it for it in items if it not in tail] + tail


def pytest_collection_finish(session):
    global _WANT_EMU_JOBS
    _WANT_EMU_JOBS = any("emu_jobs" in getattr(it, "fixturenames", ()) for it in session.items)      # (after -m / -k deselection)


@pytest.fixture(scope="session", autouse=True)
def _emu_jobs_from_the_start(request):
    if _WANT_EMU_JOBS:
        request.getfixturevalue("emu_jobs")
    yield


@pytest.fixture(scope="session")
def emu_jobs(emu):
    jobs = _EmuJobs()        # (`emu` first: the interpreter library is built once, before the processes that load it start)
    yield jobs
    for pr in jobs.procs.values():
        if pr.poll() is None:
            pr.kill()


@pytest.fixture(scope="session")
def gpu():
    """The product path: gfx950 library on cuda:0.  Fails (never skips, never falls back) when
    the library is missing on a GPU box."""
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from nope_amd import hip
    hip._set_library_for_testing(None)
    hip.lib()          # raises NopeError if libnope_hip.so was not built
    return hip
