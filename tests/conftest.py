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
