import os
import sys

import numpy as np
import pytest

REPO = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
if REPO not in sys.path:
    sys.path.insert(0, REPO)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # the stand-in backbones of the tests are initialised by the tests themselves: MAGNET()'s "args.*_ckpt is not set" notice is expected
    # here (tests/test_host_logic.py asserts it with pytest.warns, which overrides this filter inside its block)
    config.addinivalue_line("filterwarnings", r"ignore:MAGNET. args\.\w+_ckpt is not set:UserWarning")


@pytest.fixture(scope="session")
def golden():
    """Golden vectors captured from the imported reference (tests/golden/make_golden.py)."""
    return np.load(os.path.join(REPO, "tests", "golden", "golden_v1.npz"))


@pytest.fixture(scope="session")
def golden_r2():
    """Round-2 golden vectors from the imported reference (tests/golden/make_golden_r2.py): est_costvolume_CW at the
    C2 / C4 / C5 shapes, and one training step (loss + gradients)."""
    return np.load(os.path.join(REPO, "tests", "golden", "golden_r2.npz"))


@pytest.fixture(scope="session")
def golden_r4():
    """Round-4 golden vector from the imported reference (tests/golden/make_golden_r4.py): BASELINE config 5 with the reference's own
    F-Net in the loop (G15)."""
    return np.load(os.path.join(REPO, "tests", "golden", "golden_r4.npz"))


@pytest.fixture(scope="session")
def golden_r3():
    """Round-3 golden vectors from the imported reference (tests/golden/make_golden_r3.py): the reference's full forward at
    D = 64, I = 3 (G13) and est_costvolume_CW on the smooth synthetic variant at the C2 / C4 shapes (G14)."""
    return np.load(os.path.join(REPO, "tests", "golden", "golden_r3.npz"))


@pytest.fixture(scope="session")
def hip_lib():
    """Build (if stale) and load libmagnet_hip.so.  MAGNET_TEST_DEV_LIB=1 (development only) runs the suite against the -DMAGNET_DEV
    library instead, so that MAGNET_DEV_FLAGS can route every case through a kernel variant (tools/README.md)."""
    from magnet_amd import build, lib
    if os.environ.get("MAGNET_TEST_DEV_LIB") == "1":
        build.build(dev=True)
        lib.use_dev_build()
        return lib.load()
    build.build()
    return lib.load()


@pytest.fixture(scope="session")
def gpu():
    import torch
    if not torch.cuda.is_available():
        pytest.fail("test marked gpu but no GPU is visible — the HIP path must run, there is no CPU fallback")
    return torch.device("cuda:0")
