import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session", autouse=True)
def built_library():
    """Compile libquantization_amd.so (hipcc, gfx950) when it is missing or stale — the .so is
    git-ignored, so a fresh checkout has none.  hipcc cross-compiles without a GPU."""
    from quantization_amd import _lib

    return _lib.build()


@pytest.fixture(scope="session")
def qo():
    """The parity oracle (oracle/qoracle.c through ctypes)."""
    from oracle import qoracle

    qoracle.build()
    return qoracle


@pytest.fixture(scope="session")
def golden():
    import numpy as np

    return np.load(os.path.join(ROOT, "tests", "golden", "pair_kernels.npz"))
