import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden():
    def _load(name):
        return np.load(os.path.join(GOLDEN, name), allow_pickle=False)
    return _load


@pytest.fixture(scope="session")
def cuda():
    import torch
    if not torch.cuda.is_available():
        pytest.fail("GPU test selected (-m gpu) but no GPU is visible")
    return torch.device("cuda:0")


@pytest.fixture(scope="session", autouse=True)
def _native_library():
    """The HIP library is built in-tree (hipcc cross-compiles without a GPU); build it if it is not there."""
    from cerebralsignalnetworks_amd import cabi
    if not os.path.exists(cabi.LIB_PATH):
        import __graft_entry__ as graft
        graft.build()
    yield
