import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    from oracle import oracle_py
    return oracle_py.load_oracle()


@pytest.fixture(scope="session")
def golden():
    import numpy as np
    return np.load(os.path.join(ROOT, "tests", "golden", "ref_vectors.npz"))


@pytest.fixture(scope="session", autouse=True)
def _torch_hip_runtime_first(request):
    """torch ships its own copy of the HIP runtime; when GPU tests are selected bring it up before the in-tree
    library initialises the system one, so that the order does not depend on which test file runs first."""
    if any(item.get_closest_marker("gpu") for item in request.session.items):
        import torch
        if torch.cuda.is_available():
            torch.cuda.init()
    yield
