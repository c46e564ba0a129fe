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


@pytest.fixture(params=["i8", "f16", "i8-general-requant", "i8-trio"])
def cnn_arith(request, monkeypatch):
    """Every arithmetic form of the depth classifier (k_cnn.hip) must deliver the oracle's integers: contexts created inside a
    test that uses this fixture run conv2 / conv3 on the i8 MFMAs (the default), on the 16-bit MFMAs, and on the i8 MFMAs with the
    general requant form instead of the short ones (FHEVC_CNN_REQUANT, read by fhevc_set_weights), and on the i8 MFMAs as ONE 768-thread workgroup per CU
    (three groups one barrier interval apart: FHEVC_CNN_TRIO, round 4; opt-in, measured slower)."""
    arith = request.param
    monkeypatch.setenv("FHEVC_CNN_ARITH", "f16" if arith == "f16" else "i8")
    if arith == "i8-general-requant":
        monkeypatch.setenv("FHEVC_CNN_REQUANT", "general")
        monkeypatch.setenv("FHEVC_HADAMARD_FORM", "mfma")   # and the fused source Hadamard's MFMA form at 8 bit (default: the VALU form)
    else:
        monkeypatch.delenv("FHEVC_CNN_REQUANT", raising=False)
        monkeypatch.delenv("FHEVC_HADAMARD_FORM", raising=False)
    if arith == "i8-trio":
        monkeypatch.setenv("FHEVC_CNN_TRIO", "1")
    else:
        monkeypatch.delenv("FHEVC_CNN_TRIO", raising=False)
    return arith
