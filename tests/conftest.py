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
        return np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False)
    return _load


def rel_err(a, b):
    """max |a-b| / max|b| -- the 'relative fp32 tolerance' of BASELINE.json:north_star."""
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-30))


def elem_rel_err(a, b, floor=1e-2):
    """ELEMENT-WISE relative error: max over the elements with |b| > floor * max|b| of |a-b| / |b| -- the stricter companion of
    ``rel_err`` (which divides every difference by the tensor's maximum).  Elements below the floor are values that cancel to
    near zero, where a relative figure measures the reference's own round-off, not the path under test."""
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    keep = np.abs(b) > floor * max(np.abs(b).max(), 1e-30)
    if not keep.any():
        return 0.0
    return float((np.abs(a - b)[keep] / np.abs(b)[keep]).max())


def l2_rel(a, b):
    a = np.asarray(a, dtype=np.float64).ravel()
    b = np.asarray(b, dtype=np.float64).ravel()
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30))
