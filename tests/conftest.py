import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `-m gpu` on the GPU box)")


def load_golden(name: str):
    return np.load(os.path.join(GOLDEN, name), allow_pickle=False)


def golden_state_dict(npz):
    import torch
    return {k[4:]: torch.from_numpy(npz[k].copy()) for k in npz.files if k.startswith("sd::")}


@pytest.fixture(scope="session")
def gpu():
    """The device for `-m gpu` tests. No GPU -> skip; GPU but no HIP library -> the tests must FAIL loudly."""
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU in this environment")
    from sparsefactorization_amd import _lib
    _lib.load()  # raises PSFLibraryError if libpsf_chord.so is missing: that is a failure, not a skip
    return torch.device("cuda:0")


def rel_inf(a, b):
    """max|a-b| / max|b| — the parity criterion of BASELINE.json (<= 1e-5 for fp32)."""
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    denom = np.max(np.abs(b))
    return float(np.max(np.abs(a - b)) / (denom if denom > 0 else 1.0))
