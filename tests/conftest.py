import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "slow: multi-ten-second CPU test")


def pytest_collection_modifyitems(config, items):
    # GPU tests are skipped (not failed) when selected on a host without a GPU.
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="no GPU visible")
    for it in items:
        if "gpu" in it.keywords:
            it.add_marker(skip)


@pytest.fixture(scope="session")
def golden_unit():
    return np.load(os.path.join(GOLDEN, "unit.npz"), allow_pickle=False)


def load_golden(name):
    return np.load(os.path.join(GOLDEN, name), allow_pickle=False)


def rel_err(a, b):
    """max |a-b| / max |b| (the 'rel fp32' tolerance of BASELINE.json north_star)."""
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    denom = max(float(np.abs(b).max()), 1e-30)
    return float(np.abs(a - b).max()) / denom
