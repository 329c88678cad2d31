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
    # (no torch.set_num_threads here: the reference's golden vectors were made with torch's default thread count, and the
    # rounding-noise-sized gradients of the oracle tests move with the number of threads torch's CPU kernels sum over)
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "slow: multi-ten-second CPU test")


@pytest.hookimpl(tryfirst=True)
def pytest_cmdline_main(config):
    """CPU-only hosts: run the suite on four pytest-xdist workers unless -n was given (two oracle-against-reference tests run for
    minutes on their own; the rest fits beside them).  On a GPU box nothing changes: one process, one device."""
    if os.path.exists("/dev/kfd") or os.environ.get("ICL_TEST_WORKERS") == "0" or "PYTEST_XDIST_WORKER" in os.environ:
        return None      # (an xdist worker runs this hook too: it must never start workers of its own)
    if getattr(config.option, "numprocesses", "absent") is None and not config.getoption("usepdb", False):
        config.option.numprocesses = int(os.environ.get("ICL_TEST_WORKERS", "4"))
        config.option.dist = "worksteal"
    return None


# the tests that run for a minute or more on CPU, longest first (every xdist worker starts with one of them)
LONG_TESTS = ("test_icl_trainer_with_gradient_reducer_world2", "test_fused_sgd_step_scope_equals_the_plain_loop_with_torch_sgd",
              "test_three_trainer_steps_match_reference", "test_first_three_of_ten_trainer_steps_match_reference", "test_2d_unet_icl_step_matches_reference", "test_full_model_step_matches_reference",
              "test_swinunetr_icl_step_matches_reference", "test_swin_stage_matches_oracle", "test_deferred_instance_norm_equals_the_materialised_path",
              "test_swinunet2d_icl_step_matches_reference")


def pytest_collection_modifyitems(config, items):
    workers = getattr(config.option, "numprocesses", None) or int(os.environ.get("PYTEST_XDIST_WORKER_COUNT", "0"))
    if workers and workers > 1:
        # worksteal hands every worker a contiguous block of the collection: put one long test at the head of each block
        rank = {n: i for i, n in enumerate(LONG_TESTS)}
        long = sorted((it for it in items if it.originalname in rank), key=lambda it: rank[it.originalname])
        rest = [it for it in items if it.originalname not in rank]
        block = max(1, (len(items) + workers - 1) // workers)
        out = []
        for w in range(workers):
            out += long[w::workers] + rest[w * (block - 1):(w + 1) * (block - 1)]
        seen = set(map(id, out))
        out += [it for it in rest if id(it) not in seen]
        items[:] = out
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


def pct_err(a, b, q=99.9):
    """q-th percentile of |a-b| / (|b| + 1e-3 max|b|): unlike rel_err (a global max norm) it constrains the small-magnitude
    elements too — an element 1000x below the largest must still be right to about its own size."""
    a = np.asarray(a, dtype=np.float64).ravel()
    b = np.asarray(b, dtype=np.float64).ravel()
    floor = 1e-3 * max(float(np.abs(b).max()), 1e-30)
    return float(np.percentile(np.abs(a - b) / (np.abs(b) + floor), q))


def rms_err(a, b):
    """rms(a - b) / rms(b): for samples of cancellation-heavy quantities, where the max-norm is the maximum over ~1000 noisy draws and
    moves by several per cent with any change of summation order while this moves by one (tests/diag/momentum_sample.py)."""
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    return float(np.sqrt(((a - b) ** 2).mean() / max(float((b ** 2).mean()), 1e-60)))


def rel_err(a, b):
    """max |a-b| / max |b| (the 'rel fp32' tolerance of BASELINE.json north_star)."""
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    denom = max(float(np.abs(b).max()), 1e-30)
    return float(np.abs(a - b).max()) / denom
