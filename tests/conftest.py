"""pytest configuration: registers the `gpu` marker and shared fixture loaders.

`-m "not gpu"` tests run on CPU only (oracle vs golden vectors, host logic, C-ABI symbol
checks, gloo multi-process sharding).  `-m gpu` tests are the parity tests proper: they
call the HIP kernels through the C ABI and compare with the oracle / golden fixtures.
Nothing here reads /root/reference (it does not exist on the GPU box).
"""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")
# Every device buffer the package allocates during the tests starts out as NaN: a kernel that reads memory nobody
# wrote fails the parity check instead of passing on a freshly zeroed allocation (device.Backend.empty).
os.environ.setdefault("GPK_DEBUG_FILL", "nan")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: test needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def csv_data():
    d = np.load(os.path.join(GOLDEN, "csv_170501.npz"))
    return {k: d[k] for k in d.files}


@pytest.fixture(scope="session")
def ka():
    d = np.load(os.path.join(GOLDEN, "known_answers.npz"))
    return {k: d[k] for k in d.files}


@pytest.fixture(scope="session")
def eval_table():
    d = np.load(os.path.join(GOLDEN, "eval_table.npz"))
    return {k: d[k] for k in d.files}


def relerr(a, b):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    return float(np.max(np.abs(a - b)) / max(np.max(np.abs(b)), 1e-300))
