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


@pytest.fixture(scope="session")
def loader_ref():
    d = np.load(os.path.join(GOLDEN, "loader_ref.npz"))
    return {k: d[k] for k in d.files}


@pytest.fixture(scope="session")
def trainer_ref():
    d = np.load(os.path.join(GOLDEN, "trainer_ref.npz"))
    return {k: d[k] for k in d.files}


class _Duck:
    """Stand-in with scikit-learn's attribute names for boxes without scikit-learn (class name = kernel type)."""

    def __init__(self, **kw):
        self.__dict__.update(kw)


class RBF(_Duck):
    pass


class WhiteKernel(_Duck):
    pass


class ConstantKernel(_Duck):
    pass


class Sum(_Duck):
    pass


class Product(_Duck):
    pass


def reference_pickle_dict(tr):
    """The dict the reference's `GPTrainer.save_models` pickles (src/px4/gp_trainer.py:214-221), rebuilt from the
    numeric content frozen in tests/golden/trainer_ref.npz: scikit-learn regressors and StandardScalers when
    scikit-learn is importable (the real thing a reference-written file holds), attribute-compatible stand-ins
    otherwise."""
    names = [str(n) for n in tr["names"]]
    D = tr["X"].shape[1]
    try:
        from sklearn.gaussian_process import GaussianProcessRegressor as SkGPR
        from sklearn.gaussian_process.kernels import RBF as SR, ConstantKernel as SC, WhiteKernel as SW
        from sklearn.preprocessing import StandardScaler as SS
        have = True
    except ImportError:
        have = False
    d = {"gp_models": {}, "scalers_X": {}, "scalers_y": {}, "training_stats": {}, "model_name": "ref_model",
         "creation_time": 0.0}
    for n in names:
        th = tr[f"{n}_theta"]
        ls, noise = np.exp(th[:D]), float(np.exp(th[D]))
        if have:
            kern = SC(1.0, constant_value_bounds="fixed") * SR(ls, (0.1, 10.0)) + SW(noise, (1e-5, 1e1))
            g = SkGPR(kernel=kern, n_restarts_optimizer=3, alpha=1e-6, normalize_y=False)
            g.kernel_ = kern
            sx, sy = SS(), SS()
        else:
            kern = Sum(k1=Product(k1=ConstantKernel(constant_value=1.0, constant_value_bounds="fixed"),
                                  k2=RBF(length_scale=ls, length_scale_bounds=(0.1, 10.0))),
                       k2=WhiteKernel(noise_level=noise, noise_level_bounds=(1e-5, 1e1)))
            g = _Duck(kernel_=kern, alpha=1e-6, normalize_y=False)
            sx, sy = _Duck(), _Duck()
        g.X_train_, g.y_train_ = tr[f"{n}_X_train"].copy(), tr[f"{n}_y_train"].copy()
        g.alpha_, g.L_ = tr[f"{n}_alpha"].copy(), tr[f"{n}_L"].copy()
        g._y_train_mean, g._y_train_std = np.zeros(1), np.ones(1)
        g.log_marginal_likelihood_value_ = float(tr[f"{n}_lml"])
        g.n_features_in_ = D
        sx.mean_, sx.scale_ = tr[f"{n}_sx_mean"].copy(), tr[f"{n}_sx_scale"].copy()
        sy.mean_, sy.scale_ = tr[f"{n}_sy_mean"].copy(), tr[f"{n}_sy_scale"].copy()
        for sc in (sx, sy):
            sc.var_, sc.n_features_in_, sc.n_samples_seen_ = sc.scale_ ** 2, len(sc.mean_), len(g.X_train_)
        d["gp_models"][n], d["scalers_X"][n], d["scalers_y"][n] = g, sx, sy
        st = tr[f"{n}_stats"]
        d["training_stats"][n] = {"mse": float(st[0]), "rmse": float(st[1]), "r2": float(st[2]),
                                  "kernel": str(tr[f"{n}_kernel_str"]), "log_marginal_likelihood": float(st[3])}
    return d


def relerr(a, b):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    return float(np.max(np.abs(a - b)) / max(np.max(np.abs(b)), 1e-300))
