#!/usr/bin/env python3
"""Where the time of a small predict() call goes on the host side: full estimator call, DeviceGP.predict_host, the
bare C call with prebuilt arguments, and the helpers around it (medians over 2000 calls)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
from unmanned_aerial_vehicles_amd import GaussianProcessRegressor, RBF, WhiteKernel, _lib  # noqa: E402
from unmanned_aerial_vehicles_amd.device import _p  # noqa: E402

rng = np.random.default_rng(0)
N, D, P = 1000, 10, 6
X = rng.standard_normal((N, D)); Y = np.sin(X @ rng.standard_normal((D, P))) * 0.05
gp = GaussianProcessRegressor(kernel=RBF(0.5) + WhiteKernel(0.1), alpha=1e-4, normalize_y=True, optimizer=None).fit(X, Y)
dev = gp._dev
be = dev.be
q1 = X[:1] + 0.01


def med(f, n=2000):
    for _ in range(50):
        f()
    ts = []
    for _ in range(n):
        t0 = time.perf_counter(); f(); ts.append(time.perf_counter() - t0)
    ts.sort()
    return ts[n // 2] * 1e6


for M in (1, 25):
    q = X[:M] + 0.01
    print(f"M={M}")
    print("  gp.predict(return_std)        %.1f us" % med(lambda: gp.predict(q, return_std=True)))
    print("  gp.predict(mean)              %.1f us" % med(lambda: gp.predict(q)))
    print("  dev.predict_host(var)         %.1f us" % med(lambda: dev.predict_host(q, gp._y_train_mean, gp._y_train_std, 1.1, 0.0)))
    print("  dev.predict_host(mean)        %.1f us" % med(lambda: dev.predict_host(q, gp._y_train_mean, gp._y_train_std, None, 0.0)))
    dp = _lib._dp
    W = dev.inverse_factor(False)
    mean = np.empty((M, P)); var = np.empty((M,))
    ym = np.ascontiguousarray(gp._y_train_mean); ys = np.ascontiguousarray(gp._y_train_std)
    args = (be.h, _p(dev.X), _p(dev.alpha), dev.N, dev.D, dev.P, dev.ls.ctypes.data_as(dp), dev.sf2, ym.ctypes.data_as(dp),
            ys.ctypes.data_as(dp), _p(W), dev.Np, dev.Np, 1.1, 0.0, q.ctypes.data_as(dp), M, mean.ctypes.data_as(dp), var.ctypes.data_as(dp))
    fn = be.lib.gpk_predict_host
    print("  bare C call (var)             %.1f us" % med(lambda: fn(*args)))
    args_m = args[:10] + (None,) + args[11:18] + (None,)
    print("  bare C call (mean)            %.1f us" % med(lambda: fn(*args_m)))
print("be.bind_stream()                %.1f us" % med(lambda: be.bind_stream()))
print("kernel_.components()            %.1f us" % med(lambda: gp.kernel_.components()))
print("_p(tensor)                      %.1f us" % med(lambda: _p(dev.X)))
print("ndarray.ctypes.data_as          %.1f us" % med(lambda: q1.ctypes.data_as(_lib._dp)))

# ---- six per-axis ARD GPs (gp_trainer.py / pretrained_gp.py:52-98): one fused call against the per-model loop
from unmanned_aerial_vehicles_amd import BatchedARDGP  # noqa: E402
Nt = 800
Y6 = np.sin(X[:Nt] @ rng.standard_normal((D, 6))) + 0.05 * rng.standard_normal((Nt, 6))
bg = BatchedARDGP(length_scale=np.full(D, 2.0), noise_level=0.05, alpha=1e-6, normalize_y=False, optimizer=None).fit(X[:Nt], Y6)
for M in (1, 25):
    q = X[:M] + 0.01
    print(f"6 per-axis GPs, M={M}")
    print("  fused call (mean + std)       %.1f us" % med(lambda: bg.predict(q, return_std=True)))
    print("  fused call (mean)             %.1f us" % med(lambda: bg.predict(q)))
    print("  per-model loop (mean + std)   %.1f us" % med(lambda: [m.predict(q, return_std=True) for m in bg.models], 500))
try:
    from sklearn.gaussian_process import GaussianProcessRegressor as SkGPR
    from sklearn.gaussian_process.kernels import RBF as SkRBF, ConstantKernel as SkC, WhiteKernel as SkW
    sks = [SkGPR(kernel=SkC(1.0, "fixed") * SkRBF(np.full(D, 2.0)) + SkW(0.05), alpha=1e-6, optimizer=None).fit(X[:Nt], Y6[:, i])
           for i in range(6)]
    for M in (1, 25):
        q = X[:M] + 0.01
        print(f"  scikit-learn loop over 6 GPs, M={M} (mean + std)  %.1f us" % med(lambda: [g.predict(q, return_std=True) for g in sks], 200))
except ImportError:
    pass
