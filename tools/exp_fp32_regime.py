#!/usr/bin/env python3
"""An ill-conditioned model (N = 9873, sn2 = 0.0015, sf2 = 2.35, ARD): fp64 predict vs the oracle, and the fp32 mean
kernels / variance against fp64 -- where fp32 serving stops being adequate."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
from oracle import gp_oracle as O  # noqa: E402
from unmanned_aerial_vehicles_amd import RBF, ConstantKernel, GaussianProcessRegressor, WhiteKernel  # noqa: E402

rng = np.random.default_rng(16)
N, M, D = 9873, 1024, 9
X = rng.standard_normal((N, D)); Y = np.sin(X @ rng.standard_normal((D, 1))) + 0.1 * rng.standard_normal((N, 1))
Xq = rng.standard_normal((M, D))
ls = np.exp(rng.uniform(np.log(0.6), np.log(3.0), D)) * np.sqrt(D) / 2
for noise in (0.00147, 0.02, 0.1):
    k = ConstantKernel(2.35) * RBF(ls) + WhiteKernel(noise)
    g = GaussianProcessRegressor(kernel=k, alpha=1e-8, normalize_y=True, optimizer=None).fit(X, Y)
    m64, s64 = g.predict(Xq, return_std=True)
    st = O.fit_fixed(X, Y, ls, 2.35, noise, 1e-8, True)
    om, os_ = O.predict(st, Xq, return_std=True)
    print("noise", noise, "fp64 vs oracle: mean %.1e std %.1e" % (np.max(np.abs(m64 - om.ravel())) / np.max(np.abs(om)), np.max(np.abs(s64 - os_.ravel()) / os_.ravel())), flush=True)
    dev = g._dev
    for kern in ("valu", "mfma"):
        m32 = dev.predict_mean_dev(Xq, g._y_train_mean, g._y_train_std, "float32", kern).double().cpu().numpy().ravel()
        print("   fp32 mean", kern, "%.1e" % (np.max(np.abs(m32 - m64)) / np.max(np.abs(m64))), " auto picks", dev.mean_kernel_choice())
    v32 = dev.predict_var_dev(Xq, 2.35 + noise, 0.0, "float32", "inverse").cpu().numpy() * g._y_train_std[0] ** 2
    print("   fp32 std (inverse) %.1e   min posterior var / prior var %.1e" % (np.max(np.abs(np.sqrt(v32) - s64) / s64), np.min(s64 ** 2) / ((2.35 + noise) * g._y_train_std[0] ** 2)), flush=True)
