#!/usr/bin/env python3
"""LML at a few ragged sizes (to bisect a factorisation problem between builds / environment switches)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
from unmanned_aerial_vehicles_amd import GaussianProcessRegressor, RBF, WhiteKernel  # noqa: E402

for N in [int(a) for a in (sys.argv[1:] or ["4829", "5598", "4961", "4224", "6000", "6500"])]:
    rng = np.random.default_rng(N)
    X = rng.standard_normal((N, 7)); Y = np.sin(X @ rng.standard_normal((7, 2))) + 0.1 * rng.standard_normal((N, 2))
    g = GaussianProcessRegressor(kernel=RBF(2.0) + WhiteKernel(0.03), alpha=1e-8, normalize_y=True, optimizer=None).fit(X, Y)
    m, s = g.predict(X[:50], return_std=True)
    print(N, "%.10f" % g.log_marginal_likelihood_value_, "%.6e %.6e" % (np.abs(m).sum(), s.sum()), flush=True)
    del g
