#!/usr/bin/env python3
"""A handful of 25-row predict(return_std=True) calls at the reference's sizes, for a rocprofv3 kernel trace."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
from unmanned_aerial_vehicles_amd import GaussianProcessRegressor, RBF, WhiteKernel  # noqa: E402

rng = np.random.default_rng(0)
N, D, P = 1000, 10, 6
X = rng.standard_normal((N, D)); Y = np.sin(X @ rng.standard_normal((D, P))) * 0.05
gp = GaussianProcessRegressor(kernel=RBF(0.5) + WhiteKernel(0.1), alpha=1e-4, normalize_y=True, optimizer=None).fit(X, Y)
for _ in range(20):
    gp.predict(X[:25] + 0.01, return_std=True)
print("done")
