#!/usr/bin/env python3
"""predict(mean + std) latency for 32..128 rows at the reference's model size (N = 1000, D = 10, P = 6)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
from unmanned_aerial_vehicles_amd import GaussianProcessRegressor, RBF, WhiteKernel  # noqa: E402

rng = np.random.default_rng(0)
X = rng.standard_normal((1000, 10)); Y = np.sin(X @ rng.standard_normal((10, 6))) * 0.05
gp = GaussianProcessRegressor(kernel=RBF(0.5) + WhiteKernel(0.1), alpha=1e-4, normalize_y=True, optimizer=None).fit(X, Y)
for M in (32, 50, 64, 65, 128):
    q = X[:M] + 0.01
    for _ in range(50):
        gp.predict(q, return_std=True)
    ts = []
    for _ in range(1000):
        t0 = time.perf_counter(); gp.predict(q, return_std=True); ts.append(time.perf_counter() - t0)
    print(M, "rows mean+std: %.1f us" % (sorted(ts)[500] * 1e6), flush=True)
