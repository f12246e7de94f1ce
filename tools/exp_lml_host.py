#!/usr/bin/env python3
"""Wall time of one LML + gradient evaluation through the estimator at the reference's own size (N = 1000, D = 10, P = 6) and
where the host spends it (cProfile).   python tools/exp_lml_host.py [N]"""
import cProfile
import os
import pstats
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402
from unmanned_aerial_vehicles_amd import GaussianProcessRegressor  # noqa: E402
from tools import gpk_opts  # noqa: E402
gpk_opts.install()      # GPK_OPTS=ptile_inv_max_np=0 ...: A/B switches
from unmanned_aerial_vehicles_amd.kernels import RBF, WhiteKernel  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
rng = np.random.default_rng(0)
X = rng.standard_normal((N, 10))
Y = np.sin(X @ rng.standard_normal((10, 6))) + 0.1 * rng.standard_normal((N, 6))
g = GaussianProcessRegressor(kernel=RBF(0.5) + WhiteKernel(0.1), alpha=1e-4, normalize_y=True, optimizer=None).fit(X, Y)
th = g.kernel_.theta
f = lambda: g._lml_on_device(th, True)
for _ in range(20):
    f()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(200):
    f()
torch.cuda.synchronize()
print(f"N={N}: {1e3 * (time.perf_counter() - t0) / 200:.3f} ms per LML + gradient evaluation (wall)")
pr = cProfile.Profile()
pr.enable()
for _ in range(200):
    f()
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(18)
