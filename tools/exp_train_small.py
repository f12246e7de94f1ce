#!/usr/bin/env python3
"""GaussianProcessRegressor.fit with the optimiser and one restart at a small size (P = 6 outputs), restarts side by side or one
after the other.    python tools/exp_train_small.py N [concurrent: 1 | 0]   (GPK_OPTS=ptile_inv_max_np=0: level-by-level inverse factor)"""
import sys, os, time, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bench import synthetic_problem
from unmanned_aerial_vehicles_amd import GaussianProcessRegressor
from tools import gpk_opts  # noqa: E402
gpk_opts.install()      # GPK_OPTS=ptile_inv_max_np=0 ...: A/B switches
from unmanned_aerial_vehicles_amd.kernels import RBF, WhiteKernel
N = int(sys.argv[1]); conc = int(sys.argv[2]) if len(sys.argv) > 2 else 1
X, Y, _ = synthetic_problem(N, 1, D=10, P=6)
best = 1e9
for rep in range(3):
    g = GaussianProcessRegressor(kernel=RBF(0.5) + WhiteKernel(0.1), alpha=1e-4, normalize_y=True, n_restarts_optimizer=1, random_state=0)
    g.concurrent_restarts = bool(conc)
    torch.cuda.synchronize(); t0 = time.perf_counter(); g.fit(X, Y); torch.cuda.synchronize(); best = min(best, time.perf_counter() - t0)
print(f"N={N} concurrent={conc}: fit {best*1e3:.2f} ms, kernel {g.kernel_}, lml {g.log_marginal_likelihood_value_:.6f}")
