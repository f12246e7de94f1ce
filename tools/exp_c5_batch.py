#!/usr/bin/env python3
"""BASELINE configs[4] at one size: LML + gradient of three per-axis ARD GPs as three chains on three streams against ONE chain in
the handle's batched mode (gpk_lml_batched: one synchronisation).    python tools/exp_c5_batch.py [N]   (GPK_OPTS=ptile_inv_max_np=0: level-by-level inverse factor)"""
import sys, os, time, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bench import synthetic_problem
from tools.run_configs import wall
from unmanned_aerial_vehicles_amd import BatchedARDGP
from tools import gpk_opts  # noqa: E402
gpk_opts.install()      # GPK_OPTS=ptile_inv_max_np=0 ...: A/B switches
N5 = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
X, Y, _ = synthetic_problem(N5, 1)
ls = 2.0 * (1.0 + 0.1 * np.arange(9))
bg = BatchedARDGP(length_scale=ls, noise_level=0.1, alpha=1e-4, normalize_y=True, optimizer=None, predict_dtype="float32").fit(X, Y)
th = bg.thetas
bl0, g0 = bg.log_marginal_likelihood(th, eval_gradient=True, fused=False)
t0, _ = wall(lambda: bg.log_marginal_likelihood(th, eval_gradient=True, fused=False), reps=5)
bg.log_marginal_likelihood(th, eval_gradient=True, fused=True)
t, (bl, g) = wall(lambda: bg.log_marginal_likelihood(th, eval_gradient=True, fused=True), reps=5)
print(f"N={N5}: three GPs, streams {t0*1e3:.3f} ms, fused batch {t*1e3:.3f} ms; lml rel diff {np.max(np.abs((bl-bl0)/bl0)):.2e}, grad rel diff {np.max(np.abs(g-g0))/np.max(np.abs(g0)):.2e}")
