#!/usr/bin/env python3
"""Variance paths at a large ragged size with poisoned buffers: fp64 solve (reference form) vs inverse fp64 / fp32 /
bf16x3 split, and the mean kernels (vector-ALU vs matrix-core).  usage: exp_ragged_big.py [N] [M]"""
import os
import sys

os.environ.setdefault("GPK_DEBUG_FILL", "nan")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
from unmanned_aerial_vehicles_amd.device import DeviceGP, get_backend  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
M = int(sys.argv[2]) if len(sys.argv) > 2 else 4500
rng = np.random.default_rng(N)
X = rng.standard_normal((N, 9)); Y = np.sin(X @ rng.standard_normal((9, 3))) + 0.1 * rng.standard_normal((N, 3))
Y = (Y - Y.mean(0)) / Y.std(0)
Xq = rng.standard_normal((M, 9))
dev = DeviceGP(X, Y, get_backend(0))
dev.factorize(2.0, 1.0, 0.1)
dev.solve_alpha("chain"); a1 = dev.alpha_host()
dev.solve_alpha("inverse"); a2 = dev.alpha_host()
print("alpha inverse vs chain:", float(np.max(np.abs(a1 - a2)) / np.max(np.abs(a1))))
vs = dev.predict_var_dev(Xq, 1.1, 0.0, "float64", "solve").cpu().numpy()
for dt, meth in (("float64", "inverse"), ("float32", "inverse"), ("float32", "inverse_split")):
    v = dev.predict_var_dev(Xq, 1.1, 0.0, dt, meth).cpu().numpy()
    print(dt, meth, "max rel std err vs fp64 solve: %.3e" % float(np.max(np.abs(np.sqrt(v) - np.sqrt(vs)) / np.sqrt(vs))), "finite", bool(np.all(np.isfinite(v))))
m64 = dev.predict_mean_dev(Xq, np.zeros(3), np.ones(3), "float64").cpu().numpy()
for k in ("valu", "mfma"):
    m = dev.predict_mean_dev(Xq, np.zeros(3), np.ones(3), "float32", k).double().cpu().numpy()
    print("mean fp32", k, "max rel err: %.3e" % float(np.max(np.abs(m - m64)) / np.max(np.abs(m64))))
g = dev.lml_grad(0.1)
print("lml grad finite:", bool(np.all(np.isfinite(g))), g[:3])
