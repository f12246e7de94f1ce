#!/usr/bin/env python3
"""Where the time of one LML(+gradient) evaluation goes at small N (the reference's own workload sizes)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402
from unmanned_aerial_vehicles_amd.device import DeviceGP, get_backend  # noqa: E402

be = get_backend(0)


def wall(fn, reps=20):
    fn(); torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    return sorted(ts)[len(ts) // 2] * 1e3


for N in (1000, 4096):
    rng = np.random.default_rng(0)
    X = rng.standard_normal((N, 9)); Y = np.sin(X @ rng.standard_normal((9, 3)))
    dev = DeviceGP(X, Y, be)
    r = {}
    r["gram"] = wall(lambda: dev.gram(2.0, 1.0, 0.1001))
    r["gram+potrf"] = wall(lambda: dev.factorize(2.0, 1.0, 0.1001))
    dev.factorize(2.0, 1.0, 0.1001)
    def inv():
        dev._Winv = {}
        dev.inverse_factor(False)
    r["trtri"] = wall(inv)
    r["alpha(inv)"] = wall(lambda: dev.solve_alpha())
    r["lml_terms"] = wall(lambda: dev.lml_terms())
    r["wtw+grad"] = wall(lambda: dev.lml_grad(0.1))
    def full():
        dev.factorize(2.0, 1.0, 0.1001); dev.solve_alpha(); dev.lml_terms(); dev.lml_grad(0.1)
    r["full LML+grad eval"] = wall(full)
    print(N, {k: round(v, 3) for k, v in r.items()}, flush=True)
