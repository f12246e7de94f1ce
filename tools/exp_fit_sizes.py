#!/usr/bin/env python3
"""potrf / trtri / wtw+grad wall times at a few N (A/B of GEMM tile-selection thresholds via GPK_GEMM_SMALL)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402
from unmanned_aerial_vehicles_amd.device import DeviceGP, get_backend  # noqa: E402

be = get_backend(0)


def wall(fn, reps=5):
    fn(); torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    return sorted(ts)[len(ts) // 2] * 1e3


for N in [int(a) for a in (sys.argv[1:] or ["2048", "8192", "16384"])]:
    rng = np.random.default_rng(0)
    X = rng.standard_normal((N, 9)); Y = np.sin(X @ rng.standard_normal((9, 3)))
    dev = DeviceGP(X, Y, be)
    r = {}
    r["gram+potrf"] = wall(lambda: dev.factorize(2.0, 1.0, 0.1001), 3)
    dev.factorize(2.0, 1.0, 0.1001)
    def inv():
        dev._Winv = {}
        dev.inverse_factor(False)
    r["trtri"] = wall(inv, 3)
    dev.solve_alpha()
    r["wtw+grad"] = wall(lambda: dev.lml_grad(0.1), 3)
    print(N, {k: round(v, 3) for k, v in r.items()}, flush=True)
    del dev
