#!/usr/bin/env python3
"""Control-loop latency table (the reference's serving shape: `predict_residual` on ONE row, the 25-row horizon of
src/px4/mpc.py:1490-1496 at 50 Hz (:1868), and 25 x 500 Monte-Carlo rollout rows) at N_train = 1000 / 4096 / 10 000, D = 10, P = 6,
fp64: posterior mean and mean + variance, by the estimator's predict() (host arrays in and out: gpk_predict_host up to 64 rows,
the general chain above) and by the device-tensor route (DeviceGP.predict_gated_dev on a resident query tensor, results left
on the device, one synchronisation).  Medians of wall time over many calls.
    python tools/serving_latency.py [N ...]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402
from unmanned_aerial_vehicles_amd import GaussianProcessRegressor, RBF, WhiteKernel  # noqa: E402


def med(f, n):
    for _ in range(max(5, n // 20)):
        f()
    ts = []
    for _ in range(n):
        t0 = time.perf_counter(); f(); ts.append(time.perf_counter() - t0)
    ts.sort()
    return ts[n // 2] * 1e6, ts[int(n * 0.99)] * 1e6


def main():
    sizes = [int(a) for a in sys.argv[1:]] or [1000, 4096, 10000]
    print("# tools/serving_latency.py: wall time per call, microseconds, median (99th percentile); fp64; D = 10, P = 6; RBF(0.5) + White(0.1)")
    print(f"{'N_train':>8} {'rows':>7} | {'predict() mean':>22} {'predict() mean+std':>22} | {'device mean':>22} {'device mean+var':>22} | budget at 50 Hz: 20 000 us")
    for N in sizes:
        rng = np.random.default_rng(0)
        D, P = 10, 6
        X = rng.standard_normal((N, D)); Y = np.sin(X @ rng.standard_normal((D, P))) * 0.05 + 0.01 * rng.standard_normal((N, P))
        gp = GaussianProcessRegressor(kernel=RBF(0.5) + WhiteKernel(0.1), alpha=1e-4, normalize_y=True, optimizer=None).fit(X, Y)
        dev = gp._dev
        comp = gp.kernel_.components()
        kss = comp.sf2 + (comp.noise or 0.0)
        for M in (1, 25, 25 * 500):
            q = rng.standard_normal((M, D))
            qd = dev.be.upload(q)
            n = 2000 if M <= 25 else 200

            def dmean():
                dev.predict_gated_dev(qd, gp._y_train_mean, gp._y_train_std, None, 0.0, "float64")
                torch.cuda.synchronize()

            def dvar():
                dev.predict_gated_dev(qd, gp._y_train_mean, gp._y_train_std, kss, 0.0, "float64")
                torch.cuda.synchronize()

            a = med(lambda: gp.predict(q), n)
            b = med(lambda: gp.predict(q, return_std=True), n)
            c = med(dmean, n)
            d = med(dvar, n)
            f = lambda t: f"{t[0]:10.1f} ({t[1]:8.1f})"
            print(f"{N:8d} {M:7d} | {f(a):>22} {f(b):>22} | {f(c):>22} {f(d):>22} |", flush=True)


if __name__ == "__main__":
    main()
