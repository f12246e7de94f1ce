#!/usr/bin/env python3
"""Calibration of the fp32 mean gate: actual fp32 mean error (vs the fp64 kernels) against the 1-norm and 2-norm
amplification estimates sum_j |k_j alpha_j| / |mean| and sqrt(sum_j (k_j alpha_j)^2) / |mean| for random models."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402
from unmanned_aerial_vehicles_amd.device import DeviceGP, get_backend  # noqa: E402

be = get_backend(0)
rng = np.random.default_rng(5)
rows = []
cases = [(65536, 9, 3, 2.0, 1.0, 0.1001)] if "--big" in sys.argv else []
for c in range(int(os.environ.get("CASES", "40"))):
    N = int(rng.integers(200, 4000))
    D, P = int(rng.integers(1, 17)), int(rng.integers(1, 9))
    ls = float(np.exp(rng.uniform(np.log(0.6), np.log(3.0))) * np.sqrt(D) / 2)
    sf2 = float(np.exp(rng.uniform(-1, 1)))
    noise = float(np.exp(rng.uniform(np.log(1e-3), np.log(0.3))))
    cases.append((N, D, P, ls, sf2, noise))
for (N, D, P, ls, sf2, noise) in cases:
    r = np.random.default_rng(N + D)
    X = r.standard_normal((N, D))
    Y = np.sin(X @ r.standard_normal((D, P))) + 0.1 * r.standard_normal((N, P))
    Yn = (Y - Y.mean(0)) / Y.std(0)
    dev = DeviceGP(X, Yn, be)
    dev.factorize(ls, sf2, noise)
    dev.solve_alpha()
    # QUERIES=n: the batch size (default: <= 2000, the size the round-2 constants were calibrated on; the worst error of a
    # batch grows slowly with its size, so the gate's constants are bounds for batches up to 2^20 only if measured there)
    nq = int(os.environ.get("QUERIES", "0")) or min(2000, max(64, N))
    Xq = r.standard_normal((nq, D))
    z, o = np.zeros(P), np.ones(P)
    m64 = dev.predict_mean_dev(Xq, z, o, "float64")
    errs = {}
    for kern in ("valu", "mfma"):
        if kern == "mfma" and dev.mean_kernel_choice() != "mfma":
            errs[kern] = float("nan")
            continue
        m32 = dev.predict_mean_dev(Xq, z, o, "float32", kern).double()
        errs[kern] = float((m32 - m64).abs().max() / m64.abs().max())
    # amplification estimates on training rows (what the gate can afford) and on the queries themselves
    out = {}
    for name, q in (("train", dev.X[torch.linspace(0, N - 1, min(N, 1024), device=be.device).round().long()].contiguous()),
                    ("query", be.upload(Xq))):
        b = dev.predict_mean_dev(q, z, o, "float64", "valu").abs().amax()
        a1 = dev.predict_mean_dev(q, z, o, "float64", "valu", _alpha=dev.alpha.abs()).amax()
        keep_ls, keep_sf2 = dev.ls, dev.sf2
        dev.ls, dev.sf2 = keep_ls / np.sqrt(2.0), keep_sf2 ** 2
        a2 = dev.predict_mean_dev(q, z, o, "float64", "valu", _alpha=dev.alpha ** 2).amax().sqrt()
        dev.ls, dev.sf2 = keep_ls, keep_sf2
        out[name] = (float(a1 / b), float(a2 / b))
    print(f"N={N:6d} D={D:2d} P={P} ls={ls:5.2f} sf2={sf2:5.2f} noise={noise:7.4f}  err valu {errs['valu']:.1e} mfma {errs['mfma']:.1e}  "
          f"A1 train {out['train'][0]:9.1f} query {out['query'][0]:9.1f}  A2 train {out['train'][1]:8.2f} query {out['query'][1]:8.2f}  "
          f"err/A2q valu {errs['valu'] / out['query'][1]:.1e} mfma {errs['mfma'] / out['query'][1]:.1e}", flush=True)
    del dev
