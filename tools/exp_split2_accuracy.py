#!/usr/bin/env python3
"""Accuracy of the three fp32 forms of the variance launch against the fp64 kernels over random models: the exact-fp32
MFMA, the bf16 x 3 split (six products per block) and the fp16 x 2 split (three products per block).  Reports the worst
relative error of the standard deviation and of the quantity the launch actually computes, |W k*|^2 (relative to k**)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402
from unmanned_aerial_vehicles_amd.device import DeviceGP, get_backend  # noqa: E402

be = get_backend(0)
rng = np.random.default_rng(11)
cases = []
if "--big" in sys.argv:
    cases.append((65536, 9, 2.0, 1.0, 0.1001, 1.0))
for c in range(int(os.environ.get("CASES", "30"))):
    N = int(rng.integers(300, 6000))
    D = int(rng.integers(1, 17))
    ls = float(np.exp(rng.uniform(np.log(0.6), np.log(3.0))) * np.sqrt(D) / 2)
    sf2 = float(np.exp(rng.uniform(-3, 3)))                      # prior variance 0.05 .. 20
    noise = float(np.exp(rng.uniform(np.log(1e-4), np.log(0.3)))) * sf2
    qscale = float(rng.choice([0.3, 1.0, 2.0, 4.0]))             # queries inside / outside the data
    cases.append((N, D, ls, sf2, noise, qscale))
worst = {"inverse": 0.0, "inverse_split": 0.0, "inverse_split2": 0.0}
for (N, D, ls, sf2, noise, qscale) in cases:
    r = np.random.default_rng(N * 7 + D)
    X = r.standard_normal((N, D))
    dev = DeviceGP(X, np.zeros((N, 1)), be)
    dev.factorize(ls, sf2, noise)
    M = 10000 if N > 60000 else 1500
    Xq = r.standard_normal((M, D)) * qscale
    Xq[:32] = X[:32]                                             # training points: the smallest variances
    kss = sf2 + noise
    q32 = be.upload(Xq, torch.float32)
    v64 = dev.predict_var_dev(q32.double(), kss, 0.0, "float64", "inverse")
    line = f"N={N:6d} D={D:2d} ls={ls:5.2f} sf2={sf2:7.3f} noise/sf2={noise / sf2:8.1e} q x{qscale:3.1f}  var/kss min {float(v64.min()) / kss:8.1e} |"
    for m in ("inverse", "inverse_split", "inverse_split2"):
        v = dev.predict_var_dev(q32, kss, 0.0, "float32", m)
        e_std = float(((v.sqrt() - v64.sqrt()).abs() / v64.sqrt()).max())
        e_ss = float((v - v64).abs().max() / kss)                # error of |W k*|^2 relative to the prior variance
        worst[m] = max(worst[m], e_ss)
        line += f" {m}: std {e_std:.1e} ss/kss {e_ss:.1e} |"
    if "split2" in dev._Winv:
        sc = dev._Winv["split2"][1]
        line += f" row-block scales {float(sc.min()):g} .. {float(sc.max()):g}"
    print(line, flush=True)
    del dev
print("worst |d ss| / kss:", {k: f"{v:.2e}" for k, v in worst.items()})
