#!/usr/bin/env python3
"""Which stage is wrong at a given N: L L^T = K (potrf), W L = I (trtri), alpha by both solve paths."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402
from unmanned_aerial_vehicles_amd.device import DeviceGP, get_backend  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 4961
rng = np.random.default_rng(N)
X = rng.standard_normal((N, 7)); Y = np.sin(X @ rng.standard_normal((7, 2))) + 0.1 * rng.standard_normal((N, 2))
Y = (Y - Y.mean(0)) / Y.std(0)
dev = DeviceGP(X, Y, get_backend(0))
dev.gram(2.0, 1.0, 0.03)
K = dev.K.clone()
dev.factorize(2.0, 1.0, 0.03)
L = torch.tril(dev.K)
E = (L @ L.T - torch.tril(K) - torch.tril(K, -1).T)
print("N", N, "Np", dev.Np, "potrf: max |L L^T - K| =", float(E.abs().max()), "at", np.unravel_index(int(E.abs().argmax()), E.shape))
W = dev.inverse_factor(False)
Wl = torch.tril(W)
E2 = Wl @ L - torch.eye(dev.Np, dtype=torch.float64, device=W.device)
print("trtri: max |W L - I| =", float(E2.abs().max()), "at", np.unravel_index(int(E2.abs().argmax()), E2.shape), " max |upper band of W| =", float(torch.triu(W, 1)[:, :].abs().max()))
for method in ("chain", "inverse"):
    dev.solve_alpha(method)
    a = dev.alpha_host()
    Kh = (torch.tril(K) + torch.tril(K, -1).T)[:N, :N].cpu().numpy()
    r = Kh @ a - Y
    print("alpha via", method, ": max |K alpha - y| =", float(np.abs(r).max()))
