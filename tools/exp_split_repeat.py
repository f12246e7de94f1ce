#!/usr/bin/env python3
"""Repeatability of the split variance launch next to the fp32-MFMA launch (same handle, shared scratch)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402
from unmanned_aerial_vehicles_amd.device import DeviceGP, get_backend  # noqa: E402

N, M = int(os.environ.get("EXP_N", "65536")), 10000
be = get_backend(0)
rng = np.random.default_rng(0)
X = rng.standard_normal((N, 9)); Y = np.sin(X @ rng.standard_normal((9, 3)))
dev = DeviceGP(X, Y, be)
dev.factorize(2.0, 1.0, 0.1001); dev.solve_alpha()
q32 = be.upload(np.random.default_rng(1).standard_normal((M, 9)), torch.float32)
ref32 = dev.predict_var_dev(q32, 1.1, 0.0, "float32", "inverse").clone()
refsp = dev.predict_var_dev(q32, 1.1, 0.0, "float32", "inverse_split").clone()
print("first: max|split - fp32| = %.3e" % float(torch.max(torch.abs(refsp - ref32))))
for i in range(6):
    a = dev.predict_var_dev(q32, 1.1, 0.0, "float32", "inverse_split")
    b = dev.predict_var_dev(q32, 1.1, 0.0, "float32", "inverse")
    print(i, "split==first split:", bool(torch.equal(a, refsp)), " fp32==first fp32:", bool(torch.equal(b, ref32)),
          " max|split-fp32| %.3e" % float(torch.max(torch.abs(a - b))), flush=True)
