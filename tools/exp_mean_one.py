#!/usr/bin/env python3
"""One K4 launch pair for PMC profiling: python tools/exp_mean_one.py [valu|mfma] [M]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402
from unmanned_aerial_vehicles_amd.device import DeviceGP, get_backend  # noqa: E402

kern = sys.argv[1] if len(sys.argv) > 1 else "mfma"
M = int(sys.argv[2]) if len(sys.argv) > 2 else 1 << 20
be = get_backend(0)
N = 65536
rng = np.random.default_rng(0)
X = rng.standard_normal((N, 9))
dev = DeviceGP(X, np.zeros((N, 3)), be)
dev.ls, dev.sf2 = np.full(9, 2.0), 1.0
dev.set_alpha(rng.standard_normal((N, 3)) * 0.01)
Xq = be.upload(np.random.default_rng(1).standard_normal((M, 9)), torch.float32)
for _ in range(2):
    out = dev.predict_mean_dev(Xq, np.zeros(3), np.ones(3), "float32", kern)
torch.cuda.synchronize()
print("done", kern, M)
