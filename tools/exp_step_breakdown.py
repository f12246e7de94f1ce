#!/usr/bin/env python3
"""Per-kernel times of one headline step (rocprofv3 --kernel-trace --stats -- python3 tools/exp_step_breakdown.py)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402
from unmanned_aerial_vehicles_amd.device import DeviceGP, get_backend  # noqa: E402

N = int(os.environ.get("N", "32768"))
be = get_backend(0)
rng = np.random.default_rng(0)
X = rng.standard_normal((N, 9))
Y = np.sin(X @ rng.standard_normal((9, 3))) + 0.1 * rng.standard_normal((N, 3))
dev = DeviceGP(X, (Y - Y.mean(0)) / Y.std(0), be)
dev.factorize(2.0, 1.0, 0.1001)
dev.solve_alpha()
q = torch.as_tensor(np.random.default_rng(1).standard_normal((10000, 9)), dtype=torch.float32, device=be.device)
for _ in range(6):
    dev.predict_mean_dev(q, np.zeros(3), np.ones(3), "float32")
    dev.predict_var_dev(q, 1.1, 0.0, "float32", "auto")
torch.cuda.synchronize()
print("done")
