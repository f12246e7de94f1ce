#!/usr/bin/env python3
"""Where a bench.py C3 step spends its time beyond the K5 launch chain (HIP events around each piece)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402
from unmanned_aerial_vehicles_amd.device import DeviceGP, get_backend  # noqa: E402

N = int(os.environ.get("EXP_N", "65536")); M = 10000
be = get_backend(0)
rng = np.random.default_rng(0)
X = rng.standard_normal((N, 9)); Y = np.sin(X @ rng.standard_normal((9, 3)))
dev = DeviceGP(X, Y, be)
dev.factorize(2.0, 1.0, 0.1001); dev.inverse_factor(False); dev.inverse_factor(True); dev.solve_alpha(); dev._Winv.pop("f64", None); dev._f32_data()
q32 = torch.as_tensor(np.random.default_rng(1).standard_normal((M, 9)), dtype=torch.float32, device=be.device)
ystd2 = torch.ones(3, dtype=torch.float64, device=be.device)


def ev(fn, reps=5):
    fn(); torch.cuda.synchronize(); ts = []
    for _ in range(reps):
        a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
        a.record(); r = fn(); b.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(b))
    return sorted(ts)[len(ts) // 2]


mean = dev.predict_mean_dev(q32, np.zeros(3), np.ones(3), "float32")
var = dev.predict_var_dev(q32, 1.1, 0.0, "float32", "inverse")
print("K4 mean            %.3f ms" % ev(lambda: dev.predict_mean_dev(q32, np.zeros(3), np.ones(3), "float32")))
print("K5 predict_var_dev %.3f ms" % ev(lambda: dev.predict_var_dev(q32, 1.1, 0.0, "float32", "inverse")))
print("cat/outer          %.3f ms" % ev(lambda: torch.cat([mean.double(), var[:, None] * ystd2[None, :]], dim=1)))
def step():
    m = dev.predict_mean_dev(q32, np.zeros(3), np.ones(3), "float32")
    v = dev.predict_var_dev(q32, 1.1, 0.0, "float32", "inverse")
    return torch.cat([m.double(), v[:, None] * ystd2[None, :]], dim=1)
print("whole step         %.3f ms" % ev(step))
import time
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(5): step()
torch.cuda.synchronize(); print("5 steps wall/5     %.3f ms" % ((time.perf_counter() - t0) / 5 * 1e3))
