#!/usr/bin/env python3
"""One fp32 serving step at the headline shape (DeviceGP.predict_packed_dev: K4, K* in split form, K5, finalise) against the
K5 launch alone (the library's event brackets): what the step costs beyond its dominant kernel.  Run it under
`rocprofv3 --kernel-trace --stats` for the per-kernel split (K* 0.70 ms, mean 0.23 ms, finalise 0.05 ms of a 93 ms step)."""
import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np, torch, time
from unmanned_aerial_vehicles_amd import _lib
from unmanned_aerial_vehicles_amd.device import DeviceGP, get_backend
be = get_backend(0)
N, M = 65536, 10000
rng = np.random.default_rng(0)
X = rng.standard_normal((N, 9)); Y = np.sin(X @ rng.standard_normal((9, 3))) + 0.1 * rng.standard_normal((N, 3))
dev = DeviceGP(X, (Y - Y.mean(0)) / Y.std(0), be)
dev.factorize(2.0, 1.0, 0.1001); dev.solve_alpha()
q = torch.as_tensor(np.random.default_rng(1).standard_normal((M, 9)), dtype=torch.float32, device=be.device)
dev.split2_inverse_factor(); dev._Winv.pop("f64", None)
ym, ys = Y.mean(0), Y.std(0)
for _ in range(3): out = dev.predict_packed_dev(q, ym, ys, 1.1)
torch.cuda.synchronize()
dev.timing(True)
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(8): out = dev.predict_packed_dev(q, ym, ys, 1.1)
b.record(); torch.cuda.synchronize()
k5 = dev.kernel_times(_lib.GPK_TIMED_K5)
print(f"step {a.elapsed_time(b)/8:.3f} ms  K5 {np.mean(k5):.3f} ms  other {a.elapsed_time(b)/8 - np.mean(k5):.3f} ms")
