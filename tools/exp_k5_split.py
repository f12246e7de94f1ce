#!/usr/bin/env python3
"""K5: fp32-MFMA launch vs the exact bf16x3-split launch (accuracy against fp64, time)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402
from unmanned_aerial_vehicles_amd.device import DeviceGP, get_backend  # noqa: E402

N = int(os.environ.get("EXP_N", "16384")); M = int(os.environ.get("EXP_M", "10000"))
be = get_backend(0)
rng = np.random.default_rng(0)
X = rng.standard_normal((N, 9)); Y = np.sin(X @ rng.standard_normal((9, 3))) + 0.1 * rng.standard_normal((N, 3))
Yn = (Y - Y.mean(0)) / Y.std(0)
dev = DeviceGP(X, Yn, be)
dev.factorize(2.0, 1.0, 0.1001); dev.solve_alpha()
Xq = np.random.default_rng(1).standard_normal((M, 9))
q32 = be.upload(Xq, torch.float32)


def ev(fn, reps=3):
    fn(); torch.cuda.synchronize(); ts = []
    for _ in range(reps):
        a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
        a.record(); r = fn(); b.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(b) * 1e-3)
    return sorted(ts)[len(ts) // 2], r


kss = 1.1
if N <= 32768:
    v64 = dev.predict_var_dev(Xq, kss, 0.0, "float64", "inverse").cpu().numpy()
t32, v32 = ev(lambda: dev.predict_var_dev(q32, kss, 0.0, "float32", "inverse"))
ts, vs = ev(lambda: dev.predict_var_dev(q32, kss, 0.0, "float32", "inverse_split"))
v32, vs = v32.cpu().numpy(), vs.cpu().numpy()
fl = float(N) * N * M
print(f"N={N} M={M}: fp32 MFMA {t32*1e3:.2f} ms ({fl/t32/1e12:.1f} TF)   bf16x3 split {ts*1e3:.2f} ms ({fl/ts/1e12:.1f} TF-equivalent, {6*fl/ts/1e12:.0f} bf16 TF)", flush=True)
print("   split vs fp32 MFMA: max |dvar| %.3e (var range %.3e .. %.3e)" % (np.max(np.abs(vs - v32)), v32.min(), v32.max()))
if N <= 32768:
    e32 = np.abs(np.sqrt(v32) - np.sqrt(v64)) / np.sqrt(v64); es = np.abs(np.sqrt(vs) - np.sqrt(v64)) / np.sqrt(v64)
    print("   std rel err vs fp64: fp32 MFMA max %.2e mean %.2e   split max %.2e mean %.2e" % (e32.max(), e32.mean(), es.max(), es.mean()))
