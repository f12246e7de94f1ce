#!/usr/bin/env python3
"""K4 timing: exact-difference VALU kernel vs the MFMA kernel (fp32), N_train = 65536, D = 9, P = 3."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402
from unmanned_aerial_vehicles_amd.device import DeviceGP, get_backend  # noqa: E402

be = get_backend(0)
N = int(os.environ.get("EXP_N", "65536"))
rng = np.random.default_rng(0)
X = rng.standard_normal((N, 9)); W = rng.standard_normal((9, 3))
Y = np.sin(X @ W) + 0.1 * rng.standard_normal((N, 3))
dev = DeviceGP(X, Y, be)
dev.ls, dev.sf2 = np.full(9, 2.0), 1.0
dev.set_alpha(rng.standard_normal((N, 3)) * 0.01)
print("choice", dev.mean_kernel_choice(), dev._r2)
for M in (25, 10000, 1 << 20):
    Xq = be.upload(np.random.default_rng(1).standard_normal((M, 9)), torch.float32)
    res = {}
    for kern in ("valu", "mfma"):
        f = lambda: dev.predict_mean_dev(Xq, np.zeros(3), np.ones(3), "float32", kern)
        out = f(); torch.cuda.synchronize()
        ts = []
        for _ in range(5):
            a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
            a.record(); out = f(); b.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(b))
        res[kern] = (sorted(ts)[2], out.double().cpu().numpy())
    d = np.max(np.abs(res["valu"][1] - res["mfma"][1])) / np.max(np.abs(res["valu"][1]))
    if M <= 10000:      # both fp32 kernels against the fp64 kernel
        ref = dev.predict_mean_dev(Xq.double(), np.zeros(3), np.ones(3), "float64").cpu().numpy()
        sc = np.max(np.abs(ref))
        print(f"   vs fp64: valu {np.max(np.abs(res['valu'][1] - ref)) / sc:.2e}  mfma {np.max(np.abs(res['mfma'][1] - ref)) / sc:.2e}")
    pairs = float(M) * N
    print(f"M={M}: valu {res['valu'][0]:.3f} ms  mfma {res['mfma'][0]:.3f} ms  ({pairs/res['mfma'][0]/1e9:.1f} Gpair/s, "
          f"{pairs*41/res['mfma'][0]/1e9:.1f} alg TFLOP/s)  max diff/scale {d:.2e}", flush=True)
