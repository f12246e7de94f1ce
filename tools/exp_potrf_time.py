#!/usr/bin/env python3
"""potrf (and optionally trtri) wall time at N with a handle option toggled: tools/exp_potrf_time.py N option v0,v1"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402
from unmanned_aerial_vehicles_amd.device import DeviceGP, get_backend  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
opt = sys.argv[2].encode() if len(sys.argv) > 2 else b"syrk_tail"
vals = [int(v) for v in (sys.argv[3] if len(sys.argv) > 3 else "0,1,0,1").split(",")]
be = get_backend(0)
rng = np.random.default_rng(0)
X = rng.standard_normal((N, 9))
dev = DeviceGP(X, np.zeros((N, 1)), be)
ref = None
for v in vals:
    be.check(be.lib.gpk_set_option(be.h, opt, v))
    dev.gram(2.0, 1.0, 0.1001)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    dev.factorize(2.0, 1.0, 0.1001)          # gram (7 ms at 65536) + potrf
    torch.cuda.synchronize()
    t = time.perf_counter() - t0
    d = torch.diagonal(dev.K).clone()
    same = "" if ref is None else f"  diag(L) identical to the first run: {bool(torch.equal(d, ref))}"
    ref = d if ref is None else ref
    print(f"N={N} {opt.decode()}={v}: gram+potrf {t * 1e3:8.2f} ms  (potrf ~{N ** 3 / 3 / (t - 0.0066 * (N / 65536) ** 2) / 1e12:.2f} TF){same}", flush=True)
