#!/usr/bin/env python3
"""Experiment: Gram-build bandwidth vs leading dimension (HBM channel-conflict check)."""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from tools.microbench import timed  # noqa: E402
from unmanned_aerial_vehicles_amd import _lib  # noqa: E402
from unmanned_aerial_vehicles_amd.device import get_backend  # noqa: E402

be = get_backend(0)
N = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
X = torch.randn((N, 9), dtype=torch.float64, device=be.device)
ls = np.full(9, 2.0)
for dt, tdt, code, es in (("f64", torch.float64, _lib.GPK_F64, 8), ("f32", torch.float32, _lib.GPK_F32, 4)):
    Xd = X.to(tdt)
    for pad in (0, 16, 32, 64, 128, 256, 2048):
        ld = N + pad
        K = torch.empty((N * ld,), dtype=tdt, device=be.device)

        def run():
            be.bind_stream()
            be.check(be.lib.gpk_gram(be.h, code, C.c_void_p(Xd.data_ptr()), N, 9, ls.ctypes.data_as(_lib._dp), 1.0,
                                     0.1, C.c_void_p(K.data_ptr()), ld))
        med, best = timed(run, iters=5, warmup=2)
        print(f"gram {dt} N={N} ld=N+{pad}: {med*1e3:.3f} ms  {N*N*es/med/1e9:.0f} GB/s", flush=True)
        del K
