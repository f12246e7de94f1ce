#!/usr/bin/env python3
"""fp64 syrk (lower tiles) vs full GEMM of the Cholesky's top-level shapes: python tools/exp_syrk.py [m k]"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from unmanned_aerial_vehicles_amd import _lib  # noqa: E402
from unmanned_aerial_vehicles_amd.device import get_backend  # noqa: E402

be = get_backend(0)
shapes = [(int(sys.argv[1]), int(sys.argv[2]))] if len(sys.argv) > 2 else [(16384, 16384), (32768, 32768)]
for m, k in shapes:
    ld = 2 * k                     # as inside the factorisation: panels of a wider matrix
    A = torch.randn((m, ld), dtype=torch.float64, device=be.device)
    Cm = torch.zeros((m, m), dtype=torch.float64, device=be.device)
    for lower in (1, 0, 1, 0):
        ts = []
        for _ in range(3):
            a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
            be.bind_stream()
            a.record()
            be.check(be.lib.gpk_gemm_tiles(be.h, _lib.GPK_F64, 0, 0, C.c_void_p(A.data_ptr()), ld, C.c_void_p(A.data_ptr()), ld,
                                           C.c_void_p(Cm.data_ptr()), m, m, m, k, -1.0, 1.0, lower))
            b.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(b) * 1e-3)
        t = sorted(ts)[1]
        fl = (m / 128) * (m / 128 + 1) / 2 * 128 * 128 * 2.0 * k if lower else 2.0 * m * m * k
        print(f"m={m} k={k} lower={lower}: {t*1e3:.2f} ms  {fl/t/1e12:.2f} TFLOP/s", flush=True)
    del A, Cm
