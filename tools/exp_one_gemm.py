#!/usr/bin/env python3
"""One tile-GEMM launch (for PMC profiling): python tools/exp_one_gemm.py m n k f32|f64 [reps] [ld_pad]"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from unmanned_aerial_vehicles_amd import _lib  # noqa: E402
from unmanned_aerial_vehicles_amd.device import get_backend  # noqa: E402

m, n, k = (int(v) for v in sys.argv[1:4])
dt = sys.argv[4] if len(sys.argv) > 4 else "f32"
reps = int(sys.argv[5]) if len(sys.argv) > 5 else 2
pad = int(sys.argv[6]) if len(sys.argv) > 6 else 0
be = get_backend(0)
tdt, code = (torch.float32, _lib.GPK_F32) if dt == "f32" else (torch.float64, _lib.GPK_F64)
ld = k + pad
A = torch.randn((m, ld), dtype=tdt, device=be.device)
B = torch.randn((n, ld), dtype=tdt, device=be.device)
Cm = torch.zeros((m, n), dtype=tdt, device=be.device)
for _ in range(reps):
    be.bind_stream()
    be.check(be.lib.gpk_gemm_tiles(be.h, code, 0, 0, C.c_void_p(A.data_ptr()), ld, C.c_void_p(B.data_ptr()), ld,
                                   C.c_void_p(Cm.data_ptr()), n, m, n, k, 1.0, 0.0, 0))
torch.cuda.synchronize()
print("done", m, n, k, dt, "ld", ld)
