#!/usr/bin/env python3
"""K5-shaped GEMMs: dense (constant k) vs the triangular K5 launch at the same flops, fp32."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402
from unmanned_aerial_vehicles_amd import _lib  # noqa: E402
from unmanned_aerial_vehicles_amd.device import get_backend  # noqa: E402

be = get_backend(0)
N, Mp = 65536, 10112
W = torch.randn((N, N // 2), dtype=torch.float32, device=be.device)          # dense operand: k = N / 2
Kq = torch.randn((Mp, N // 2), dtype=torch.float32, device=be.device)
Cm = torch.empty((N, Mp), dtype=torch.float32, device=be.device)


def ev(fn, reps=3):
    fn(); torch.cuda.synchronize(); ts = []
    for _ in range(reps):
        a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(b) * 1e-3)
    return sorted(ts)[len(ts) // 2]


def dense():
    be.bind_stream()
    be.check(be.lib.gpk_gemm_tiles(be.h, _lib.GPK_F32, 0, 0, C.c_void_p(W.data_ptr()), N // 2, C.c_void_p(Kq.data_ptr()), N // 2,
                                   C.c_void_p(Cm.data_ptr()), Mp, N, Mp, N // 2, 1.0, 0.0, 0))
t = ev(dense)
print(f"dense  {N} x {Mp} x {N//2}: {t*1e3:.1f} ms  {2.0*N*Mp*(N//2)/t/1e12:.1f} TFLOP/s", flush=True)
del W, Kq, Cm
# the real K5 launch (triangular W, sum-of-squares epilogue)
Wf = torch.randn((N, N), dtype=torch.float32, device=be.device)
X = torch.randn((N, 9), dtype=torch.float32, device=be.device)
Xq = torch.randn((10000, 9), dtype=torch.float32, device=be.device)
work = torch.empty((N * Mp,), dtype=torch.float32, device=be.device)
var = torch.empty((Mp,), dtype=torch.float64, device=be.device)
ls = np.full(9, 2.0)
def k5():
    be.bind_stream()
    be.check(be.lib.gpk_predict_var_inv(be.h, _lib.GPK_F32, C.c_void_p(X.data_ptr()), N, 9, ls.ctypes.data_as(_lib._dp), 1.0,
                                        C.c_void_p(Wf.data_ptr()), N, N, C.c_void_p(Xq.data_ptr()), 10000, 1.1, 0.0,
                                        C.c_void_p(work.data_ptr()), C.c_void_p(var.data_ptr())))
t = ev(k5)
print(f"K5 (triangular, fused epilogue): {t*1e3:.1f} ms  {float(N)*N*10000/t/1e12:.1f} TFLOP/s (algorithmic)", flush=True)
