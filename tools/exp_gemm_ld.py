#!/usr/bin/env python3
"""Experiment: tile-GEMM throughput vs operand leading dimension (power-of-two stride -> L2 set conflicts?)."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from tools.microbench import timed  # noqa: E402
from unmanned_aerial_vehicles_amd import _lib  # noqa: E402
from unmanned_aerial_vehicles_amd.device import get_backend  # noqa: E402

be = get_backend(0)
for dt, tdt, code in (("f32", torch.float32, _lib.GPK_F32), ("f64", torch.float64, _lib.GPK_F64)):
    for (m, n, k) in ((8192, 8192, 8192), (16384, 10240, 16384)):
        for pad in (0, 16, 32, 64, 544):
            lda = k + pad
            A = torch.randn((m * lda,), dtype=tdt, device=be.device)
            B = torch.randn((n * lda,), dtype=tdt, device=be.device)
            Cm = torch.zeros((m, n), dtype=tdt, device=be.device)

            def run():
                be.bind_stream()
                be.check(be.lib.gpk_gemm_tiles(be.h, code, 0, 0, C.c_void_p(A.data_ptr()), lda, C.c_void_p(B.data_ptr()),
                                               lda, C.c_void_p(Cm.data_ptr()), n, m, n, k, 1.0, 0.0, 0))
            med, best = timed(run, iters=3, warmup=1)
            print(f"gemm {dt} {m}x{n}x{k} ld=k+{pad}: {med*1e3:.3f} ms  {2.0*m*n*k/med/1e12:.1f} TFLOP/s", flush=True)
            del A, B, Cm
