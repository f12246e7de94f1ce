#!/usr/bin/env python3
"""Tile GEMMs with varying k-range, gpk_trtri and gpk_wtw per size, two settings of one option on the same factor and box; outputs
must be bit-identical.  Default: static tile mapping against the balanced persistent schedule (gemm_balanced 0 / 1);
OPT=gemm_tiny_tiles A=0 B=128: 64 x 64 tiles against 32 x 32 tiles for launches of fewer than 128 128-tiles.
    python tools/exp_balanced.py [sizes ...]"""
import ctypes as C
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch
    from unmanned_aerial_vehicles_amd import _lib
    from unmanned_aerial_vehicles_amd.device import get_backend
    be = get_backend(0)
    OPT = os.environ.get("OPT", "gemm_balanced")
    VA, VB = int(os.environ.get("A", "0")), int(os.environ.get("B", "1"))
    restore = {"gemm_balanced": 1, "gemm_tiny_tiles": 64}.get(OPT, VB)
    sizes = [int(a) for a in sys.argv[1:]] or [1024, 4096, 10112, 16384]
    p = lambda t: C.c_void_p(t.data_ptr())
    for n in sizes:
        rng = np.random.default_rng(0)
        X = torch.as_tensor(rng.standard_normal((n, 9)), device=be.device)
        ls = np.full(9, 2.0)
        K = be.empty((n, n), torch.float64)
        be.check(be.lib.gpk_gram(be.h, _lib.GPK_F64, p(X), n, 9, ls.ctypes.data_as(_lib._dp), 1.0, 0.1001, p(K), n))
        winv = be.empty((n, 128), torch.float64)
        info = C.c_int(0)
        be.check(be.lib.gpk_potrf(be.h, p(K), n, n, p(winv), C.byref(info)))
        W = be.empty((n, n), torch.float64)
        Ki = be.empty((n, n), torch.float64)
        work = be.empty(((n // 2 + 128) ** 2,), torch.float64)
        res, outs = {}, {}
        for mode in (0, 1, 0, 1):
            be.check(be.lib.gpk_set_option(be.h, OPT.encode(), VB if mode else VA))
            bt, bw = 1e30, 1e30
            for _ in range(5):
                W.fill_(float("nan"))
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                be.check(be.lib.gpk_trtri(be.h, p(K), n, n, p(winv), p(W), n, p(work)))
                torch.cuda.synchronize()
                bt = min(bt, time.perf_counter() - t0)
                t0 = time.perf_counter()
                be.check(be.lib.gpk_wtw(be.h, p(W), n, n, p(Ki), n))
                torch.cuda.synchronize()
                bw = min(bw, time.perf_counter() - t0)
            res[mode] = (bt, bw)
            outs[mode] = (W.clone(), Ki.clone()) if n <= 40000 else (W[:4096].clone(), Ki[-4096:].clone())
        same = bool(torch.equal(torch.nan_to_num(outs[0][0]), torch.nan_to_num(outs[1][0])) and torch.equal(outs[0][1].tril(), outs[1][1].tril()))
        be.check(be.lib.gpk_set_option(be.h, OPT.encode(), restore))
        fl = n ** 3 / 3
        la, lb = ("static", "balanced") if OPT == "gemm_balanced" else (f"{OPT}={VA}", f"{OPT}={VB}")
        print(f"N={n:6d}  trtri {la} {res[0][0] * 1e3:8.3f} ms ({fl / res[0][0] / 1e12:5.1f} TF)  {lb} {res[1][0] * 1e3:8.3f} ms "
              f"({fl / res[1][0] / 1e12:5.1f} TF)   wtw {la} {res[0][1] * 1e3:8.3f} ms ({fl / res[0][1] / 1e12:5.1f} TF)  {lb} "
              f"{res[1][1] * 1e3:8.3f} ms ({fl / res[1][1] / 1e12:5.1f} TF)   bit-identical {same}", flush=True)
        del K, W, Ki, work, winv, outs


if __name__ == "__main__":
    main()
