#!/usr/bin/env python3
"""gpk_potrf: one-launch tile Cholesky (gpk_ptile.hip) against the recursive launch chain, same matrix, same box.
    python tools/exp_ptile.py [sizes ...]"""
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
    from tools import gpk_opts
    gpk_opts.install()                                     # GPK_OPTS="name=value,..."
    be = get_backend(0)
    sizes = [int(a) for a in sys.argv[1:]] or [1024, 2048, 4096, 8192, 16384]
    p = lambda t: C.c_void_p(t.data_ptr())
    for n in sizes:
        rng = np.random.default_rng(0)
        X = torch.as_tensor(rng.standard_normal((n, 9)), device=be.device)
        ls = np.full(9, 2.0)
        ld = n + int(os.environ.get('PAD', '0'))          # PAD=32: leading dimension off the power of two
        K0 = be.empty((n, ld), torch.float64)
        be.check(be.lib.gpk_gram(be.h, _lib.GPK_F64, p(X), n, 9, ls.ctypes.data_as(_lib._dp), 1.0, 0.1001, p(K0), ld))
        winv = be.empty((n, 128), torch.float64)
        info = C.c_int(0)
        res = {}
        for mode in (0, 1, 0, 1):
            be.check(be.lib.gpk_set_option(be.h, b"ptile", mode))
            best = 1e30
            for _ in range(4):
                K = K0.clone()
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                be.check(be.lib.gpk_potrf(be.h, p(K), n, ld, p(winv), C.byref(info)))
                best = min(best, time.perf_counter() - t0)
            res.setdefault(mode, []).append(best)
            if mode == 0:
                Lref, Wref = torch.tril(K[:, :n]).clone(), winv.clone()
            else:
                dl = float((torch.tril(K[:, :n]) - Lref).abs().max() / Lref.abs().max())
                dw = float((winv - Wref).abs().max() / Wref.abs().max())
        be.check(be.lib.gpk_set_option(be.h, b"ptile", 1))
        fl = n ** 3 / 3
        print(f"N={n:6d}  recursion {min(res[0]) * 1e3:8.3f} ms = {fl / min(res[0]) / 1e12:6.2f} TF   one launch "
              f"{min(res[1]) * 1e3:8.3f} ms = {fl / min(res[1]) / 1e12:6.2f} TF   |dL| {dl:.1e} |dW| {dw:.1e}", flush=True)
        del K0, K, Lref, Wref, winv


if __name__ == "__main__":
    main()
