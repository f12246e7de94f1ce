#!/usr/bin/env python3
"""One-launch Cholesky: ONE global ticket counter (ptile_xcd = 0) against one task queue per XCD with the tile rows dealt
round-robin (ptile_xcd = 1), same matrix, same box; with one and with two workgroups per CU.  Factors must be bit-identical.
    python tools/exp_ptile_xcd.py [sizes ...]"""
import ctypes as C
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


MODES = tuple(int(x) for x in os.environ.get('MODES', '0,1,804,808').split(','))   # 0 ticket, 1 row queues, RRCC groups


def main():
    import torch
    from unmanned_aerial_vehicles_amd import _lib
    from unmanned_aerial_vehicles_amd.device import get_backend
    be = get_backend(0)
    sizes = [int(a) for a in sys.argv[1:]] or [4096, 8192, 10112, 12288, 16384]
    p = lambda t: C.c_void_p(t.data_ptr())
    opt = lambda k, v: be.check(be.lib.gpk_set_option(be.h, k, v))
    opt(b"ptile_xcd_min_nt", 0)
    for n in sizes:
        rng = np.random.default_rng(0)
        X = torch.as_tensor(rng.standard_normal((n, 9)), device=be.device)
        ls = np.full(9, 2.0)
        npad = (n + 127) // 128 * 128
        K0 = be.empty((npad, npad), torch.float64)
        be.check(be.lib.gpk_gram(be.h, _lib.GPK_F64, p(X), n, 9, ls.ctypes.data_as(_lib._dp), 1.0, 0.1001, p(K0), npad))
        winv = be.empty((npad, 128), torch.float64)
        info = C.c_int(0)
        out = []
        ref = None
        for single in (999, 0):
            opt(b"ptile_single_max_nt", single)
            for xcd in MODES + MODES:
                opt(b"ptile_xcd", min(xcd, 2))
                if xcd >= 2:
                    opt(b"ptile_grp_rows", xcd // 100)
                    opt(b"ptile_grp_cols", xcd % 100)
                best = 1e30
                for _ in range(5):
                    K = K0.clone()
                    torch.cuda.synchronize()
                    t0 = time.perf_counter()
                    be.check(be.lib.gpk_potrf(be.h, p(K), npad, npad, p(winv), C.byref(info)))
                    best = min(best, time.perf_counter() - t0)
                L = torch.tril(K)
                if ref is None:
                    ref = (L.clone(), winv.clone())
                same = bool(torch.equal(L, ref[0]) and torch.equal(winv, ref[1]))
                out.append((single, xcd, best, same))
        fl = n ** 3 / 3
        line = f"N={n:6d}"
        for single in (999, 0):
            for xcd in MODES:
                b = min(t for s_, x_, t, _ in out if s_ == single and x_ == xcd)
                line += f"  {'1' if single else '2'}/CU {('%dx%d' % (xcd // 100, xcd % 100)) if xcd >= 2 else ('rows' if xcd else 'ticket')} {b * 1e3:7.3f} ms {fl / b / 1e12:5.1f} TF"
        line += "  bit-identical" if all(o[3] for o in out) else "  MISMATCH"
        print(line, flush=True)
        del K0, K, L, ref, winv
    opt(b"ptile_single_max_nt", 96)
    opt(b"ptile_xcd", 1)


if __name__ == "__main__":
    main()
