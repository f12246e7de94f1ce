#!/usr/bin/env python3
"""Per-task time stamps of the one-launch Cholesky (option ptile_trace_path): where a step of the critical path goes.
    python tools/exp_ptile_trace.py N"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
    path = os.path.join(ROOT, "gpurun_out", f"ptile_trace_{n}.txt")
    import torch
    from unmanned_aerial_vehicles_amd import _lib
    from unmanned_aerial_vehicles_amd.device import get_backend
    from tools import gpk_opts
    gpk_opts.install()
    be = get_backend(0)
    p = lambda t: C.c_void_p(t.data_ptr())
    X = torch.as_tensor(np.random.default_rng(0).standard_normal((n, 9)), device=be.device)
    ls = np.full(9, 2.0)
    ld = n + int(os.environ.get('PAD', '0'))
    K0 = be.empty((n, ld), torch.float64)
    be.check(be.lib.gpk_gram(be.h, _lib.GPK_F64, p(X), n, 9, ls.ctypes.data_as(_lib._dp), 1.0, 0.1001, p(K0), ld))
    winv = be.empty((n, 128), torch.float64)
    info = C.c_int(0)
    for it in range(3):
        K = K0.clone()
        torch.cuda.synchronize()
        if it == 2:
            be.set_options(ptile_trace_path=path)
        be.check(be.lib.gpk_potrf(be.h, p(K), n, ld, p(winv), C.byref(info)))
    t = np.loadtxt(path)
    sub = t[-4:].ravel()
    if sub[0] > 0:      # a library built with -DGPK_PTILE_SUBSTAMPS=<step>: one 16-column step of D(2) in shader cycles
        names = {0: "factoring wave: step start", 1: "block factored and inverted", 2: "W_bb, L_bb in LDS", 3: "barrier 1 passed", 4: "its stores issued",
                 8: "next row's wave: barrier 1 passed", 9: "L(b+1, b) = W_bb x block", 10: "diagonal block updated, stores issued", 11: "acknowledgement wait over", 12: "barrier 2 passed"}
        print("sub-steps of D(2), one 16-column step (shader cycles from the factoring wave's step start):")
        for k in sorted(names):
            if sub[k] > 0:
                print(f"   {names[k]:40s} {sub[k] - sub[0]:8.0f}")
    t = t[:-4]
    nt = n // 128
    t0 = t[:, 0].min()
    us = lambda v: (v - t0) / 100.0
    idx = 0
    print(f"N={n}: total {us(t[:, :11].max()):.1f} us")
    for j in range(nt):
        for i in range(j, nt):
            r = t[idx]
            if i == j:
                ph = " ".join(f"{us(r[2 + b]):7.1f}" for b in range(8))
                clk = (r[15] - r[14]) / max(r[10] - r[0], 1) * 100.0
                print(f"      shader clock over the task: {clk:.0f} MHz")
                # (merged task: stamps 12 / 13 are the forward substitution's first poll passed / its end)
                print(f"D({j:2d})     start {us(r[0]):7.1f} lastcol-seen {us(r[11]):7.1f} kt-8|sub-start {us(r[12]):7.1f} kt-4|sub-end {us(r[13]):7.1f} kloop {us(r[1]):7.1f} | phaseA ends {ph} | done {us(r[10]):7.1f}")
            elif i <= j + 2 or i == nt - 1:
                print(f"T({i:2d},{j:2d})  start {us(r[0]):7.1f} kloop {us(r[1]):7.1f} Wready {us(r[2]):7.1f} Wlds {us(r[3]):7.1f} "
                      f"apply {us(r[4]):7.1f} done {us(r[5]):7.1f}")
            idx += 1
        if j >= 3 and n > 1024:
            break


if __name__ == "__main__":
    main()
