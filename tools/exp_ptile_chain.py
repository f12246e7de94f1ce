#!/usr/bin/env python3
"""Is the diagonal chain of the one-launch Cholesky waiting for its own tasks' backlog?  Per tile column: the column time (end of D(j)
minus end of D(j - 1)), how long before the chain arrived D(j) and the tile under it were TAKEN (lead), and how long they still
needed after their last tile column was final (tail: ~4 us when the task was ready and waiting, more when it was still
working through its backlog of finished columns).    python tools/exp_ptile_chain.py [N ...]"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def run(n, be, torch, _lib):
    path = os.path.join(ROOT, "gpurun_out", f"ptile_trace_{n}.txt")
    p = lambda t: C.c_void_p(t.data_ptr())
    X = torch.as_tensor(np.random.default_rng(0).standard_normal((n, 9)), device=be.device)
    ls = np.full(9, 2.0)
    K0 = be.empty((n, n), torch.float64)
    be.check(be.lib.gpk_gram(be.h, _lib.GPK_F64, p(X), n, 9, ls.ctypes.data_as(_lib._dp), 1.0, 0.1001, p(K0), n))
    winv = be.empty((n, 128), torch.float64)
    info = C.c_int(0)
    for it in range(3):
        K = K0.clone()
        torch.cuda.synchronize()
        if it == 2:
            be.set_options(ptile_trace_path=path)
        be.check(be.lib.gpk_potrf(be.h, p(K), n, n, p(winv), C.byref(info)))
    t = np.loadtxt(path)[:-4]
    nt = n // 128
    t0 = t[:, 0].min()
    us = lambda v: (v - t0) / 100.0
    # which tile each task was: a listed launch stamps its list word (i | j << 9 | problem << 18) into slot 15 (the diagonal tasks
    # overwrite it with a cycle stamp, but they are the tasks with an end stamp in slot 10, and any list keeps them in column
    # order); an unlisted launch is column-major.  (Update tasks of a chunked launch stamp the same word: the tile's own task is
    # the LAST one listed for it.)
    isd = t[:, 10] > 0
    e = t[:, 15].astype(np.int64)
    where = {}
    if (e[~isd] != 0).any():
        for k in np.nonzero(~isd)[0]:
            where[(int(e[k] & 511), int((e[k] >> 9) & 511))] = k
    else:
        k = 0
        for j in range(nt):
            for i in range(j, nt):
                where[(i, j)] = k; k += 1
    D = t[isd]
    if (e[~isd] != 0).any():                       # a chunked launch: every listing of a tile but the last is an update task
        last = set(where.values())
        u = np.array([k for k in np.nonzero(~isd)[0] if k not in last], dtype=np.int64)
        if len(u):
            o = np.array(sorted(last), dtype=np.int64)
            du = (t[u, 5] - t[u, 0]) / 100.0; ku = (t[u, 1] - t[u, 0]) / 100.0
            do = (t[o, 5] - t[o, 0]) / 100.0; ko = (t[o, 1] - t[o, 0]) / 100.0
            print(f"  {len(u)} update tasks: {du.mean():.1f} us each (k-loop {ku.mean():.1f}, store + publish {(du - ku).mean():.1f}), "
                  f"sum {du.sum() / 1e3:.1f} ms of workgroup time; {len(o)} own tasks: {do.mean():.1f} us each (k-loop {ko.mean():.1f}), sum {do.sum() / 1e3:.1f} ms")
    assert len(D) == nt
    F = np.array([t[where[(j + 1, j)]] for j in range(nt - 1)])
    d_done = us(D[:, 10]); d_start = us(D[:, 0]); d_avail = us(D[:, 11]); d_kend = us(D[:, 1])
    f_done = us(F[:, 5]); f_start = us(F[:, 0]); f_avail = us(F[:, 11]); f_kend = us(F[:, 1])
    total = us(max(t[:, 5].max(), t[:, 10].max()))
    print(f"N={n}: {total:.0f} us = {n ** 3 / 3 / total / 1e6:.1f} TF, {nt} tile columns, {total / nt:.1f} us per column")
    print("  columns      column time   D(j): taken before its last column was final / needed after   tile under it: the same   D(j) k-loop us per finished column")
    for lo, hi in ((1, nt // 4), (nt // 4, nt // 2), (nt // 2, 3 * nt // 4), (3 * nt // 4, nt - 1)):
        js = np.arange(max(lo, 1), hi)
        ct = np.diff(d_done)[js - 1]
        dl = d_avail[js] - d_start[js]; dt = d_kend[js] - d_avail[js]
        fl = f_avail[js] - f_start[js]; ft = f_kend[js] - f_avail[js]
        print(f"  {lo:3d}..{hi - 1:3d}   {ct.mean():7.1f} us      {dl.mean():8.1f} / {dt.mean():6.1f} us (max {dt.max():6.1f})"
              f"                {fl.mean():8.1f} / {ft.mean():6.1f} us (max {ft.max():6.1f})      {((d_kend[js] - d_start[js]) / js).mean():6.2f}")


def main():
    import torch
    from tools import gpk_opts
    from unmanned_aerial_vehicles_amd import _lib
    from unmanned_aerial_vehicles_amd.device import get_backend
    gpk_opts.install()
    be = get_backend(0)
    for n in [int(a) for a in sys.argv[1:]] or [8192]:
        run(n, be, torch, _lib)


if __name__ == "__main__":
    main()
