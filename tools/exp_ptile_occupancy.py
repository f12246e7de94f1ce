#!/usr/bin/env python3
"""Where the workgroup-time of the one-launch Cholesky goes at a size where throughput, not the diagonal chain, is the bound:
per-task time stamps (option ptile_trace_path) summed over all off-diagonal tasks - k-loop time per k-tile against the matrix pipe's
rate, time spent waiting for tile columns, the closing product, idle between tasks.    python tools/exp_ptile_occupancy.py [N]"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
    path = os.path.join(ROOT, "gpurun_out", f"ptile_trace_{n}.txt")
    import torch
    from tools import gpk_opts
    from unmanned_aerial_vehicles_amd import _lib
    from unmanned_aerial_vehicles_amd.device import get_backend
    gpk_opts.install()
    be = get_backend(0)
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
    total = us(max(t[:, 5].max(), t[:, 10].max()))
    rows = []
    idx = 0
    for j in range(nt):
        for i in range(j, nt):
            r = t[idx]; idx += 1
            if i > j + 2:                       # plain off-diagonal tasks
                rows.append((j, us(r[0]), us(r[11]) if r[11] > 0 else us(r[0]), us(r[1]), us(r[2]), us(r[5])))
    a = np.array(rows)
    j, start, avail, kend, wready, done = a.T
    nkt = 8 * j
    slots = 512 if nt > 96 else 256
    wg_time = slots * total
    busy = (done - start).sum()
    # k-loop: from the task's start to the end of its k-loop; "waiting" = the part before the last column was seen final
    wait_cols = np.clip(avail - start, 0, None)
    print(f"N={n}: {total:.0f} us, {len(a)} plain off-diagonal tasks on {slots} workgroup slots; {n ** 3 / 3 / total / 1e6:.1f} TF")
    print(f"  workgroup-time in tasks          {busy / wg_time:6.3f} of slots x total (the rest: chain tasks, idle between tasks, start-up and tail)")
    print(f"  k-loops (start -> k-loop end)    {(kend - start).sum() / wg_time:6.3f}   = {((kend - start).sum() / nkt.sum()):.3f} us per k-tile per workgroup (matrix-pipe floor with {slots // 256} workgroup(s) per CU: {1.707 * slots / 256:.2f})")
    late = wait_cols > 5.0
    print(f"  ... of which before the last tile column was final (tasks that waited > 5 us: {late.mean():.2f} of them)   {wait_cols.sum() / wg_time:6.3f}")
    kt_after = (kend - np.maximum(start, avail))
    print(f"  k-loop end -> W_jj ready         {(wready - kend).sum() / wg_time:6.3f}")
    print(f"  closing product + store + publish {(done - wready).sum() / wg_time:6.3f}")
    for lo, hi in ((0, nt // 4), (nt // 4, nt // 2), (nt // 2, 3 * nt // 4), (3 * nt // 4, nt)):
        m = (j >= max(lo, 1)) & (j < hi)
        if m.any():
            print(f"  columns {lo:3d}..{hi - 1:3d}: {((kend - start)[m].sum() / nkt[m].sum()):.3f} us per k-tile, waited for columns {wait_cols[m].sum() / (kend - start)[m].sum():.2f} of the k-loop time, "
                  f"W wait {((wready - kend)[m].mean()):.1f} us, closing {((done - wready)[m].mean()):.1f} us per task")


if __name__ == "__main__":
    main()
