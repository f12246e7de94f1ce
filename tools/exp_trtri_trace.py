#!/usr/bin/env python3
"""The launches of one gpk_trtri (level-by-level inverse factor) in order, with their durations: run under
    rocprofv3 --kernel-trace --output-format csv -d DIR -- python3 tools/exp_trtri_trace.py N
then    python3 tools/exp_trtri_trace.py --join DIR    prints them."""
import ctypes as C
import glob
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def join(d):
    import csv
    f = sorted(glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True))[-1]
    rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
    # the LAST run of launches between two gram kernels is the traced trtri: take everything after the last potrf
    names = [r["Kernel_Name"] for r in rows]
    last = max(k for k, n in enumerate(names) if "ptile_potrf" in n)
    rows = rows[last + 1:]
    t0 = int(rows[0]["Start_Timestamp"])
    tot = 0.0
    by = {}
    for r in rows:
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        n = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "")[:60]
        g = f"{r['Grid_Size_X']}x{r['Grid_Size_Y']}"
        print(f"{(s - t0) / 1e3:9.1f} us {(e - s) / 1e3:8.1f} us  {n}  grid {g}")
        by[n] = by.get(n, 0.0) + (e - s) / 1e3
        tot += (e - s) / 1e3
    print(f"span {(int(rows[-1]['End_Timestamp']) - t0) / 1e3:.1f} us, {len(rows)} launches, sum {tot:.1f} us")
    for n, v in sorted(by.items(), key=lambda kv: -kv[1]):
        print(f"  {v:9.1f} us  {n}")


def main():
    if sys.argv[1] == "--join":
        return join(sys.argv[2])
    import torch
    from unmanned_aerial_vehicles_amd import _lib
    from unmanned_aerial_vehicles_amd.device import get_backend
    n = int(sys.argv[1])
    be = get_backend(0)
    p = lambda t: C.c_void_p(t.data_ptr())
    X = torch.as_tensor(np.random.default_rng(0).standard_normal((n, 9)), device=be.device)
    ls = np.full(9, 2.0)
    K = be.empty((n, n), torch.float64)
    winv = be.empty((n, 128), torch.float64)
    W = be.empty((n, n), torch.float64)
    work = be.empty(((n // 2 + 128) ** 2,), torch.float64)
    info = C.c_int(0)
    for it in range(3):
        be.check(be.lib.gpk_gram(be.h, _lib.GPK_F64, p(X), n, 9, ls.ctypes.data_as(_lib._dp), 1.0, 0.1001, p(K), n))
        be.check(be.lib.gpk_potrf(be.h, p(K), n, n, p(winv), C.byref(info)))
        be.check(be.lib.gpk_trtri(be.h, p(K), n, n, p(winv), p(W), n, p(work)))
        torch.cuda.synchronize()


if __name__ == "__main__":
    main()
