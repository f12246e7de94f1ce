#!/usr/bin/env python3
"""Kernel timeline of ONE LML + gradient evaluation (config C5, one ARD GP, or the fused batch of three with `batch`):
    rocprofv3 --kernel-trace --output-format csv -d OUT -- python3 tools/exp_lml_trace.py N [batch]
    python3 tools/exp_lml_trace.py --join OUT      (prints start offset / duration / gap per launch of the last evaluation)"""
import csv
import glob
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def join(out):
    f = glob.glob(os.path.join(out, "**", "*kernel_trace.csv"), recursive=True)[0]
    rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
    # evaluations are separated by > 300 us of idle time (host read-back + python)
    groups, cur, last_end = [], [], None
    for r in rows:
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        if last_end is not None and s - last_end > 300000 and cur:
            groups.append(cur)
            cur = []
        cur.append((r["Kernel_Name"], s, e))
        last_end = e
    groups.append(cur)
    g = max(groups[-3:], key=len) if len(groups) >= 3 else groups[-1]
    t0 = g[0][1]
    prev = t0
    agg = {}
    for name, s, e in g:
        short = name.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:48]
        if "<" in name and "gemm_kernel" in name:
            short = name[name.index("gemm_kernel"):][:44]
        a = agg.setdefault(short, [0, 0.0])
        a[0] += 1
        a[1] += (e - s) / 1e3
        if (e - s) > 20000 or (s - prev) > 20000:
            print(f"{(s - t0) / 1e3:9.1f} us  {(e - s) / 1e3:8.1f} us  gap {(s - prev) / 1e3:7.1f}  {short}")
        prev = e
    print(f"span {(g[-1][2] - t0) / 1e3:.1f} us, {len(g)} launches")
    for k, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        print(f"   {t:9.1f} us  {c:5d}  {k}")


def main():
    if sys.argv[1] == "--join":
        return join(sys.argv[2])
    import numpy as np
    import torch
    from bench import synthetic_problem
    from unmanned_aerial_vehicles_amd import BatchedARDGP, GaussianProcessRegressor
    from unmanned_aerial_vehicles_amd.kernels import RBF, ConstantKernel, WhiteKernel
    N = int(sys.argv[1])
    X, Y, _ = synthetic_problem(N, 1)
    ls = 2.0 * (1.0 + 0.1 * np.arange(9))
    if len(sys.argv) > 2:
        bg = BatchedARDGP(length_scale=ls, noise_level=0.1, alpha=1e-4, normalize_y=True, optimizer=None).fit(X, Y)
        f = lambda: bg.log_marginal_likelihood(bg.thetas, eval_gradient=True, fused=True)
    else:
        kern = ConstantKernel(1.0, "fixed") * RBF(ls, (0.1, 10.0)) + WhiteKernel(0.1, (1e-5, 1e1))
        gb = GaussianProcessRegressor(kernel=kern, alpha=1e-4, normalize_y=True, optimizer=None).fit(X, Y[:, 0])
        f = lambda: gb._lml_on_device(gb.kernel_.theta, True)
    for _ in range(4):
        torch.cuda.synchronize()
        time.sleep(0.01)
        t0 = time.perf_counter()
        f()
        print(f"eval {1e3 * (time.perf_counter() - t0):.3f} ms", flush=True)


if __name__ == "__main__":
    main()
