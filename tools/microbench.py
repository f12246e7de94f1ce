#!/usr/bin/env python3
"""Kernel microbenchmarks on one MI355X (HIP-event timing on the stream the kernels run on).

    python tools/microbench.py [gemm] [gram] [potrf] [predict] [--n 16384]
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def _p(t):
    return C.c_void_p(t.data_ptr())


def timed(fn, iters=5, warmup=2):
    import torch
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(iters)]
    for a, b in evs:
        a.record()
        fn()
        b.record()
    torch.cuda.synchronize()
    ts = sorted(a.elapsed_time(b) for a, b in evs)
    return ts[len(ts) // 2] * 1e-3, ts[0] * 1e-3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("what", nargs="*", default=["gemm", "gram", "potrf", "predict"])
    ap.add_argument("--n", type=int, default=16384)
    ap.add_argument("--m", type=int, default=10000)
    args = ap.parse_args()
    import torch
    from unmanned_aerial_vehicles_amd import _lib
    from unmanned_aerial_vehicles_amd.device import DeviceGP, get_backend
    from bench import synthetic_problem
    be = get_backend(0)
    out = {}
    if "gemm" in args.what:
        for dt, tdt, code in (("f64", torch.float64, _lib.GPK_F64), ("f32", torch.float32, _lib.GPK_F32)):
            for (m, n, k) in ((4096, 4096, 4096), (8192, 8192, 8192), (8192, 8192, 512)):
                A = torch.randn((m, k), dtype=tdt, device=be.device)
                B = torch.randn((n, k), dtype=tdt, device=be.device)
                Cm = torch.zeros((m, n), dtype=tdt, device=be.device)
                for ta, tb in ((0, 0), (0, 1), (1, 1)):
                    lda = k if not ta else m
                    ldb = k if not tb else n
                    Av = A if not ta else A.reshape(k, m)
                    Bv = B if not tb else B.reshape(k, n)

                    def run():
                        be.bind_stream()
                        be.check(be.lib.gpk_gemm_tiles(be.h, code, ta, tb, _p(Av), lda, _p(Bv), ldb, _p(Cm), n, m, n,
                                                       k, 1.0, 0.0, 0))
                    med, best = timed(run)
                    tf = 2.0 * m * n * k / med / 1e12
                    out[f"gemm_{dt}_{m}x{n}x{k}_t{ta}{tb}"] = {"s": med, "TFLOPs": tf}
                    print(f"gemm {dt} {m}x{n}x{k} ta={ta} tb={tb}: {med*1e3:.3f} ms  {tf:.1f} TFLOP/s", flush=True)
    N = args.n
    X, Y, Xq = synthetic_problem(N, args.m)
    if "gram" in args.what or "potrf" in args.what or "predict" in args.what:
        Yn = (Y - Y.mean(0)) / Y.std(0)
        dev = DeviceGP(X, Yn, be)
    if "gram" in args.what:
        med, best = timed(lambda: dev.gram(2.0, 1.0, 0.1001))
        gb = (dev.Np ** 2 * 8 + N * 9 * 8) / med / 1e9
        out["gram_f64"] = {"N": N, "s": med, "GBps": gb}
        print(f"gram f64 N={N}: {med*1e3:.3f} ms  {gb:.0f} GB/s (best {best*1e3:.3f} ms)", flush=True)
        Xf = dev.X.float()
        Kf = torch.empty((dev.Np, dev.Np), dtype=torch.float32, device=be.device)
        ls = np.full(9, 2.0)

        def run32():
            be.bind_stream()
            be.check(be.lib.gpk_gram(be.h, _lib.GPK_F32, _p(Xf), N, 9, ls.ctypes.data_as(_lib._dp), 1.0, 0.1001,
                                     _p(Kf), dev.Np))
        med, best = timed(run32)
        gb = (dev.Np ** 2 * 4 + N * 9 * 4) / med / 1e9
        out["gram_f32"] = {"N": N, "s": med, "GBps": gb}
        print(f"gram f32 N={N}: {med*1e3:.3f} ms  {gb:.0f} GB/s", flush=True)
        del Kf
    if "potrf" in args.what or "predict" in args.what:
        def fac():
            dev.factorize(2.0, 1.0, 0.1001)
        t0 = time.perf_counter()
        fac()
        torch.cuda.synchronize()
        t1 = time.perf_counter() - t0
        t0 = time.perf_counter()
        fac()
        torch.cuda.synchronize()
        t2 = time.perf_counter() - t0
        gf = N ** 3 / 3 / t2 / 1e9
        out["potrf_f64"] = {"N": N, "s_first": t1, "s": t2, "GFLOPs": gf}
        print(f"gram+potrf f64 N={N}: {t2*1e3:.1f} ms ({t1*1e3:.1f} first)  {gf:.0f} GFLOP/s", flush=True)
        t0 = time.perf_counter()
        dev.solve_alpha()
        torch.cuda.synchronize()
        print(f"potrs N={N}: {(time.perf_counter()-t0)*1e3:.1f} ms", flush=True)
    if "predict" in args.what:
        for dt in ("float64", "float32"):
            q = torch.as_tensor(Xq, device=be.device, dtype=torch.float64 if dt == "float64" else torch.float32)
            med, _ = timed(lambda: dev.predict_mean_dev(q, np.zeros(3), np.ones(3), dt), iters=3, warmup=1)
            print(f"predict mean {dt} N={N} M={args.m}: {med*1e3:.2f} ms  {args.m/med:.0f} pred/s", flush=True)
            out[f"mean_{dt}"] = {"s": med, "pred_per_s": args.m / med}
            med, _ = timed(lambda: dev.predict_var_dev(q, 1.1, 0.0, dt), iters=3, warmup=1)
            fl = N * N * args.m / med / 1e12
            print(f"predict var {dt} N={N} M={args.m}: {med*1e3:.2f} ms  {args.m/med:.0f} pred/s  {fl:.1f} TFLOP/s(N^2 M)",
                  flush=True)
            out[f"var_{dt}"] = {"s": med, "pred_per_s": args.m / med, "TFLOPs": fl}
    os.makedirs("gpurun_out", exist_ok=True)
    with open("gpurun_out/microbench.json", "w") as f:
        json.dump(out, f, indent=1)


if __name__ == "__main__":
    main()
