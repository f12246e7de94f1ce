#!/usr/bin/env python3
"""A/B experimental builds of libgpk.so on ONE box: for each library given, time the dense tile GEMM
(fp32, fp64), the Cholesky and the K5 variance launch in a child process (GPK_LIBRARY selects the build).

  python tools/exp_variants.py [lib1.so lib2.so ...]        (default: the in-tree libgpk.so)
"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r"""
import ctypes as C, json, os, sys, time
sys.path.insert(0, %r)
import numpy as np, torch
from unmanned_aerial_vehicles_amd import _lib
from unmanned_aerial_vehicles_amd.device import get_backend, DeviceGP
be = get_backend(0)
def ev(fn, iters=3, warm=1):
    for _ in range(warm): fn()
    torch.cuda.synchronize(); ts = []
    for _ in range(iters):
        a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(b) * 1e-3)
    return sorted(ts)[len(ts) // 2]
out = {}
for dt, tdt, code in (("f32", torch.float32, _lib.GPK_F32), ("f64", torch.float64, _lib.GPK_F64)):
    m = n = k = 8192
    A = torch.randn((m, k), dtype=tdt, device=be.device); B = torch.randn((n, k), dtype=tdt, device=be.device)
    Cm = torch.zeros((m, n), dtype=tdt, device=be.device)
    def run():
        be.bind_stream()
        be.check(be.lib.gpk_gemm_tiles(be.h, code, 0, 0, C.c_void_p(A.data_ptr()), k, C.c_void_p(B.data_ptr()), k,
                                       C.c_void_p(Cm.data_ptr()), n, m, n, k, 1.0, 0.0, 0))
    t = ev(run, 5, 2); out["gemm_" + dt + "_TF"] = round(2.0 * m * n * k / t / 1e12, 2)
    def run2():
        be.bind_stream()
        be.check(be.lib.gpk_gemm_tiles(be.h, code, 0, 1, C.c_void_p(A.data_ptr()), k, C.c_void_p(B.data_ptr()), k,
                                       C.c_void_p(Cm.data_ptr()), n, m, n, k, 1.0, 0.0, 0))
    t = ev(run2, 5, 2); out["gemm_nt_" + dt + "_TF"] = round(2.0 * m * n * k / t / 1e12, 2)
    def run3():
        be.bind_stream()
        be.check(be.lib.gpk_gemm_tiles(be.h, code, 0, 0, C.c_void_p(A.data_ptr()), k, C.c_void_p(A.data_ptr()), k,
                                       C.c_void_p(Cm.data_ptr()), n, m, m, k, -1.0, 1.0, 1))
    t = ev(run3, 5, 2); out["syrk_" + dt + "_TF"] = round(1.0 * m * (m + 128) * k / t / 1e12, 2)
    del A, B, Cm
N = int(os.environ.get("EXP_N", "32768")); M = 10000
rng = np.random.default_rng(0); X = rng.standard_normal((N, 9)); Y = np.sin(X @ rng.standard_normal((9, 3)))
dev = DeviceGP(X, Y, be)
t = ev(lambda: dev.factorize(2.0, 1.0, 0.1001), 2, 1); out["potrf_TF"] = round(N ** 3 / 3 / t / 1e12, 2)
dev.solve_alpha(); 
t0 = time.perf_counter(); dev.inverse_factor(True); torch.cuda.synchronize(); out["trtri_s"] = round(time.perf_counter() - t0, 3)
Xq = np.random.default_rng(1).standard_normal((M, 9))
Xq_d = be.upload(Xq, torch.float32)
t = ev(lambda: dev.predict_var_dev(Xq_d, 1.1, 0.0, "float32", "inverse"), 3, 1); out["k5_TF"] = round(float(N) * N * M / t / 1e12, 2)
print("RESULT " + json.dumps(out))
""" % ROOT

libs = sys.argv[1:] or [""]
for lib in libs:
    env = dict(os.environ)
    if lib:
        env["GPK_LIBRARY"] = os.path.abspath(lib)
    r = subprocess.run([sys.executable, "-c", CHILD], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    res = [ln for ln in r.stdout.splitlines() if ln.startswith("RESULT ")]
    print(os.path.basename(lib) or "libgpk.so", res[0][7:] if res else "FAILED\n" + r.stdout[-2000:], flush=True)
