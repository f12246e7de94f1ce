#!/usr/bin/env python3
"""Race hunt for the one-launch Cholesky's device-side protocol (ticket order, progress words, 16-column hand-overs): the same
factorisation REPS times per shape - single problems and batches - every result bit-identical to the first and the launch never
aborted; buffers poisoned with NaN before every run; then the fused factor + inverse-factor launch the same way,
single problems, batches, and three launches at once from three host threads.    python tools/stress_ptile.py [REPS]"""
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
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 100
    p = lambda t: C.c_void_p(t.data_ptr())
    bad = 0
    t0 = time.time()
    for n, B in ((640, 1), (1024, 1), (1152, 3), (2048, 1), (2176, 2), (4096, 1), (4096, 3), (8192, 1), (1024, 8)):
        rng = np.random.default_rng(n + B)
        X = torch.as_tensor(rng.standard_normal((n, 7)), device=be.device)
        K0 = be.empty((B, n, n), torch.float64)
        for b in range(B):
            ls = np.full(7, 1.5 + 0.3 * b)
            be.check(be.lib.gpk_gram(be.h, _lib.GPK_F64, p(X), n, 7, ls.ctypes.data_as(_lib._dp), 1.0, 0.05 + 0.01 * b, p(K0[b]), n))
        ref = None
        for r in range(reps):
            K = K0.clone()
            winv = torch.full((B, n, 128), float("nan"), dtype=torch.float64, device=be.device)
            info = (C.c_int * 8)()
            if B > 1:
                be.check(be.lib.gpk_batch_begin(be.h, B))
                be.check(be.lib.gpk_batch_buffer(be.h, p(K), n * n * 8))
                be.check(be.lib.gpk_batch_buffer(be.h, p(winv), n * 128 * 8))
            try:
                be.check(be.lib.gpk_potrf(be.h, p(K), n, n, p(winv), info))
            finally:
                if B > 1:
                    be.lib.gpk_batch_end(be.h)
            L = torch.tril(K)
            if ref is None:
                ref = (L.clone(), winv.clone())
                assert bool(torch.isfinite(L).all()) and bool(torch.isfinite(winv).all())
            elif not (torch.equal(L, ref[0]) and torch.equal(winv, ref[1])):
                bad += 1
                print(f"MISMATCH N={n} B={B} run {r}: |dL| {float((L - ref[0]).abs().max()):.3e}", flush=True)
        print(f"N={n:5d} B={B}: {reps} runs bit-identical" if bad == 0 else f"N={n} B={B}: {bad} mismatches so far", flush=True)
    # factor + inverse factor in the one launch (gpk_lml_eval with a gradient, Np <= ptile_inv_max_np): gradient and K^-1
    from unmanned_aerial_vehicles_amd.device import DeviceGP
    for n, P in ((640, 1), (1000, 3), (2048, 1), (3000, 2), (4096, 3), (5120, 1)):
        rng = np.random.default_rng(n)
        X = rng.standard_normal((n, 7))
        Yn = rng.standard_normal((n, P))
        dev = DeviceGP(X, Yn, be)
        ref = None
        for r in range(reps):
            if getattr(dev, "_Kinv", None) is not None:
                dev._Kinv.fill_(float("nan"))
            ld, quad, g = dev.lml_eval(1.5, 1.0, 0.0501, 0.05, True)
            Ki = torch.tril(dev._Kinv[:n, :n])
            if ref is None:
                ref = (ld, quad.copy(), g.copy(), Ki.clone())
                assert bool(torch.isfinite(Ki).all()) and np.isfinite(g).all()
            elif not (ld == ref[0] and np.array_equal(quad, ref[1]) and np.array_equal(g, ref[2]) and torch.equal(Ki, ref[3])):
                bad += 1
                print(f"MISMATCH fused N={n} P={P} run {r}: |dKinv| {float((Ki - ref[3]).abs().max()):.3e}", flush=True)
        print(f"fused N={n:5d} P={P}: {reps} runs bit-identical" if bad == 0 else f"fused N={n} P={P}: {bad} mismatches so far", flush=True)
    # the same for a batch (gpk_lml_batched: three problems in the one launch while their rows together stay below the bound)
    from unmanned_aerial_vehicles_amd import BatchedARDGP
    for n in (640, 1000, 1500):
        rng = np.random.default_rng(n + 1)
        X = rng.standard_normal((n, 7))
        Y = rng.standard_normal((n, 3))
        bg = BatchedARDGP(length_scale=1.5 * np.ones(7), noise_level=0.1, alpha=1e-4, normalize_y=True, optimizer=None).fit(X, Y)
        ref = None
        for r in range(reps):
            l, g = bg.log_marginal_likelihood(bg.thetas, eval_gradient=True, fused=True)
            if ref is None:
                ref = (l.copy(), g.copy())
                assert np.isfinite(l).all() and np.isfinite(g).all()
            elif not (np.array_equal(l, ref[0]) and np.array_equal(g, ref[1])):
                bad += 1
                print(f"MISMATCH fused batch N={n} run {r}", flush=True)
        print(f"fused N={n:5d} B=3: {reps} runs bit-identical" if bad == 0 else f"fused N={n} B=3: {bad} mismatches so far", flush=True)
    # three fused launches at once from three host threads on private handles and streams (how the optimiser's restarts and
    # the per-axis models run): each thread's evaluations bit-identical to the single-threaded reference
    from concurrent.futures import ThreadPoolExecutor
    from unmanned_aerial_vehicles_amd.device import worker_backends
    for n in (1000, 2048, 4096):
        rng = np.random.default_rng(n + 7)
        X = rng.standard_normal((n, 7))
        Yn = rng.standard_normal((n, 2))
        ref = DeviceGP(X, Yn, be).lml_eval(1.5, 1.0, 0.0501, 0.05, True)
        workers = worker_backends(be.device_index, 3)
        torch.cuda.synchronize()

        def run(i):
            wbe, stream = workers[i]
            miss = 0
            with torch.cuda.stream(stream):
                dev = DeviceGP(X, Yn, wbe)
                for r in range(max(reps // 4, 10)):
                    ld, quad, g = dev.lml_eval(1.5, 1.0, 0.0501, 0.05, True)
                    if not (ld == ref[0] and np.array_equal(quad, ref[1]) and np.array_equal(g, ref[2])):
                        miss += 1
                stream.synchronize()
            return miss
        with ThreadPoolExecutor(max_workers=3) as ex:
            miss = sum(ex.map(run, range(3)))
        bad += miss
        print(f"fused N={n:5d} x 3 threads: {max(reps // 4, 10)} runs each bit-identical" if miss == 0 else f"fused N={n} x 3 threads: {miss} mismatches", flush=True)
    print(f"{'OK' if bad == 0 else 'FAILED'}: {bad} mismatches, {time.time() - t0:.0f} s")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
