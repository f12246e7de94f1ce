#!/usr/bin/env python3
"""gpk_potrs_inv (alpha = W^T (W y)): two tile-GEMM launches on a 128-column panel against the two streaming passes
(k3_stream_min_np), same W, same box.   python tools/exp_k3.py [sizes ...]"""
import ctypes as C
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch
    from unmanned_aerial_vehicles_amd.device import DeviceGP, get_backend
    be = get_backend(0)
    for n in [int(a) for a in sys.argv[1:]] or [1024, 2048, 4096, 8192]:
        for P in (1, 6):
            rng = np.random.default_rng(0)
            dev = DeviceGP(rng.standard_normal((n, 9)), rng.standard_normal((n, P)), be)
            dev.factorize(2.0, 1.0, 0.1001)
            dev.inverse_factor(False)
            out = {}
            for thr in (1 << 30, 128):
                be.check(be.lib.gpk_set_option(be.h, b"k3_stream_min_np", thr))
                best = 1e30
                for _ in range(5):
                    torch.cuda.synchronize()
                    t0 = time.perf_counter()
                    dev.solve_alpha("inverse")
                    torch.cuda.synchronize()
                    best = min(best, time.perf_counter() - t0)
                out[thr] = (best, dev.alpha.clone())
            be.check(be.lib.gpk_set_option(be.h, b"k3_stream_min_np", 512))
            d = float((out[128][1] - out[1 << 30][1]).abs().max() / out[1 << 30][1].abs().max())
            print(f"N={n:6d} P={P}  gemm route {out[1 << 30][0] * 1e3:7.3f} ms   streaming {out[128][0] * 1e3:7.3f} ms   rel diff {d:.1e}", flush=True)


if __name__ == "__main__":
    main()
