#!/usr/bin/env python3
"""Fit -> serve at the headline size: the inverse factor (gpk_trtri_absmax) and the fp16 x 2 split that follows it (one pass over W),
wall and event times, three repetitions (the first pays the allocations).    python tools/exp_split_time.py [N]"""
import sys, os, time, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from unmanned_aerial_vehicles_amd.device import DeviceGP, get_backend
be = get_backend(0)
N = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
rng = np.random.default_rng(0)
X = rng.standard_normal((N, 9)); Y = rng.standard_normal((N, 3))
dev = DeviceGP(X, Y, be)
dev.factorize(1.5, 1.0, 0.0501)
for rep in range(3):
    dev._Winv = {}
    torch.cuda.synchronize(); t0 = time.perf_counter()
    dev.inverse_factor(False)
    torch.cuda.synchronize(); t1 = time.perf_counter()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    dev.split2_inverse_factor()
    e1.record()
    t2 = time.perf_counter()
    torch.cuda.synchronize(); t3 = time.perf_counter()
    print(f"rep {rep}: trtri {t1-t0:.4f} s; split: host returns after {(t2-t1)*1e3:.3f} ms, wall {(t3-t1)*1e3:.3f} ms, events {e0.elapsed_time(e1):.3f} ms", flush=True)
