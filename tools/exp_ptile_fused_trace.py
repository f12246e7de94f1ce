#!/usr/bin/env python3
"""Per-task time stamps (GPK_PTILE_TRACE) of the one-launch factorisation WITH the tiles of the inverse factor in its task list
(gpk_lml_eval with a gradient), column by column: the diagonal chain, the last factor tile and the last W^T tile of every column.
    python tools/exp_ptile_fused_trace.py [N] [fused: 1 | 0 = value-only evaluation, plain factorisation]"""
import ctypes as C, os, sys, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch
from unmanned_aerial_vehicles_amd.device import DeviceGP, get_backend
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
fused = int(sys.argv[2]) if len(sys.argv) > 2 else 1
be = get_backend(0)
rng = np.random.default_rng(0)
X = rng.standard_normal((n, 9)); Y = rng.standard_normal((n, 1))
dev = DeviceGP(X, Y, be)
path = os.path.join(ROOT, "gpurun_out", f"ptile_trace_fused_{n}.txt")
for it in range(3):
    if it == 2: os.environ["GPK_PTILE_TRACE"] = path
    dev.lml_eval(2.0, 1.0, 0.1001, 0.1, bool(fused))
os.environ.pop("GPK_PTILE_TRACE", None)
t = np.loadtxt(path)[:-4]
nt = n // 128
t0 = t[:, 0].min(); us = lambda v: (v - t0) / 100.0
print(f"N={n} fused={fused}: tasks {len(t)}, total {us(t[:, :11].max()):.1f} us")
idx = 0; prevD = None
for c in range(nt):
    per = nt if fused else nt - c
    rows = t[idx:idx + per]; idx += per
    D = rows[0]
    Tdone = [us(r[5]) for r in rows[1:nt - c]]
    Idone = [us(r[5]) for r in rows[nt - c:]] if fused else []
    Istart = [us(r[0]) for r in rows[nt - c:]] if fused else []
    Ik = [us(r[1]) for r in rows[nt - c:]] if fused else []
    dD = us(D[10]) - prevD if prevD is not None else 0.0
    prevD = us(D[10])
    if c < 4 or c % 4 == 0 or c >= nt - 3:
        print(f"col {c:2d}: D start {us(D[0]):7.1f} kloop {us(D[1]):7.1f} done {us(D[10]):7.1f} (+{dD:5.1f}) | T done max {max(Tdone) if Tdone else 0:7.1f} | "
              f"INV start min {min(Istart) if Istart else 0:7.1f} kloop-end max {max(Ik) if Ik else 0:7.1f} done max {max(Idone) if Idone else 0:7.1f}")
