#!/usr/bin/env python3
"""Exactness sweep of the tile mapping with integer data over many grid shapes (rectangular and lower-triangular),
for whatever tile edge the library picks (GPK_GEMM_SMALL decides 64 vs 128)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from unmanned_aerial_vehicles_amd import _lib  # noqa: E402
from unmanned_aerial_vehicles_amd.device import _p, get_backend  # noqa: E402

be = get_backend(0)
bad = 0
KS = [int(a) for a in (sys.argv[1:] or ['64'])]
shapes = [(mt, nt, 0, k) for k in KS for mt in (5, 12, 19, 20, 23, 24, 31) for nt in (3, 8, 19, 20, 31)] + [(mt, mt, 1, k) for k in KS for mt in range(9, 45)]
for mt, nt, lo, k in shapes:
    m, n = 128 * mt, 128 * nt
    g = torch.Generator(device="cpu").manual_seed(mt * 1000 + nt)
    A = torch.randint(-3, 4, (m, k), generator=g).double().to(be.device)
    B = A if lo else torch.randint(-3, 4, (n, k), generator=g).double().to(be.device)
    Cm = torch.zeros((m, n), dtype=torch.float64, device=be.device)
    be.bind_stream()
    for rep in (1, 2):
        be.check(be.lib.gpk_gemm_tiles(be.h, _lib.GPK_F64, 0, 0, _p(A), k, _p(B), k, _p(Cm), n, m, n, k, 1.0, 1.0, lo))
    ref = 2.0 * (A @ B.T)
    if lo:
        # tiles strictly above the diagonal (at the granularity the kernel used) are left untouched: compare the lower part
        ok = torch.equal(torch.tril(Cm), torch.tril(ref))
    else:
        ok = torch.equal(Cm, ref)
    if not ok:
        bad += 1
        d = (torch.tril(Cm - ref) if lo else (Cm - ref)).abs()
        idx = torch.nonzero(d > 0)
        print("MISMATCH k", k, mt, nt, "lower" if lo else "rect", "first bad (row, col):", idx[0].tolist(), "count", len(idx), "tile64", [i // 64 for i in idx[0].tolist()], flush=True)
print("shapes", len(shapes), "bad", bad)
sys.exit(1 if bad else 0)
