#!/usr/bin/env python3
"""The variance launch (K5) at the headline shape, form by form, on one box: `split2[:n]` = the fp16 x 2 launch with its
operands going from L2 straight into registers (the fp32 serving default),
`bf16x3` = the exact three-way bf16 split (LDS-staged), `fp32` = the exact-fp32 MFMA GEMM.  Kernel time from the
library's own event brackets, agreement with the first form, error against the fp64 kernels.  With SWEEP=1 (default) a
correctness sweep over small shapes first: every tile height of the fp16 x 2 launch, ragged N and M, a low-noise model.
(Round 3's A/B against the round-2 LDS-staged fp16 x 2 kernel and the 16x16x32 variant: profiles/r03_k5_direct_forms_ab.log.)"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402
from unmanned_aerial_vehicles_amd import _lib  # noqa: E402
from unmanned_aerial_vehicles_amd.device import DeviceGP, get_backend  # noqa: E402

be = get_backend(0)
METHOD = {"split2": "inverse_split2", "bf16x3": "inverse_split", "fp32": "inverse"}


def model(N, noise=0.1, seed=0):
    rng = np.random.default_rng(seed)
    X = rng.standard_normal((N, 9))
    Y = np.sin(X @ rng.standard_normal((9, 3))) + 0.1 * rng.standard_normal((N, 3))
    Yn = (Y - Y.mean(0)) / Y.std(0)
    dev = DeviceGP(X, Yn, be)
    dev.factorize(2.0, 1.0, noise + 1e-4)
    return dev


def queries(M, seed=1):
    return torch.as_tensor(np.random.default_rng(seed).standard_normal((M, 9)), dtype=torch.float32, device=be.device)


if os.environ.get("SWEEP", "1") == "1":
    for N, M, tile in [(1000, 64, 0), (1000, 300, 1), (2048, 1000, 0), (4096, 1024, 0), (4096, 10000, 0), (4096, 10000, 1),
                       (8192, 10000, 0), (8192, 10000, 2), (5000, 3000, 0), (8200, 9000, 0)]:
        dev = model(N, 0.1 if N != 5000 else 1e-3)
        q = queries(M)
        kss = 1.0 + (0.1 if N != 5000 else 1e-3)
        v64 = dev.predict_var_dev(q.double(), kss, 0.0, "float64", "inverse")
        v32 = dev.predict_var_dev(q, kss, 0.0, "float32", "inverse")
        be.check(be.lib.gpk_set_option(be.h, b"k5_split2_tile", tile))
        v2 = dev.predict_var_dev(q, kss, 0.0, "float32", "inverse_split2")
        be.check(be.lib.gpk_set_option(be.h, b"k5_split2_tile", 0))
        rel = lambda v: float(((v.sqrt() - v64.sqrt()).abs() / v64.sqrt()).max())  # noqa: E731
        print(f"N={N} M={M} tile={tile}: std err vs fp64: fp32 MFMA {rel(v32):.2e}  fp16x2 {rel(v2):.2e}", flush=True)
        assert rel(v2) < max(2 * rel(v32), 1e-5), "fp16 x 2 launch disagrees"
        del dev

if os.environ.get("HEADLINE", "1") == "1":
    N = int(os.environ.get("N", "65536"))
    M = int(os.environ.get("M", "10000"))
    reps = int(os.environ.get("REPS", "6"))
    dev = model(N)
    q = queries(M)
    v64 = dev.predict_var_dev(q.double(), 1.1, 0.0, "float64", "inverse")
    forms = os.environ.get("FORMS", "split2,bf16x3,fp32,split2").split(",")
    names = {f.partition(":")[0] for f in forms}
    dev.inverse_factor(True)
    dev._Winv.pop("f64", None)
    if "split2" in names:
        dev.split2_inverse_factor()
    if "bf16x3" in names:
        dev.split_inverse_factor()
    if "fp32" not in names:
        dev._Winv.pop("f32", None)
    ref = None
    for form in forms:
        name, _, sync = form.partition(":")
        dev.predict_var_dev(q, 1.1, 0.0, "float32", METHOD[name])
        dev.timing(True)
        for _ in range(reps):
            v = dev.predict_var_dev(q, 1.1, 0.0, "float32", METHOD[name])
        ms = dev.kernel_times(_lib.GPK_TIMED_K5)
        dev.timing(False)
        err = float(((v.sqrt() - v64.sqrt()).abs() / v64.sqrt()).max())
        ref = v.clone() if ref is None else ref
        per = {"split2": 3.0, "bf16x3": 6.0, "fp32": 1.0}[name]
        print(f"{form:12s}: kernel {np.mean(ms):8.2f} ms (min {np.min(ms):.2f} max {np.max(ms):.2f}) = "
              f"{float(N) * N * M / np.mean(ms) / 1e9:6.1f} TF fp32-equivalent, {per * N * N * M / np.mean(ms) / 1e9:7.1f} TF issued, "
              f"{M / np.mean(ms):6.1f} k pred/s kernel-only; std err vs fp64 {err:.2e}; max |v - v(first)| {float((v - ref).abs().max()):.2e}",
              flush=True)
