#!/usr/bin/env python3
"""A/B of the fp16 x 2 variance launch: LDS-staged 512 x 128 tiles (`inverse_split2`) against the fragment-order launch
whose operands go from L2 straight into registers (`inverse_split2f`), same box, same factor, same queries.  First a
correctness sweep over small shapes (every tile height, ragged N and M), then the headline shape: kernel time from the
library's own event brackets, agreement of the two results, error against the fp64 kernels."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402
from unmanned_aerial_vehicles_amd import _lib  # noqa: E402
from unmanned_aerial_vehicles_amd.device import DeviceGP, get_backend  # noqa: E402

be = get_backend(0)


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
        be.check(be.lib.gpk_set_option(be.h, b"k5_split2_tile", tile))
        v2 = dev.predict_var_dev(q, kss, 0.0, "float32", "inverse_split2")
        DeviceGP.SPLIT2F_LAYOUT = 1
        vf = dev.predict_var_dev(q, kss, 0.0, "float32", "inverse_split2f")
        DeviceGP.SPLIT2F_LAYOUT = 2
        vg = dev.predict_var_dev(q, kss, 0.0, "float32", "inverse_split2f")
        be.check(be.lib.gpk_set_option(be.h, b"k5_split2_tile", 0))
        rel = lambda v: float(((v.sqrt() - v64.sqrt()).abs() / v64.sqrt()).max())  # noqa: E731
        print(f"N={N} M={M} tile={tile}: std err vs fp64: lds {rel(v2):.2e}  direct {rel(vf):.2e}  direct16 {rel(vg):.2e};  "
              f"max |direct - lds| / kss {float((vf - v2).abs().max()) / kss:.2e}  max |direct16 - lds| / kss "
              f"{float((vg - v2).abs().max()) / kss:.2e}", flush=True)
        assert rel(vf) < max(2 * rel(v2), 1e-5) and rel(vg) < max(2 * rel(v2), 1e-5), "direct launch disagrees"
        del dev

if os.environ.get("HEADLINE", "1") == "1":
    N = int(os.environ.get("N", "65536"))
    M = int(os.environ.get("M", "10000"))
    reps = int(os.environ.get("REPS", "6"))
    dev = model(N)
    q = queries(M)
    v64 = dev.predict_var_dev(q.double(), 1.1, 0.0, "float64", "inverse")
    forms = os.environ.get("FORMS", "lds,direct:24,direct16:24,direct16:0,lds,direct:24,direct16:24").split(",")
    dev.inverse_factor(True)
    dev._Winv.pop("f64", None)
    ops = {}
    if any(f.startswith("lds") for f in forms):
        ops["lds"] = dev.split2_inverse_factor()
    for name, lay in (("direct", 1), ("direct16", 2)):
        if any(f.partition(":")[0] == name for f in forms):
            DeviceGP.SPLIT2F_LAYOUT = lay
            ops[name] = dev.split2f_inverse_factor()
            dev._Winv.pop("split2f")
    dev._Winv.pop("f32", None)
    ref = None
    for form in forms:
        name, _, sync = form.partition(":")
        method = "inverse_split2" if name == "lds" else "inverse_split2f"
        if name != "lds":
            DeviceGP.SPLIT2F_LAYOUT = 1 if name == "direct" else 2
            dev._Winv["split2f"] = ops[name]
        if sync:
            be.check(be.lib.gpk_set_option(be.h, b"k5_direct_sync", int(sync)))
        dev.predict_var_dev(q, 1.1, 0.0, "float32", method)
        dev.timing(True)
        for _ in range(reps):
            v = dev.predict_var_dev(q, 1.1, 0.0, "float32", method)
        ms = dev.kernel_times(_lib.GPK_TIMED_K5)
        dev.timing(False)
        err = float(((v.sqrt() - v64.sqrt()).abs() / v64.sqrt()).max())
        ref = v.clone() if ref is None else ref
        print(f"{form:12s}: kernel {np.mean(ms):8.2f} ms (min {np.min(ms):.2f} max {np.max(ms):.2f}) = "
              f"{3.0 * N * N * M / np.mean(ms) / 1e9:7.1f} TF fp16 issued, {M / np.mean(ms):6.1f} k pred/s kernel-only; "
              f"std err vs fp64 {err:.2e}; max |v - v(first)| {float((v - ref).abs().max()):.2e}", flush=True)
