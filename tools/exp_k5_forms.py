#!/usr/bin/env python3
"""A/B of the two forms of the bf16 x 3 variance launch at the headline shape (N = 65 536, 10 000 queries): form 1
(32x32x16 MFMAs, register-staged LDS) vs form 2 (16x16x32 fused-term MFMAs, LDS filled by DMA), same box, same
factor, same queries; kernel time from the library's own event brackets, agreement of the two results, error vs fp64."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402
from unmanned_aerial_vehicles_amd import _lib  # noqa: E402
from unmanned_aerial_vehicles_amd.device import DeviceGP, get_backend  # noqa: E402

N = int(os.environ.get("N", "65536"))
M = int(os.environ.get("M", "10000"))
reps = int(os.environ.get("REPS", "6"))
be = get_backend(0)
rng = np.random.default_rng(0)
X = rng.standard_normal((N, 9))
Y = np.sin(X @ rng.standard_normal((9, 3))) + 0.1 * rng.standard_normal((N, 3))
Yn = (Y - Y.mean(0)) / Y.std(0)
dev = DeviceGP(X, Yn, be)
dev.factorize(2.0, 1.0, 0.1001)
dev.split_inverse_factor()
q = torch.as_tensor(np.random.default_rng(1).standard_normal((M, 9)), dtype=torch.float32, device=be.device)
v64 = dev.predict_var_dev(q.double(), 1.1, 0.0, "float64", "inverse")
dev._Winv.pop("f64", None)
if os.environ.get("ZERO") == "1":
    # diagnostic: all-zero operands (results are meaningless): W3 zeroed, and length-scales so short that every k* underflows
    # to 0 - the same instruction stream on data that toggles no bits.  A large speed-up means the launch is limited by
    # power (the clock the chip holds under this load), not by its instruction schedule.
    dev._Winv["split"].zero_()
    dev.ls = np.full(9, 1e-3)
res = {}
for form in [int(f) for f in os.environ.get("FORMS", "1,2,1,2").split(",")]:
    be.check(be.lib.gpk_set_option(be.h, b"k5_split_form", form))
    dev.predict_var_dev(q, 1.1, 0.0, "float32", "inverse_split")
    dev.timing(True)
    for _ in range(reps):
        v = dev.predict_var_dev(q, 1.1, 0.0, "float32", "inverse_split")
    ms = dev.kernel_times(_lib.GPK_TIMED_K5)
    dev.timing(False)
    err = float(((v.sqrt() - v64.sqrt()).abs() / v64.sqrt()).max())
    res.setdefault(form, []).append(v.clone())
    print(f"form {form}: kernel {np.mean(ms):8.2f} ms (min {np.min(ms):.2f} max {np.max(ms):.2f}) = "
          f"{6.0 * N * N * M / np.mean(ms) / 1e9:7.1f} TF bf16, {M / np.mean(ms) * 1e3 / 1e3:6.1f} k pred/s kernel-only; "
          f"std err vs fp64 {err:.2e}", flush=True)
if os.environ.get("SPLIT2", "1") == "1":
    if os.environ.get("ZERO") == "1":
        dev.ls = np.full(9, 2.0)                 # (split2_inverse_factor needs the real W once more)
        dev._Winv.pop("split", None)
        dev.split2_inverse_factor()
        dev._Winv["split2"][0].zero_()
        dev.ls = np.full(9, 1e-3)
    else:
        dev.split2_inverse_factor()
        dev._Winv.pop("split", None)
    # SPLIT2_TILES: tile variants of the fp16 x 2 launch (option k5_split2_tile - 1: 64 x 64 per wave in 256 x 128 tiles;
    # 2: 128 x 64 per wave in 512 x 128 tiles; 0: the library's own rule)
    ref2 = None
    for form in [int(f) for f in os.environ.get("SPLIT2_TILES", "0").split(",")]:
        be.check(be.lib.gpk_set_option(be.h, b"k5_split2_tile", form))
        dev.predict_var_dev(q, 1.1, 0.0, "float32", "inverse_split2")
        dev.timing(True)
        for _ in range(reps):
            v2 = dev.predict_var_dev(q, 1.1, 0.0, "float32", "inverse_split2")
        ms = dev.kernel_times(_lib.GPK_TIMED_K5)
        dev.timing(False)
        err = float(((v2.sqrt() - v64.sqrt()).abs() / v64.sqrt()).max())
        ref2 = v2.clone() if ref2 is None else ref2
        print(f"fp16x2 split (tile option {form}): kernel {np.mean(ms):8.2f} ms (min {np.min(ms):.2f} max {np.max(ms):.2f}) = "
              f"{3.0 * N * N * M / np.mean(ms) / 1e9:7.1f} TF fp16, {M / np.mean(ms):6.1f} k pred/s kernel-only; "
              f"std err vs fp64 {err:.2e}; max |v - v(first form)| {float((v2 - ref2).abs().max()):.2e}; scale {dev._Winv['split2'][1]:g}", flush=True)
    be.check(be.lib.gpk_set_option(be.h, b"k5_split2_tile", 0))
if 1 in res and 2 in res:
    print("max |form1 - form2| =", float((res[1][0] - res[2][0]).abs().max()), " repeatable:",
          all(torch.equal(a, res[f][0]) for f in res for a in res[f]))
