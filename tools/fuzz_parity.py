#!/usr/bin/env python3
"""Randomised parity sweep: random (N, D, P, M, hyper-parameters, ARD or isotropic, predict dtype, variance
method) through the estimator seam against the CPU oracle, with the documented tolerances as pass bars.  The long
form of tests/test_gpu_fuzz.py (which runs 54 fixed-seed cases in the suite): this is the tool that hunts for
size-dependent indexing mistakes (tile edges, zero band, super-tile gating, ragged query batches).  `FUZZ_CASES` (default 150), `FUZZ_SEED`, `FUZZ_MAX_N` (default 3000)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("GPK_DEBUG_FILL", "nan")   # poison fresh device buffers: reads of unwritten memory become NaN
import numpy as np  # noqa: E402
from oracle import gp_oracle as O  # noqa: E402
from unmanned_aerial_vehicles_amd import BatchedARDGP, GaussianProcess, RBF, ConstantKernel, GaussianProcessRegressor, WhiteKernel  # noqa: E402

cases = int(os.environ.get("FUZZ_CASES", "150"))
seed = int(os.environ.get("FUZZ_SEED", "0"))
max_n = int(os.environ.get("FUZZ_MAX_N", "3000"))
rng = np.random.default_rng(seed)
EDGE = [4095, 4096, 4097, 3071, 3072, 3200, 4224, 5120, 6144, 1, 2, 3, 31, 32, 33, 63, 64, 65, 127, 128, 129, 255, 256, 257, 383, 384, 385, 511, 512, 513, 1023, 1024, 1025,
        2047, 2048, 2049]


def pick(hi):
    if rng.random() < 0.5:
        return int(rng.choice([e for e in EDGE if e <= hi]))
    return int(rng.integers(1, hi + 1))


def rel(a, b):
    return float(np.max(np.abs(a - b)) / max(np.max(np.abs(b)), 1e-300))


bad = 0
t0 = time.time()
for c in range(cases):
    N, M = pick(max_n), pick(max_n)
    D, P = int(rng.integers(1, 17)), int(rng.integers(1, 13))
    ard = rng.random() < 0.4
    ls = np.exp(rng.uniform(np.log(0.6), np.log(3.0), D)) * np.sqrt(D) / 2 if ard else float(np.exp(rng.uniform(np.log(0.6), np.log(3.0))) * np.sqrt(D) / 2)
    sf2 = float(np.exp(rng.uniform(-1, 1))) if rng.random() < 0.5 else 1.0
    noise = float(np.exp(rng.uniform(np.log(1e-3), np.log(0.3))))
    normalize = bool(rng.random() < 0.5)
    pd = "float32" if rng.random() < 0.4 else "float64"
    vm = str(rng.choice(["auto", "inverse", "solve"] + (["inverse_split", "inverse_split2"] if pd == "float32" else [])))
    X = rng.standard_normal((N, D))
    Y = np.sin(X @ rng.standard_normal((D, P))) + 0.1 * rng.standard_normal((N, P))
    Xq = rng.standard_normal((M, D)) * rng.choice([0.5, 1.0, 2.0])
    kern = RBF(ls) + WhiteKernel(noise)
    if sf2 != 1.0:
        kern = ConstantKernel(sf2) * RBF(ls) + WhiteKernel(noise)
    tag = f"case {c}: N={N} M={M} D={D} P={P} ard={ard} sf2={sf2:.3g} noise={noise:.3g} norm={normalize} {pd} {vm}"
    try:
        g = GaussianProcessRegressor(kernel=kern, alpha=1e-8, normalize_y=normalize, optimizer=None, predict_dtype=pd,
                                     var_method=vm).fit(X, Y)
        mean, std = g.predict(Xq, return_std=True)
        mean_only = g.predict(Xq)
        lml = g.log_marginal_likelihood_value_
        lml2, grad = g.log_marginal_likelihood(g.kernel_.theta, eval_gradient=True)
    except Exception as e:  # noqa: BLE001
        print("EXC ", tag, repr(e), flush=True); bad += 1
        continue
    st = O.fit_fixed(X, Y, ls, sf2, noise, 1e-8, normalize)
    om, os_ = O.predict(st, Xq, return_std=True)
    olml = O.log_marginal_likelihood(st)
    ograd = O.lml_gradient(st, ard=ard)
    mean, std, om, os_ = (np.asarray(a).reshape(M, -1) for a in (mean, std, om, os_))
    e_mean, e_std = rel(mean, om), rel(std, os_)
    e_lml = abs(lml - olml) / abs(olml)
    # the oracle's gradient covers (sf2?, ls..., noise) in the kernel's own theta order when all are free
    grad = np.asarray(grad)[1:] if len(grad) == len(ograd) + 1 else np.asarray(grad)    # leading ConstantKernel term: not in the oracle
    e_grad = rel(grad, np.asarray(ograd)) if np.shape(grad) == np.shape(ograd) else float("inf")
    # fp32 mean: every term K*_mj alpha_j carries a few fp32 roundings, so the bound scales with sum_j |K*_mj alpha_j|
    cond = float(np.max(np.abs(O.rbf_cross(Xq, st.X, st.length_scale, st.signal_variance)) @ np.abs(st.alpha) * st.y_std) / max(np.max(np.abs(om)), 1e-300))
    # (fp32 kernel entries carry the rounding of the squared distance, ~1e-7 * d^2 in the exponent: a few 1e-6 relative)
    # the documented bars; fp32 with the serving gates of the estimator active
    tol_m, tol_s = (1e-4, 1e-3) if pd == "float32" else (1e-8, 1e-7)
    ok = (e_mean < tol_m and e_std < tol_s and e_lml < 1e-9 and abs(lml2 - lml) <= 1e-9 * abs(lml) and not (e_grad > 1e-6)
          and np.array_equal(np.asarray(mean_only).reshape(M, -1), mean))
    if not ok:
        bad += 1
    print(("ok   " if ok else "FAIL ") + tag + f"  mean {e_mean:.1e} std {e_std:.1e} lml {e_lml:.1e} grad {e_grad:.1e}", flush=True)
# ---- the ROS-package GP (gaussian_process.py:63-265): no target scaling, variance floored at 1e-10, its own LML
pk_cases = int(os.environ.get("FUZZ_PACKAGE_CASES", str(max(cases // 3, 1))))
for c in range(pk_cases):
    N, M = pick(min(max_n, 2000)), pick(600)
    D, P = int(rng.integers(1, 17)), int(rng.integers(1, 13))
    ls = float(np.exp(rng.uniform(np.log(0.6), np.log(3.0))) * np.sqrt(D) / 2)
    sf2, noise = float(np.exp(rng.uniform(-1, 1))), float(np.exp(rng.uniform(np.log(1e-3), np.log(0.3))))
    X = rng.standard_normal((N, D)); Y = np.sin(X @ rng.standard_normal((D, P))) + 0.1 * rng.standard_normal((N, P))
    Xq = rng.standard_normal((M, D))
    tag = f"package case {c}: N={N} M={M} D={D} P={P} ls={ls:.3g} sf2={sf2:.3g} noise={noise:.3g}"
    gp = GaussianProcess(input_dim=D, output_dim=P)
    gp.max_data_points = 10 ** 9
    gp.kernel.length_scale, gp.kernel.signal_variance, gp.noise_variance = ls, sf2, noise
    gp.add_training_data(X, Y)
    gp.fit()
    if N < 2:                         # the reference refuses to fit and predicts the prior
        m, v = gp.predict(Xq)
        ok = np.all(m == 0) and np.allclose(v, sf2)
        e_mean = e_var = e_lml = 0.0
    else:
        m, v = gp.predict(Xq)
        o = O.PackageGPOracle(ls, sf2, noise).fit(X, Y)
        om, ov = o.predict(Xq)
        e_mean, e_var = rel(m, om), rel(v, ov)
        e_lml = abs(gp.log_marginal_likelihood() - o.log_marginal_likelihood()) / abs(o.log_marginal_likelihood())
        ok = e_mean < 1e-8 and e_var < 1e-7 and e_lml < 1e-9 and v.shape == (M, P)
    bad += not ok
    print(("ok   " if ok else "FAIL ") + tag + f"  mean {e_mean:.1e} var {e_var:.1e} lml {e_lml:.1e}", flush=True)
# ---- per-axis ARD GPs served together (gp_trainer.py / pretrained_gp.py): <= 32 rows take gpk_predict_host_multi
ba_cases = int(os.environ.get("FUZZ_BATCHED_CASES", str(max(cases // 5, 1))))
for c in range(ba_cases):
    N, M = max(pick(min(max_n, 2500)), 3), int(rng.integers(1, 41))
    D, B = int(rng.integers(1, 17)), int(rng.integers(2, 9))
    X = rng.standard_normal((N, D)); Y = np.sin(X @ rng.standard_normal((D, B))) + 0.1 * rng.standard_normal((N, B))
    Xq = rng.standard_normal((M, D))
    normalize = bool(rng.random() < 0.5)
    tag = f"batched case {c}: N={N} M={M} D={D} B={B} norm={normalize}"
    bg = BatchedARDGP(length_scale=np.full(D, np.sqrt(D)), noise_level=0.05, alpha=1e-8, normalize_y=normalize, optimizer=None).fit(X, Y)
    lss, noises = [], []
    for m in bg.models:                       # distinct hyper-parameters per model
        th = m.kernel_.theta + rng.uniform(-0.4, 0.4, m.kernel_.theta.shape)
        m.kernel_.theta = th
        m._refactor()
        comp = m.kernel_.components()
        lss.append(comp.ls_vector(D)); noises.append(comp.noise)
    bg._fused = None
    mean, std = bg.predict(Xq, return_std=True)
    mean_only = bg.predict(Xq)
    e_mean = e_std = 0.0
    for b in range(B):
        st = O.fit_fixed(X, Y[:, b], lss[b], 1.0, noises[b], 1e-8, normalize)
        om, os_ = O.predict(st, Xq, return_std=True)
        e_mean, e_std = max(e_mean, rel(mean[:, b], om.ravel())), max(e_std, rel(std[:, b], os_.ravel()))
    used = M <= 32 and bg._serve is not None and bg._serve.get("ok", False)
    ok = e_mean < 1e-8 and e_std < 1e-7 and np.array_equal(mean_only, mean) and (used or M > 32)
    bad += not ok
    print(("ok   " if ok else "FAIL ") + tag + f"  mean {e_mean:.1e} std {e_std:.1e} one-call path {used}", flush=True)
print(f"{cases} + {pk_cases} + {ba_cases} cases, {bad} failures, {time.time() - t0:.0f} s")
sys.exit(1 if bad else 0)
