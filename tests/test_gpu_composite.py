"""The composite C ABI (gpk_fit / gpk_predict / gpk_lml / gpk_export / gpk_import) through ctypes on shapes the plain-C
known-answer test (tests/c_abi/composite.c) does not reach: ragged N (identity padding inside the handle-owned
factor), D = 16 / P = 12 (fp32 mean on the exact-difference kernel), ARD length-scales, a query batch larger than one
panel (two trips through the panel loop), fp64 and fp32, against the CPU oracle."""
import ctypes as C

import numpy as np
import pytest

from conftest import relerr
from oracle import gp_oracle as O

pytestmark = pytest.mark.gpu


def _dp(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


@pytest.mark.parametrize("N,D,P,M,ard", [(2500, 16, 12, 20000, True), (777, 5, 2, 300, False), (1300, 9, 3, 17000, False)])
def test_composite_calls_against_oracle(N, D, P, M, ard):
    from unmanned_aerial_vehicles_amd import _lib
    lib = _lib.load()
    rng = np.random.default_rng(N + D)
    X = rng.standard_normal((N, D))
    Y = np.sin(X @ rng.standard_normal((D, P))) + 0.1 * rng.standard_normal((N, P))
    Xq = np.ascontiguousarray(rng.standard_normal((M, D)) * 1.2)
    ls = np.ascontiguousarray(1.0 + 0.1 * np.arange(D)) * np.sqrt(D) / 2 if ard else np.array([np.sqrt(D) / 1.5])
    sf2, noise, jitter = 1.7, 0.08, 1e-6
    st = O.fit_fixed(X, Y, ls if ard else float(ls[0]), sf2, noise, jitter, True)
    om, os_ = O.predict(st, Xq, return_std=True)
    h = C.c_void_p()
    assert lib.gpk_create(C.byref(h), 0) == _lib.GPK_OK
    try:
        def ok(rc):
            assert rc == _lib.GPK_OK, lib.gpk_last_error(h).decode()
        ok(lib.gpk_set_stream(h, C.c_void_p(-1)))                              # GPK_OWN_STREAM
        ok(lib.gpk_fit(h, _dp(X), N, D, _dp(Y), P, _dp(ls), len(ls), sf2, noise, jitter, 1))
        mean, var = np.empty((M, P)), np.empty((M, P))
        ok(lib.gpk_predict(h, Xq.ctypes.data_as(C.c_void_p), M, mean.ctypes.data_as(C.c_void_p),
                           var.ctypes.data_as(C.c_void_p), _lib.GPK_F64, 1))
        assert relerr(mean, om.reshape(M, P)) < 1e-8 and relerr(np.sqrt(var), os_.reshape(M, P)) < 1e-8
        # fp32 serving form of the same call (queries rounded to fp32 first: compare at the rounded queries)
        q32 = np.ascontiguousarray(Xq, dtype=np.float32)
        m32, v32 = np.empty((M, P), dtype=np.float32), np.empty((M, P), dtype=np.float32)
        ok(lib.gpk_predict(h, q32.ctypes.data_as(C.c_void_p), M, m32.ctypes.data_as(C.c_void_p),
                           v32.ctypes.data_as(C.c_void_p), _lib.GPK_F32, 1))
        om32, os32 = O.predict(st, q32.astype(np.float64), return_std=True)
        assert np.max(np.abs(m32 - om32.reshape(M, P))) < 1e-4 * np.max(np.abs(om32))
        assert np.max(np.abs(np.sqrt(v32.astype(np.float64)) - os32.reshape(M, P)) / os32.reshape(M, P)) < 1e-3
        # package convention: k** = sf2, floor 1e-10
        ok(lib.gpk_predict(h, Xq.ctypes.data_as(C.c_void_p), 64, mean.ctypes.data_as(C.c_void_p),
                           var.ctypes.data_as(C.c_void_p), _lib.GPK_F64, 0))
        _, os_pkg = O.predict(st, Xq[:64], return_std=True, diag_includes_noise=False)
        assert relerr(np.sqrt(var[:64]), np.maximum(os_pkg.reshape(64, P), 1e-5 * st.y_std)) < 1e-7
        # LML value and gradient (theta = log [ls..., noise])
        lml = C.c_double()
        ok(lib.gpk_lml(h, None, 0, C.byref(lml), None))
        assert abs(lml.value - O.log_marginal_likelihood(st)) < 1e-9 * abs(lml.value)
        theta = np.log(np.r_[ls, noise])
        grad = np.zeros(len(theta))
        ok(lib.gpk_lml(h, _dp(theta), len(theta), C.byref(lml), _dp(grad)))
        assert abs(lml.value - O.log_marginal_likelihood(st)) < 1e-9 * abs(lml.value)
        assert relerr(grad, O.lml_gradient(st, ard=ard)) < 1e-7
        # export -> import into a fresh handle -> identical predictions
        L, alpha = np.empty((N, N)), np.empty((N, P))
        ym, ys = np.empty(P), np.empty(P)
        n_, d_, p_ = C.c_int64(), C.c_int(), C.c_int()
        ok(lib.gpk_export(h, C.byref(n_), C.byref(d_), C.byref(p_), _dp(L), _dp(alpha), _dp(ym), _dp(ys), C.byref(lml)))
        assert (n_.value, d_.value, p_.value) == (N, D, P)
        assert relerr(alpha, st.alpha) < 1e-8 and relerr(L, np.tril(st.L)) < 1e-10 and not np.triu(L, 1).any()
        assert relerr(ym, st.y_mean) < 1e-14 and relerr(ys, st.y_std) < 1e-14
        h2 = C.c_void_p()
        assert lib.gpk_create(C.byref(h2), 0) == _lib.GPK_OK
        try:
            assert lib.gpk_set_stream(h2, C.c_void_p(-1)) == _lib.GPK_OK
            assert lib.gpk_import(h2, _dp(X), N, D, _dp(L), _dp(alpha), P, _dp(ls), len(ls), sf2, noise, _dp(ym), _dp(ys)) == _lib.GPK_OK
            mean2, var2 = np.empty((M, P)), np.empty((M, P))
            assert lib.gpk_predict(h2, Xq.ctypes.data_as(C.c_void_p), M, mean2.ctypes.data_as(C.c_void_p),
                                   var2.ctypes.data_as(C.c_void_p), _lib.GPK_F64, 1) == _lib.GPK_OK
            assert relerr(mean2, om.reshape(M, P)) < 1e-8 and relerr(np.sqrt(var2), os_.reshape(M, P)) < 1e-8
        finally:
            lib.gpk_destroy(h2)
        ok(lib.gpk_model_release(h))
        assert lib.gpk_predict(h, Xq.ctypes.data_as(C.c_void_p), 4, mean.ctypes.data_as(C.c_void_p), None, _lib.GPK_F64, 1) == _lib.GPK_BAD_ARG
    finally:
        lib.gpk_destroy(h)


@pytest.mark.parametrize("noise", [1e-3, 0.02])
def test_composite_fp32_predict_is_gated(noise):
    """gpk_predict(GPK_F32) applies the two fp32 serving gates itself: on a low-noise model, with training
    points among the queries (variances far below the prior's), the fp32 request still meets mean 1e-4 / std 1e-3 against
    the oracle - through the fp64 kernels for the whole model (noise 1e-3: the mean gate) or for the low-variance rows only
    (noise 0.02).  The raw fp32 variance launch on the same queries (DeviceGP, gated=False) misses the bar on those rows."""
    from unmanned_aerial_vehicles_amd import _lib
    from unmanned_aerial_vehicles_amd.device import DeviceGP, get_backend
    lib = _lib.load()
    N, D, P, M = 3000, 9, 3, 1500
    rng = np.random.default_rng(7)
    X = rng.standard_normal((N, D))
    Y = np.sin(X @ rng.standard_normal((D, P))) + 0.1 * rng.standard_normal((N, P))
    Xq = np.ascontiguousarray(np.vstack([X[:600], rng.standard_normal((M - 600, D))]), dtype=np.float32)
    ls, sf2, jitter = np.array([2.0]), 1.0, 1e-8
    st = O.fit_fixed(X, Y, 2.0, sf2, noise, jitter, True)
    om, os_ = O.predict(st, Xq.astype(np.float64), return_std=True)
    h = C.c_void_p()
    assert lib.gpk_create(C.byref(h), 0) == _lib.GPK_OK
    try:
        assert lib.gpk_set_stream(h, C.c_void_p(-1)) == _lib.GPK_OK
        assert lib.gpk_fit(h, _dp(X), N, D, _dp(Y), P, _dp(ls), 1, sf2, noise, jitter, 1) == _lib.GPK_OK
        m32, v32 = np.empty((M, P), dtype=np.float32), np.empty((M, P), dtype=np.float32)
        assert lib.gpk_predict(h, Xq.ctypes.data_as(C.c_void_p), M, m32.ctypes.data_as(C.c_void_p),
                               v32.ctypes.data_as(C.c_void_p), _lib.GPK_F32, 1) == _lib.GPK_OK, lib.gpk_last_error(h).decode()
        e_m = np.max(np.abs(m32 - om.reshape(M, P))) / np.max(np.abs(om))
        e_s = np.max(np.abs(np.sqrt(v32.astype(np.float64)) - os_.reshape(M, P)) / os_.reshape(M, P))
        assert e_m < 1e-4 and e_s < 1e-3, (e_m, e_s)
    finally:
        lib.gpk_destroy(h)
    # the same model through the raw fp32 kernels: the low variances at the training points are where they fail
    dev = DeviceGP(X, st.Yn, get_backend(0))
    dev.factorize(2.0, sf2, noise + jitter)
    dev.solve_alpha()
    kss = sf2 + noise
    _, v_raw = dev.predict_gated_dev(Xq, st.y_mean, st.y_std, kss, 0.0, "float32", "auto", gated=False)
    _, v_gat = dev.predict_gated_dev(Xq, st.y_mean, st.y_std, kss, 0.0, "float32", "auto", gated=True)
    s_ref = os_.reshape(M, P)[:, 0] / st.y_std[0]
    e_raw = np.max(np.abs(np.sqrt(v_raw.cpu().numpy()) - s_ref) / s_ref)
    e_gat = np.max(np.abs(np.sqrt(v_gat.cpu().numpy()) - s_ref) / s_ref)
    assert e_gat < 1e-3 and e_raw > e_gat, (e_raw, e_gat)


@pytest.mark.parametrize("M", [5000, 20000])        # 20 000 queries: two trips through the 16 384-query panel loop
def test_one_call_serving_step(M):
    """gpk_predict_mean_var_split2 (K4 + K* in split form + the variance launch + un-normalise / pack / count in ONE C call)
    against the separate launches and gpk_pack_mean_var: identical rows, and the count of rows below the re-check
    threshold equals what the separate variance shows."""
    import torch
    from unmanned_aerial_vehicles_amd import _lib
    from unmanned_aerial_vehicles_amd.device import DeviceGP, get_backend
    be = get_backend(0)
    N, D, P = 4096, 9, 3
    rng = np.random.default_rng(3)
    X = rng.standard_normal((N, D))
    Y = np.sin(X @ rng.standard_normal((D, P))) + 0.1 * rng.standard_normal((N, P))
    ym, ys = Y.mean(0), Y.std(0)
    dev = DeviceGP(X, (Y - ym) / ys, be)
    dev.factorize(2.0, 1.0, 0.1001)
    dev.solve_alpha()
    Xq = np.vstack([X[:700], rng.standard_normal((M - 700, D))])
    q = be.upload(Xq, torch.float32)
    kss = 1.1
    mean = dev.predict_mean_dev(q, ym, ys, "float32")
    var = dev.predict_var_dev(q, kss, 0.0, "float32", "inverse_split2")
    packed = be.empty((M, 2 * P), torch.float64)
    be.check(be.lib.gpk_pack_mean_var(be.h, _lib.GPK_F32, C.c_void_p(mean.data_ptr()), C.c_void_p(var.data_ptr()), M, P,
                                      _dp(np.ascontiguousarray(ys)), C.c_void_p(packed.data_ptr())))
    want = var[:, None] * torch.as_tensor(ys ** 2, device=be.device)[None, :]
    assert torch.equal(packed[:, :P], mean.double()) and float(((packed[:, P:] - want).abs() / want).max()) < 4e-16
    one = dev.predict_packed_dev(q, ym, ys, kss, 0.0, "float32", "auto", gated=False)
    assert torch.equal(one, packed)
    # gated: the rows below the re-check fraction of the prior (raised here so that the training points among the queries
    # fall under it) are recomputed in fp64, the others are untouched
    assert dev.fp32_mean_ok(q)
    dev.FP32_VAR_RECHECK_FRACTION = 0.12
    gated = dev.predict_packed_dev(q, ym, ys, kss, 0.0, "float32", "auto", gated=True)
    low = var < dev.FP32_VAR_RECHECK_FRACTION * kss
    assert int(low.sum()) > 0 and torch.equal(gated[~low], packed[~low])
    v64 = dev.predict_var_dev(q.double(), kss, 0.0, "float64", "inverse")
    ys2 = torch.as_tensor(ys ** 2, device=be.device)
    assert float(((gated[low][:, P:] - v64[low][:, None] * ys2[None, :]).abs() / (v64[low][:, None] * ys2[None, :])).max()) < 1e-9


@pytest.mark.parametrize("N,D,B,ard", [(301, 10, 6, True), (1500, 9, 3, True), (640, 4, 2, False), (130, 16, 8, True)])
def test_batched_composite_calls_against_oracle(N, D, B, ard):
    """gpk_fit_batched / gpk_predict_batched / gpk_lml_batched: B per-axis models with their own hyper-parameters on
    shared inputs (gp_trainer.py:139-179) - odd N (row stride padding), both predict routes (<= 32 rows: one call for all
    models; panels beyond), LML values and gradients at trial thetas, a trial point that is not positive definite."""
    from unmanned_aerial_vehicles_amd import _lib
    lib = _lib.load()
    rng = np.random.default_rng(7 * N + B)
    X = rng.standard_normal((N, D))
    Y = np.ascontiguousarray(np.sin(X @ rng.standard_normal((D, B))) * (1.0 + np.arange(B)) + 0.1 * rng.standard_normal((N, B)))
    n_ls = D if ard else 1
    ls = np.ascontiguousarray(np.exp(rng.uniform(-0.3, 0.5, (B, n_ls))) * np.sqrt(D) / 1.5)
    sf2 = np.ascontiguousarray(np.exp(rng.uniform(-0.5, 0.5, B)))
    noise = np.ascontiguousarray(np.exp(rng.uniform(np.log(0.01), np.log(0.2), B)))
    jitter = 1e-6
    sts = [O.fit_fixed(X, Y[:, b], ls[b] if ard else float(ls[b, 0]), sf2[b], noise[b], jitter, True) for b in range(B)]
    h = C.c_void_p()
    assert lib.gpk_create(C.byref(h), 0) == _lib.GPK_OK
    try:
        def ok(rc):
            assert rc == _lib.GPK_OK, lib.gpk_last_error(h).decode()
        ok(lib.gpk_set_stream(h, C.c_void_p(-1)))
        info = (C.c_int * B)()
        ok(lib.gpk_fit_batched(h, B, _dp(X), N, D, _dp(Y), _dp(ls), n_ls, _dp(sf2), _dp(noise), jitter, 1, info))
        assert not any(info)
        lml = np.empty(B)
        ok(lib.gpk_lml_batched(h, None, 0, _dp(lml), None))
        assert relerr(lml, np.array([O.log_marginal_likelihood(s) for s in sts])) < 1e-10
        for M in (1, 25, 32, 33, 700):
            Xq = np.ascontiguousarray(rng.standard_normal((M, D)) * 1.1)
            mean, var = np.empty((M, B)), np.empty((M, B))
            ok(lib.gpk_predict_batched(h, _dp(Xq), M, _dp(mean), _dp(var), 1))
            mean_only = np.empty((M, B))
            ok(lib.gpk_predict_batched(h, _dp(Xq), M, _dp(mean_only), None, 1))
            assert np.array_equal(mean_only, mean)
            vpk = np.empty((M, B))
            ok(lib.gpk_predict_batched(h, _dp(Xq), M, _dp(mean_only), _dp(vpk), 0))        # package convention
            for b in range(B):
                om, os_ = O.predict(sts[b], Xq, return_std=True)
                assert relerr(mean[:, b], om.ravel()) < 1e-8 and relerr(np.sqrt(var[:, b]), os_.ravel()) < 1e-8
                _, op = O.predict(sts[b], Xq, return_std=True, diag_includes_noise=False)
                assert relerr(np.sqrt(vpk[:, b]), np.maximum(op.ravel(), 1e-5 * sts[b].y_std)) < 1e-7
        # trial hyper-parameters: values and gradients of all models from one chain
        thetas = np.ascontiguousarray(np.log(np.c_[ls, noise]) + rng.uniform(-0.3, 0.3, (B, n_ls + 1)))
        grad = np.zeros_like(thetas)
        ok(lib.gpk_lml_batched(h, _dp(thetas), n_ls + 1, _dp(lml), _dp(grad)))
        lml_only = np.empty(B)
        ok(lib.gpk_lml_batched(h, _dp(thetas), n_ls + 1, _dp(lml_only), None))
        for b in range(B):
            lsb, nb = np.exp(thetas[b, :n_ls]), float(np.exp(thetas[b, n_ls]))
            st = O.fit_fixed(X, Y[:, b], lsb if ard else float(lsb[0]), sf2[b], nb, jitter, True)
            assert abs(lml[b] - O.log_marginal_likelihood(st)) < 1e-9 * abs(lml[b]) and lml_only[b] == lml[b]
            assert relerr(grad[b], O.lml_gradient(st, ard=ard)) < 1e-7
        # the fitted state was left alone by the trial evaluation
        ok(lib.gpk_lml_batched(h, None, 0, _dp(lml), None))
        assert relerr(lml, np.array([O.log_marginal_likelihood(s) for s in sts])) < 1e-10
        ok(lib.gpk_model_release(h))
        assert lib.gpk_predict_batched(h, _dp(X), 1, _dp(np.empty(B)), None, 1) == _lib.GPK_BAD_ARG
        # bad arguments
        assert lib.gpk_fit_batched(h, 9, _dp(X), N, D, _dp(Y), _dp(ls), n_ls, _dp(sf2), _dp(noise), jitter, 1, info) == _lib.GPK_BAD_ARG
        assert lib.gpk_fit_batched(h, B, _dp(X), N, D, _dp(Y), _dp(ls), 2 if D != 2 else 3, _dp(sf2), _dp(noise), jitter, 1, info) == _lib.GPK_BAD_ARG
    finally:
        lib.gpk_destroy(h)


def test_batched_composite_not_positive_definite():
    """Duplicate training rows with zero noise and jitter: model 1's matrix is singular -> GPK_NOT_PD with the pivot in
    info[1] while model 0 (with noise) factorises; in gpk_lml_batched the same trial point gives (-inf, 0) for that model
    only (sklearn/_gpr.py:586-589)."""
    from unmanned_aerial_vehicles_amd import _lib
    lib = _lib.load()
    rng = np.random.default_rng(5)
    N, D, B = 300, 3, 2
    X = rng.standard_normal((N, D))
    X[150:] = X[:150]                                  # every row twice
    Y = np.ascontiguousarray(rng.standard_normal((N, B)))
    ls, sf2 = np.full((B, 1), 1.0), np.ones(B)
    h = C.c_void_p()
    assert lib.gpk_create(C.byref(h), 0) == _lib.GPK_OK
    try:
        assert lib.gpk_set_stream(h, C.c_void_p(-1)) == _lib.GPK_OK
        info = (C.c_int * B)()
        noise = np.array([0.1, 0.0])
        assert lib.gpk_fit_batched(h, B, _dp(X), N, D, _dp(Y), _dp(ls), 1, _dp(sf2), _dp(noise), 0.0, 0, info) == _lib.GPK_NOT_PD
        assert info[0] == 0 and 150 < info[1] <= 300
        assert lib.gpk_predict_batched(h, _dp(X), 1, _dp(np.empty(B)), None, 1) == _lib.GPK_BAD_ARG
        noise = np.array([0.1, 0.1])
        assert lib.gpk_fit_batched(h, B, _dp(X), N, D, _dp(Y), _dp(ls), 1, _dp(sf2), _dp(noise), 0.0, 0, info) == _lib.GPK_OK
        thetas = np.array([[0.0, np.log(0.1)], [0.0, -800.0]])        # exp(-800) = 0: singular for model 1
        lml, grad = np.empty(B), np.ones((B, 2))
        assert lib.gpk_lml_batched(h, _dp(thetas), 2, _dp(lml), _dp(grad)) == _lib.GPK_OK
        st = O.fit_fixed(X, Y[:, 0], 1.0, 1.0, 0.1, 0.0, False)
        assert abs(lml[0] - O.log_marginal_likelihood(st)) < 1e-9 * abs(lml[0]) and relerr(grad[0], O.lml_gradient(st)) < 1e-7
        assert lml[1] == -np.inf and not grad[1].any()
    finally:
        lib.gpk_destroy(h)


def test_batched_composite_against_sklearn_goldens(csv_data, ka):
    """KA6 and KA6b (scikit-learn, per-output ARD model of gp_trainer.py:163-174 on the reference's flight CSV, two
    hyper-parameter settings on the same target column) as the two models of one gpk_fit_batched call."""
    from unmanned_aerial_vehicles_amd import _lib
    lib = _lib.load()
    X, Xq = np.ascontiguousarray(csv_data["X10"][:, :9]), np.ascontiguousarray(csv_data["Xq10"][:, :9])
    y = csv_data["Y6"][:, 3]
    N, D, B, M = X.shape[0], 9, 2, Xq.shape[0]
    Y = np.ascontiguousarray(np.stack([y, y], axis=1))
    ls = np.ascontiguousarray(np.stack([np.ones(9), ka["ka6b_ls"]]))
    sf2, noise = np.ones(2), np.array([0.01, 0.05])
    h = C.c_void_p()
    assert lib.gpk_create(C.byref(h), 0) == _lib.GPK_OK
    try:
        assert lib.gpk_set_stream(h, C.c_void_p(-1)) == _lib.GPK_OK
        info = (C.c_int * B)()
        assert lib.gpk_fit_batched(h, B, _dp(X), N, D, _dp(Y), _dp(ls), D, _dp(sf2), _dp(noise), 1e-6, 0, info) == _lib.GPK_OK
        lml = np.empty(B)
        assert lib.gpk_lml_batched(h, None, 0, _dp(lml), None) == _lib.GPK_OK
        assert relerr(lml, np.array([float(ka["ka6_lml"]), float(ka["ka6b_lml"])])) < 1e-10
        mean, var = np.empty((M, B)), np.empty((M, B))
        assert lib.gpk_predict_batched(h, _dp(Xq), M, _dp(mean), _dp(var), 1) == _lib.GPK_OK       # 64 rows: panel route
        assert relerr(mean[:, 0], ka["ka6_mean"]) < 1e-8 and relerr(np.sqrt(var[:, 0]), ka["ka6_std"]) < 1e-8
        assert relerr(mean[:, 1], ka["ka6b_mean"]) < 1e-8 and relerr(np.sqrt(var[:, 1]), ka["ka6b_std"]) < 1e-8
        m25, v25 = np.empty((25, B)), np.empty((25, B))
        assert lib.gpk_predict_batched(h, _dp(Xq), 25, _dp(m25), _dp(v25), 1) == _lib.GPK_OK       # one-call route
        assert relerr(m25[:, 1], ka["ka6b_mean"][:25]) < 1e-8 and relerr(np.sqrt(v25[:, 0]), ka["ka6_std"][:25]) < 1e-8
        thetas = np.ascontiguousarray(np.log(np.c_[ls, noise]))
        grad = np.empty((B, D + 1))
        assert lib.gpk_lml_batched(h, _dp(thetas), D + 1, _dp(lml), _dp(grad)) == _lib.GPK_OK
        assert relerr(lml, np.array([float(ka["ka6_lml"]), float(ka["ka6b_lml"])])) < 1e-10
        assert relerr(grad[0], ka["ka6_grad"]) < 1e-7 and relerr(grad[1], ka["ka6b_grad"]) < 1e-7
    finally:
        lib.gpk_destroy(h)
