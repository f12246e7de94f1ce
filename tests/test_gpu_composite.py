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
