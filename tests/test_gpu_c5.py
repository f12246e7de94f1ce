"""BASELINE configs[4] (C5) at SURVEY.md section 8(d)'s sizes: three per-axis ARD GPs on shared inputs, log-marginal
likelihood AND its gradient (the hyper-parameter step of src/px4/gp_trainer.py:163-179).

  N = 4096   against scikit-learn itself (tests/golden/c5_ref.npz, written by make_golden_r4.py from
             sklearn/gaussian_process/_gpr.py:537-652 + kernels.py:1571-1580): through `BatchedARDGP` (fused launch chain and
             per-model chains) and through the C ABI (`gpk_fit_batched` + `gpk_lml_batched`), LML 1e-10 / gradient 1e-8.
  N = 16384  no oracle can hold the (N, N, D) tensor scikit-learn materialises: the analytic gradient of the fused evaluation
             against a fourth-order central finite difference of its own LML along 3 random directions, 1e-6."""
import ctypes as C
import os

import numpy as np
import pytest

from conftest import relerr
from oracle import gp_oracle as O

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def _dp(a):
    from unmanned_aerial_vehicles_amd import _lib
    return a.ctypes.data_as(_lib._dp)


@pytest.fixture(scope="module")
def c5():
    return dict(np.load(os.path.join(HERE, "golden", "c5_ref.npz")))


def test_c5_lml_and_gradient_vs_sklearn_at_n4096(c5):
    from unmanned_aerial_vehicles_amd import BatchedARDGP
    N = int(c5["N"])
    X, Y, _ = O.synthetic_problem(N, 1)
    ls, noise, jitter = c5["length_scale"], float(c5["noise_level"]), float(c5["alpha"])
    thetas = np.ascontiguousarray(c5["theta"])                       # (3, 10): log l_0..l_8, log noise (the constant is fixed)
    assert relerr(thetas, np.log(np.r_[ls, noise])[None, :].repeat(3, 0)) < 1e-14
    bg = BatchedARDGP(length_scale=ls, noise_level=noise, alpha=jitter, normalize_y=True, optimizer=None).fit(X, Y)
    gscale = np.abs(c5["grad"]).max(axis=1, keepdims=True)
    for fused in (True, False):
        lml, grad = bg.log_marginal_likelihood(thetas, eval_gradient=True, fused=fused)
        assert relerr(lml, c5["lml"]) < 1e-10, fused
        assert np.max(np.abs(grad - c5["grad"]) / gscale) < 1e-8, fused
    bg.release_fused_buffers()
    del bg

    # the same through the C ABI: a caller without Python (include/gpk.h: gpk_fit_batched, gpk_lml_batched)
    from unmanned_aerial_vehicles_amd import _lib
    lib = _lib.load()
    h = C.c_void_p()
    assert lib.gpk_create(C.byref(h), 0) == _lib.GPK_OK
    try:
        assert lib.gpk_set_stream(h, C.c_void_p(-1)) == _lib.GPK_OK
        B = 3
        info = (C.c_int * B)()
        lsB = np.ascontiguousarray(np.tile(ls, (B, 1)))
        Xc, Yc = np.ascontiguousarray(X), np.ascontiguousarray(Y)
        rc = lib.gpk_fit_batched(h, B, _dp(Xc), N, 9, _dp(Yc), _dp(lsB), 9, _dp(np.ones(B)), _dp(np.full(B, noise)),
                                 jitter, 1, info)
        assert rc == _lib.GPK_OK, lib.gpk_last_error(h)
        lml, grad = np.empty(B), np.empty((B, 10))
        assert lib.gpk_lml_batched(h, _dp(thetas), 10, _dp(lml), _dp(grad)) == _lib.GPK_OK, lib.gpk_last_error(h)
        assert relerr(lml, c5["lml"]) < 1e-10
        assert np.max(np.abs(grad - c5["grad"]) / gscale) < 1e-8
    finally:
        lib.gpk_destroy(h)


def test_c5_gradient_finite_difference_at_n16384():
    from unmanned_aerial_vehicles_amd import BatchedARDGP
    N = 16384
    X, Y, _ = O.synthetic_problem(N, 1)
    ls = 2.0 * (1.0 + 0.1 * np.arange(9))
    bg = BatchedARDGP(length_scale=ls, noise_level=0.1, alpha=1e-4, normalize_y=True, optimizer=None).fit(X, Y)
    th0 = np.array(bg.thetas, dtype=np.float64)
    lml0, grad = bg.log_marginal_likelihood(th0, eval_gradient=True, fused=True)
    assert np.isfinite(lml0).all() and np.isfinite(grad).all()
    rng = np.random.default_rng(16384)
    h = 2e-3
    for _ in range(3):
        d = rng.standard_normal(th0.shape)
        d /= np.linalg.norm(d, axis=1, keepdims=True)                 # one unit direction per model (the LMLs are separable)
        f = {k: bg.log_marginal_likelihood(th0 + k * h * d, eval_gradient=False, fused=True) for k in (-2, -1, 1, 2)}
        fd = (-f[2] + 8.0 * f[1] - 8.0 * f[-1] + f[-2]) / (12.0 * h)
        an = np.einsum("bk,bk->b", grad, d)
        assert np.max(np.abs(fd - an) / np.linalg.norm(grad, axis=1)) < 1e-6, (fd, an)
    bg.release_fused_buffers()
