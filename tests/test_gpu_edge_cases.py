"""GPU edge cases: ragged sizes (N, M not multiples of anything), extreme feature/output counts, empty
query batches, duplicate rows, non-positive-definite matrices and the reference's failure conventions."""
import numpy as np
import pytest

from conftest import relerr
from oracle import gp_oracle as O

pytestmark = pytest.mark.gpu


def _fit(X, Y, ls, noise, alpha=1e-4, normalize_y=True, **kw):
    from unmanned_aerial_vehicles_amd import RBF, GaussianProcessRegressor, WhiteKernel
    return GaussianProcessRegressor(kernel=RBF(ls) + WhiteKernel(noise), alpha=alpha, normalize_y=normalize_y,
                                    optimizer=None, **kw).fit(X, Y)


@pytest.mark.parametrize("N,D,P,M", [(2, 1, 1, 1), (31, 3, 2, 5), (129, 16, 12, 130), (257, 7, 16, 1), (640, 2, 3, 257)])
def test_ragged_shapes_against_oracle(N, D, P, M):
    rng = np.random.default_rng(N * 131 + D)
    X = rng.standard_normal((N, D))
    Y = rng.standard_normal((N, P)) if P > 1 else rng.standard_normal(N)
    Xq = rng.standard_normal((M, D)) * 0.7
    ls = 0.8 + 0.1 * np.arange(D) if D > 1 else 0.8
    g = _fit(X, Y, ls, 0.05)
    st = O.fit_fixed(X, Y, ls, 1.0, 0.05, 1e-4)
    mean, std = g.predict(Xq, return_std=True)
    rm, rs = O.predict(st, Xq, return_std=True)
    if P == 1:
        rm, rs = rm[:, 0], rs[:, 0]
    assert mean.shape == rm.shape and std.shape == rs.shape
    assert relerr(mean, rm) < 1e-9 and relerr(std, rs) < 1e-8
    assert abs(g.log_marginal_likelihood_value_ - O.log_marginal_likelihood(st)) < 1e-10 * abs(O.log_marginal_likelihood(st))
    if D <= 16:
        lml, grad = g.log_marginal_likelihood(g.kernel_.theta, eval_gradient=True)
        assert relerr(grad, O.lml_gradient(st, ard=D > 1)) < 1e-7
    for method in ("solve", "inverse"):
        g.var_method = method
        assert relerr(g.predict(Xq, return_std=True)[1], rs) < 1e-8


def test_empty_query_batch_and_wrong_width():
    rng = np.random.default_rng(0)
    X, Y = rng.standard_normal((50, 4)), rng.standard_normal((50, 2))
    g = _fit(X, Y, 1.0, 0.1)
    mean, std = g.predict(np.empty((0, 4)), return_std=True)
    assert mean.shape == (0, 2) and std.shape == (0, 2)
    with pytest.raises(ValueError):
        g.predict(np.zeros((3, 5)))
    with pytest.raises(ValueError):
        _fit(X, Y[:10], 1.0, 0.1)
    with pytest.raises(ValueError):
        _fit(np.full((5, 2), np.nan), np.zeros(5), 1.0, 0.1)
    # non-finite queries are refused as scikit-learn refuses them (its check_array), not mapped to the prior
    bad = np.zeros((3, 4))
    bad[1, 2] = np.nan
    with pytest.raises(ValueError, match="NaN or infinity"):
        g.predict(bad)
    bad[1, 2] = np.inf
    with pytest.raises(ValueError, match="NaN or infinity"):
        g.predict(bad, return_std=True)
    # more than 16 features: refused when the model is built (every predict / gradient kernel is compiled for D <= 16)
    with pytest.raises(ValueError, match="D must be in"):
        _fit(rng.standard_normal((40, 17)), rng.standard_normal(40), 1.0, 0.1)


def test_duplicates_and_not_positive_definite():
    from unmanned_aerial_vehicles_amd import RBF, GaussianProcess, GaussianProcessRegressor
    rng = np.random.default_rng(1)
    X = rng.standard_normal((100, 3))
    X[50:] = X[:50]                                  # exact duplicates: K is singular without noise
    y = rng.standard_normal(100)
    g = _fit(X, y, 1.0, 0.1)                         # noise regularises it
    st = O.fit_fixed(X, y, 1.0, 1.0, 0.1, 1e-4)
    assert relerr(g.predict(X[:7]), O.predict(st, X[:7])[:, 0]) < 1e-8
    # sklearn convention: a not-PD final fit raises numpy.linalg.LinAlgError (_gpr.py:350-358) ...
    with pytest.raises(np.linalg.LinAlgError, match="not returning a positive definite"):
        GaussianProcessRegressor(kernel=RBF(1.0), alpha=0.0, optimizer=None).fit(X, y)
    # package convention: fit() inflates the noise tenfold and leaves alpha = 0, L = None (gaussian_process.py:193-201)
    p = GaussianProcess(input_dim=3, output_dim=1)
    p.noise_variance = 0.0
    p.add_training_data(X, y.reshape(-1, 1))
    p.noise_variance = -1e-3                          # forces a non-positive pivot
    p.fit()
    assert p.L is None and not p.alpha.any() and np.isclose(p.noise_variance, -1e-2)
    m, v = p.predict(X[:3])
    assert not m.any() and np.allclose(v, p.kernel.signal_variance)
    assert p.log_marginal_likelihood() == -np.inf


def test_lml_minus_inf_inside_optimiser():
    """A theta that makes K numerically singular must give (-inf, 0) instead of raising (_gpr.py:588-589),
    and must leave the fitted model usable."""
    from unmanned_aerial_vehicles_amd import RBF, GaussianProcessRegressor, WhiteKernel
    rng = np.random.default_rng(2)
    X = np.repeat(rng.standard_normal((20, 2)), 2, axis=0)       # exact duplicate rows
    y = rng.standard_normal(40)
    g = GaussianProcessRegressor(kernel=RBF(1.0) + WhiteKernel(0.1), alpha=0.0, optimizer=None).fit(X, y)
    before = g.predict(X[:5])
    lml, grad = g.log_marginal_likelihood(np.log([1.0, 1e-300]), eval_gradient=True)
    assert lml == -np.inf and grad.shape == (2,) and not grad.any()
    assert g.log_marginal_likelihood(np.log([1.0, 1e-300])) == -np.inf
    assert np.array_equal(g.predict(X[:5]), before)


def test_prior_prediction_unfitted():
    from unmanned_aerial_vehicles_amd import RBF, GaussianProcessRegressor, WhiteKernel
    g = GaussianProcessRegressor(kernel=RBF(1.0) + WhiteKernel(0.5))
    mean, std = g.predict(np.zeros((4, 3)), return_std=True)
    assert mean.shape == (4,) and not mean.any() and np.allclose(std, np.sqrt(1.5))


def test_simple_gp_swallows_failures(csv_data):
    """The model seam never raises into the control loop (simple_gp.py:199-201)."""
    from unmanned_aerial_vehicles_amd import SimpleQuadrotorGP
    gp = SimpleQuadrotorGP()
    gp.gp_model = _fit(csv_data["X10"][:200], csv_data["Y6"][:200], 0.5, 0.1)
    gp.is_trained = True
    m, v = gp.predict_residual(np.zeros(5), np.zeros(4))          # wrong width -> fallback, no exception
    assert not m.any() and (v == 1).all()
    assert gp.load_model("/nonexistent/model.pkl") is False


def test_predict_during_refit_threads(csv_data):
    """The package GP is refitted on a timer thread while a subscription thread predicts
    (quadrotor_gp_mpc/quadrotor_gp_mpc/main.py:818-826, unlocked in the reference).  Here fit() builds a new
    device model and swaps it in atomically: concurrent predicts never fail and always come from one
    consistent model (old or new)."""
    import threading
    from unmanned_aerial_vehicles_amd import GaussianProcess
    X, Y = csv_data["X10"][:400, :9], csv_data["Y6"][:400, 3:6]
    g = GaussianProcess(input_dim=9, output_dim=3)
    g.add_training_data(X[:200], Y[:200])
    g.fit()
    xq = X[5:6]
    m_old = g.predict(xq)[0].copy()
    g2 = GaussianProcess(input_dim=9, output_dim=3)
    g2.add_training_data(X, Y)
    g2.fit()
    m_new = g2.predict(xq)[0].copy()
    errors, seen = [], set()
    stop = threading.Event()

    def predictor():
        while not stop.is_set():
            try:
                m, v = g.predict(xq)
                if np.allclose(m, m_old, rtol=1e-12, atol=1e-15):
                    seen.add("old")
                elif np.allclose(m, m_new, rtol=1e-12, atol=1e-15):
                    seen.add("new")
                else:
                    errors.append(("mixed", m.copy()))
                if not (np.isfinite(v).all() and (v > 0).all()):
                    errors.append(("var", v.copy()))
            except Exception as e:  # noqa: BLE001
                errors.append(("exc", repr(e)))

    t = threading.Thread(target=predictor)
    t.start()
    try:
        for _ in range(5):
            g.X_train, g.Y_train = X[:200], Y[:200]
            g.fit()
            g.X_train, g.Y_train = X, Y
            g.fit()
    finally:
        stop.set()
        t.join(30)
    assert not errors, errors[:3]
    assert "new" in seen or "old" in seen


@pytest.mark.parametrize("N,M", [(4961, 300), (8900, 1100)])
def test_no_reads_of_unwritten_memory(N, M, monkeypatch):
    """Every fresh device buffer is poisoned with NaN (GPK_DEBUG_FILL): a kernel that reads memory nobody wrote --
    the part of a triangular operand beyond its zero band, an unwritten scratch tile -- turns the result into NaN
    instead of passing by the luck of a zero-filled allocation.  Ragged tile counts (39 and 70 tiles: groups of the
    tile walk run across bands and the last band is short) through factorisation, inverse factor, both alpha
    solves and every variance path."""
    from unmanned_aerial_vehicles_amd.device import DeviceGP, get_backend
    monkeypatch.setenv("GPK_DEBUG_FILL", "nan")
    rng = np.random.default_rng(N)
    X = rng.standard_normal((N, 7)); Y = np.sin(X @ rng.standard_normal((7, 2))) + 0.1 * rng.standard_normal((N, 2))
    Y = (Y - Y.mean(0)) / Y.std(0)
    Xq = rng.standard_normal((M, 7))
    dev = DeviceGP(X, Y, get_backend(0))
    dev.factorize(2.0, 1.0, 0.03)
    dev.solve_alpha("chain")
    a_chain = dev.alpha_host()
    dev.solve_alpha("inverse")
    a_inv = dev.alpha_host()
    assert np.all(np.isfinite(a_chain)) and relerr(a_inv, a_chain) < 1e-9
    logdet, quad = dev.lml_terms()
    assert np.isfinite(logdet) and np.all(np.isfinite(quad))
    v_solve = dev.predict_var_dev(Xq, 1.03, 0.0, "float64", "solve").cpu().numpy()
    v_inv = dev.predict_var_dev(Xq, 1.03, 0.0, "float64", "inverse").cpu().numpy()
    v32 = dev.predict_var_dev(Xq, 1.03, 0.0, "float32", "inverse").cpu().numpy()
    vsp = dev.predict_var_dev(Xq, 1.03, 0.0, "float32", "inverse_split").cpu().numpy()
    assert np.all(np.isfinite(v_solve)) and np.max(np.abs(v_inv - v_solve)) < 1e-10
    assert np.max(np.abs(np.sqrt(v32) - np.sqrt(v_solve)) / np.sqrt(v_solve)) < 1e-3
    assert np.max(np.abs(np.sqrt(vsp) - np.sqrt(v_solve)) / np.sqrt(v_solve)) < 1e-3
    g = dev.lml_grad(0.03)
    assert np.all(np.isfinite(g))


def test_fast_paths_match_their_plain_forms(monkeypatch):
    """The round's shortcuts against the forms they replace, on a second handle with the switches off (the switches
    are read when a handle is created): fused 256-wide solve base vs three launches, level-by-level inverse vs the
    recursion, 64-tile launches vs 128-tile ones, small-batch serving kernels vs the general chain."""
    from unmanned_aerial_vehicles_amd.device import Backend, DeviceGP, get_backend
    fast = get_backend(0)
    plain = Backend(0).set_options(trsm256=0, trtri_levels=0, gemm_small_tiles=128, small_path=0)
    rng = np.random.default_rng(5)
    for N in (1000, 2048, 4096, 4961):
        X = rng.standard_normal((N, 8)); Y = np.sin(X @ rng.standard_normal((8, 3))) + 0.1 * rng.standard_normal((N, 3))
        Y = (Y - Y.mean(0)) / Y.std(0)
        out = []
        for be in (fast, plain):
            dev = DeviceGP(X, Y, be)
            dev.factorize(1.8, 1.0, 0.05)
            L = np.tril(dev.K.cpu().numpy())
            W = np.tril(dev.inverse_factor(False).cpu().numpy())
            dev.solve_alpha()
            q = rng.standard_normal((25, 8)) if not out else out[0][3]
            mean, var = dev.predict_host(q, np.zeros(3), np.ones(3), 1.05, 0.0)
            out.append((L, W, dev.alpha_host(), q, mean, var))
        (L1, W1, a1, _, m1, v1), (L2, W2, a2, _, m2, v2) = out
        assert relerr(L1, L2) < 1e-13 and relerr(W1, W2) < 1e-11 and relerr(a1, a2) < 1e-10, N
        assert relerr(m1, m2) < 1e-11 and relerr(v1, v2) < 1e-9, N


def test_small_launch_build_gives_the_same_evaluation():
    """The launches of up to 36 tile columns run a second, 256-register build of the one-launch factorisation (two k-tiles in
    flight in the off-diagonal k-loops, the tiles of the inverse factor included): one LML + gradient evaluation with it and
    with the 128-register build (ptile_sr = 0) - every task does the same arithmetic in the same order, so the terms, the
    gradient and alpha are bit-identical."""
    from unmanned_aerial_vehicles_amd.device import DeviceGP, get_backend
    be = get_backend(0)
    rng = np.random.default_rng(11)
    for N in (1000, 3000):
        X = rng.standard_normal((N, 7)); Y = np.sin(X @ rng.standard_normal((7, 2))) + 0.1 * rng.standard_normal((N, 2))
        Y = (Y - Y.mean(0)) / Y.std(0)
        out = []
        try:
            for sr in (1, 0):
                be.set_options(ptile_sr=sr)
                dev = DeviceGP(X, Y, be)
                ld, quad, g = dev.lml_eval(1.3, 1.0, 0.05 + 1e-4, 0.05, True)
                out.append((ld, np.array(quad), np.array(g), dev.alpha_host()))
        finally:
            be.set_options(ptile_sr=1)
        (ld1, q1, g1, a1), (ld0, q0, g0, a0) = out
        assert ld1 == ld0 and np.array_equal(q1, q0) and np.array_equal(g1, g0) and np.array_equal(a1, a0), N

