"""CPU oracle for the GP residual-model path.  TEST INFRASTRUCTURE ONLY.

This file is a NumPy/SciPy restatement of the arithmetic the reference runs on the CPU
for its Gaussian-Process residual model.  It exists so that the HIP kernels can be
checked against something independent; it is *never* imported by the product package
(`unmanned_aerial_vehicles_amd/`).  Only `tests/`, `__graft_entry__.smoke()` and the
`cpu_baseline` leg of `bench.py` may import it.

Where the algorithm lives
-------------------------
The flight-tested reference path (`src/px4/simple_gp.py:156-201`) delegates all GP
arithmetic to a third-party dependency that is not vendored under the reference tree:
scikit-learn (unpinned in the reference's `README.md:136,141`; pinned by this project
to scikit-learn 1.7.2 / scipy 1.15.3 / numpy 2.2.6).  The functions below restate the
published algorithm of `sklearn.gaussian_process.GaussianProcessRegressor`
(Rasmussen & Williams, Alg. 2.1) and cite the sklearn lines they follow as
`sklearn/...`.  The second, self-contained reference GP
(`quadrotor_gp_mpc/quadrotor_gp_mpc/gaussian_process.py`) is restated in
`PackageGPOracle`.

Parity pinning
--------------
The reference holds no unit tests or golden vectors for this path, so
the oracle is pinned against outputs of the reference itself and of scikit-learn run in
the build container: `tests/golden/make_golden.py` imports the reference modules from
`/root/reference` and sklearn, and freezes inputs/outputs as `tests/golden/*.npz`;
`tests/test_oracle_golden.py` checks every function here against those fixtures.
"""
from __future__ import annotations

import numpy as np
from scipy.linalg import cho_solve, cholesky, solve_triangular

LOG_2PI = float(np.log(2.0 * np.pi))


# --------------------------------------------------------------------------------------
# R1 / R4: RBF kernel matrices by exact differences
# --------------------------------------------------------------------------------------
def _as_ls(length_scale, D):
    ls = np.asarray(length_scale, dtype=np.float64).reshape(-1)
    if ls.size == 1:
        ls = np.full(D, float(ls[0]))
    if ls.size != D:
        raise ValueError(f"length_scale has {ls.size} entries, X has {D} features")
    return ls


def sqdist(XA, XB, length_scale):
    """Scaled squared Euclidean distance by exact differences.

    Follows `sklearn/gaussian_process/kernels.py:1556` (`pdist(X / length_scale,
    'sqeuclidean')`) and `:1564` (`cdist(X / ls, Y / ls, 'sqeuclidean')`): the inputs are
    *divided* by the length-scale first, then differenced, squared and summed over the
    feature axis in feature order.
    """
    XA = np.asarray(XA, dtype=np.float64)
    XB = np.asarray(XB, dtype=np.float64)
    ls = _as_ls(length_scale, XA.shape[1])
    A = XA / ls
    B = XB / ls
    out = np.zeros((A.shape[0], B.shape[0]))
    for d in range(A.shape[1]):
        diff = A[:, d][:, None] - B[:, d][None, :]
        out += diff * diff
    return out


def rbf_gram(X, length_scale, signal_variance=1.0, diag_add=0.0):
    """Training Gram matrix K = sf2 * exp(-0.5 d2) (+ diag_add * I).

    `sklearn/gaussian_process/kernels.py:1553-1560` (RBF), `:1402` (WhiteKernel adds
    `noise_level * I` when Y is None), `sklearn/gaussian_process/_gpr.py:347`
    (`K[diag] += alpha`), and `quadrotor_gp_mpc/quadrotor_gp_mpc/gaussian_process.py:158-171`.
    The diagonal of the RBF part is exactly sf2 (`squareform` + `fill_diagonal(K, 1)`,
    `kernels.py:1559-1560`).
    """
    K = float(signal_variance) * np.exp(-0.5 * sqdist(X, X, length_scale))
    np.fill_diagonal(K, float(signal_variance))
    if diag_add:
        K[np.diag_indices_from(K)] += diag_add
    return K


def rbf_cross(Xq, X, length_scale, signal_variance=1.0):
    """Cross kernel K*[m, j] = sf2 * exp(-0.5 ||(xq_m - x_j)/ls||^2); no white noise
    (`sklearn/gaussian_process/kernels.py:1564-1565`, WhiteKernel cross term is zero `:1414`)."""
    return float(signal_variance) * np.exp(-0.5 * sqdist(Xq, X, length_scale))


# --------------------------------------------------------------------------------------
# R2 / R3 / R6: fit at fixed hyper-parameters
# --------------------------------------------------------------------------------------
class FitState:
    """What `GaussianProcessRegressor.fit` leaves behind (`sklearn/_gpr.py:343-364`)."""

    def __init__(self, X, L, alpha, y_mean, y_std, length_scale, signal_variance, noise, jitter):
        self.X = X
        self.L = L
        self.alpha = alpha
        self.y_mean = y_mean
        self.y_std = y_std
        self.length_scale = length_scale
        self.signal_variance = signal_variance
        self.noise = noise
        self.jitter = jitter


def normalize_targets(Y, normalize_y=True):
    """`sklearn/_gpr.py:271-282`: population std (ddof=0); zero std -> 1
    (`_handle_zeros_in_scale`, which also treats std < 10*eps as zero)."""
    Y = np.asarray(Y, dtype=np.float64)
    if Y.ndim == 1:
        Y = Y[:, None]
    if not normalize_y:
        return Y.copy(), np.zeros(Y.shape[1]), np.ones(Y.shape[1])
    mean = np.mean(Y, axis=0)
    std = np.std(Y, axis=0)
    std = np.where(std < 10 * np.finfo(std.dtype).eps, 1.0, std)
    return (Y - mean) / std, mean, std


def fit_fixed(X, Y, length_scale, signal_variance=1.0, noise=0.0, jitter=0.0, normalize_y=True):
    """Gram + Cholesky + alpha at fixed hyper-parameters (`sklearn/_gpr.py:343-364`).

    `noise` is the WhiteKernel level (in the kernel's diag), `jitter` is the regressor's
    `alpha` added on top of it (`_gpr.py:347`).  Raises `numpy.linalg.LinAlgError` if
    K is not positive definite (`_gpr.py:350-358`).
    """
    X = np.array(X, dtype=np.float64)
    Yn, y_mean, y_std = normalize_targets(Y, normalize_y)
    K = rbf_gram(X, length_scale, signal_variance, diag_add=float(noise) + float(jitter))
    L = cholesky(K, lower=True, check_finite=False)
    alpha = cho_solve((L, True), Yn, check_finite=False)
    st = FitState(X, L, alpha, y_mean, y_std, _as_ls(length_scale, X.shape[1]),
                  float(signal_variance), float(noise), float(jitter))
    st.Yn = Yn
    return st


def log_marginal_likelihood(st: FitState):
    """`sklearn/_gpr.py:609-613`: sum over outputs of
    -1/2 y^T alpha - sum(log diag L) - N/2 log 2 pi."""
    Yn = st.Yn
    per_out = -0.5 * np.einsum("ik,ik->k", Yn, st.alpha)
    per_out -= np.log(np.diag(st.L)).sum()
    per_out -= st.L.shape[0] / 2.0 * LOG_2PI
    return float(per_out.sum())


def lml_gradient(st: FitState, ard=False):
    """Gradient of the LML with respect to log-hyper-parameters
    (`sklearn/_gpr.py:615-647`; kernel gradients `sklearn/kernels.py:1571-1580`,
    WhiteKernel `:1403-1408`).

    theta layout (sklearn order for `[C(fixed) *] RBF + WhiteKernel`):
      iso: [log ls, log noise]; ARD: [log ls_0 .. log ls_{D-1}, log noise].
    d K / d log ls_d   = K_rbf * (x_id - x_jd)^2 / ls_d^2   (iso: summed over d)
    d K / d log noise  = noise * I
    grad_p = 1/2 sum_ij (sum_k alpha_ik alpha_jk - P * Kinv_ij) dK_ij/dtheta_p
    """
    X, L, alpha = st.X, st.L, st.alpha
    N, P = alpha.shape
    Kinv = cho_solve((L, True), np.eye(N), check_finite=False)
    Q = alpha @ alpha.T - P * Kinv
    Krbf = st.signal_variance * np.exp(-0.5 * sqdist(X, X, st.length_scale))
    A = X / st.length_scale
    g_ls = np.zeros(X.shape[1])
    for d in range(X.shape[1]):
        diff = A[:, d][:, None] - A[:, d][None, :]
        g_ls[d] = 0.5 * np.sum(Q * Krbf * diff * diff)
    g_noise = 0.5 * st.noise * np.trace(Q)
    if ard:
        return np.concatenate([g_ls, [g_noise]])
    return np.array([g_ls.sum(), g_noise])


# --------------------------------------------------------------------------------------
# R4 / R5: posterior mean and variance
# --------------------------------------------------------------------------------------
def predict(st: FitState, Xq, return_std=False, diag_includes_noise=True):
    """`sklearn/_gpr.py:441-494`.

    mean = K* alpha * y_std + y_mean; V = L^-1 K*^T;
    var = k(x*,x*) - sum_i V_im^2, clipped at 0, times y_std^2; std = sqrt(var).
    `k(x*,x*)` is `kernel_.diag(X)` = sf2 + noise for `RBF + WhiteKernel`
    (`sklearn/kernels.py:885,1433`); the regressor's jitter `alpha` is NOT included.
    """
    Xq = np.atleast_2d(np.asarray(Xq, dtype=np.float64))
    Ks = rbf_cross(Xq, st.X, st.length_scale, st.signal_variance)
    mean = (Ks @ st.alpha) * st.y_std + st.y_mean
    if not return_std:
        return mean
    V = solve_triangular(st.L, Ks.T, lower=True, check_finite=False)
    kss = st.signal_variance + (st.noise if diag_includes_noise else 0.0)
    var = kss - np.einsum("ij,ij->j", V, V)
    var = np.where(var < 0, 0.0, var)
    var = np.outer(var, st.y_std ** 2)
    return mean, np.sqrt(var)


# --------------------------------------------------------------------------------------
# R9: SimpleQuadrotorGP.predict_residual semantics
# --------------------------------------------------------------------------------------
def predict_residual(st: FitState, state, control):
    """`src/px4/simple_gp.py:187-201`: one row [state(6), control(4)] -> (mean(P,), std^2 (P,))."""
    x = np.concatenate([np.asarray(state, float), np.asarray(control, float)]).reshape(1, -1)
    mean, std = predict(st, x, return_std=True)
    return mean.flatten(), std.flatten() ** 2


def build_gp_residuals(st: FitState, X_guess, U_guess, dt, gain=0.1, n_states=6):
    """`src/px4/mpc.py:1475-1511`: D[3:6, k] = gain * mean_k[3:6] / dt for each horizon point."""
    N = U_guess.shape[1]
    D = np.zeros((n_states, N))
    for k in range(N):
        mean, _ = predict_residual(st, X_guess[:, k], U_guess[:, k])
        if mean.shape[0] >= n_states:
            D[3:6, k] = gain * (mean / dt)[3:6]
    return D


# --------------------------------------------------------------------------------------
# R10: the ROS-package GaussianProcess (self-contained NumPy GP)
# --------------------------------------------------------------------------------------
class PackageGPOracle:
    """Restates `quadrotor_gp_mpc/quadrotor_gp_mpc/gaussian_process.py:19-265`.

    Differences from the sklearn path that must be reproduced:
    * squared distances by the ||x||^2 + ||y||^2 - 2 x.y expansion, floored at 0 (`:38-39`);
    * K = sf2 * exp(-0.5 d2 / ls^2) + noise * I (`:41`, `:166-169`);
    * alpha per output column by cho_solve, *no* target normalisation (`:187-189`);
    * predict variance = diag(K**) - sum(K* . cho_solve(L, K*)), diag(K**) = sf2 without
      noise, floored at 1e-10, tiled over all outputs (`:229-233`);
    * LML = -0.5 (2 sum log diag L + sum_k y_k^T alpha_k + N P log 2 pi) (`:250-261`).
    """

    def __init__(self, length_scale=1.0, signal_variance=1.0, noise_variance=0.01):
        self.length_scale = float(length_scale)
        self.signal_variance = float(signal_variance)
        self.noise_variance = float(noise_variance)
        self.X = None
        self.Y = None
        self.L = None
        self.alpha = None

    def kernel(self, X1, X2):
        d = np.sum(X1 ** 2, 1).reshape(-1, 1) + np.sum(X2 ** 2, 1) - 2 * np.dot(X1, X2.T)
        d = np.maximum(d, 0)
        return self.signal_variance * np.exp(-0.5 * d / self.length_scale ** 2)

    def fit(self, X, Y):
        self.X = np.array(X, dtype=np.float64)
        self.Y = np.array(Y, dtype=np.float64)
        K = self.kernel(self.X, self.X) + self.noise_variance * np.eye(len(self.X))
        self.L = cholesky(K, lower=True)
        self.alpha = np.zeros_like(self.Y)
        for i in range(self.Y.shape[1]):
            self.alpha[:, i] = cho_solve((self.L, True), self.Y[:, i])
        return self

    def predict(self, Xq):
        Xq = np.atleast_2d(np.asarray(Xq, dtype=np.float64))
        Ks = self.kernel(self.X, Xq)
        mean = Ks.T @ self.alpha
        v = cho_solve((self.L, True), Ks)
        var = self.signal_variance - np.sum(Ks * v, axis=0)
        var = np.maximum(var, 1e-10)
        var = np.tile(var.reshape(-1, 1), (1, self.Y.shape[1]))
        return mean, var

    def log_marginal_likelihood(self):
        log_det = 2 * np.sum(np.log(np.diag(self.L)))
        quad = sum(np.dot(self.Y[:, i], self.alpha[:, i]) for i in range(self.Y.shape[1]))
        n = len(self.X)
        return float(-0.5 * (log_det + quad + n * self.Y.shape[1] * np.log(2 * np.pi)))


# --------------------------------------------------------------------------------------
# R12: per-output scaled ARD GPs (gp_trainer.py / pretrained_gp.py)
# --------------------------------------------------------------------------------------
def standard_scale(A):
    """`sklearn.preprocessing.StandardScaler` as used at `src/px4/gp_trainer.py:152-159`:
    population std, zero scale -> 1."""
    A = np.asarray(A, dtype=np.float64)
    mean = A.mean(axis=0)
    scale = A.std(axis=0)
    scale = np.where(scale < 10 * np.finfo(np.float64).eps, 1.0, scale)
    return (A - mean) / scale, mean, scale


# --------------------------------------------------------------------------------------
# synthetic workload of SURVEY.md §8(d) / BASELINE.md §4
# --------------------------------------------------------------------------------------
def synthetic_problem(N, M, D=9, P=3):
    """Deterministic inputs used by the benchmark and the large-size property tests."""
    rng = np.random.default_rng(0)
    X = rng.standard_normal((N, D))
    W = rng.standard_normal((D, P))
    Y = np.sin(X @ W) + 0.1 * rng.standard_normal((N, P))
    Xq = np.random.default_rng(1).standard_normal((M, D))
    return X, Y, Xq


def synthetic_flight_problem(N, D=10, P=6, seed=7):
    """Flight-like training rows for the reference's offline-training workload (src/px4/train_gp_offline.py:124-140 ->
    src/px4/simple_gp.py:156-185: N <= 10 000 rows of [x, y, z, vx, vy, vz, ax, ay, az, yaw_rate] -> 6 residuals).
    Deterministic: positions / velocities / accelerations of unit-ish scale, a numerically silent yaw-rate column (as in
    the flight CSVs), residuals = a smooth drag-like function of velocity and acceleration + sensor noise, of the
    magnitude the CSVs hold (1e-2).  `bench.py --workload train` carries the same generator."""
    rng = np.random.default_rng(seed)
    X = 0.6 * rng.standard_normal((N, D))
    X[:, 2] -= 3.0                              # altitude around -3 m (NED)
    if D >= 10:
        X[:, 9] *= 1e-3                         # yaw rate: almost constant
    W = rng.standard_normal((6, P))
    va = X[:, 3:9]
    Y = 0.03 * np.sin(va @ W) - 0.01 * np.pad(va[:, :3] * np.abs(va[:, :3]), ((0, 0), (0, max(P - 3, 0))))[:, :P]
    Y = Y + 0.005 * rng.standard_normal((N, P))
    return X, Y
