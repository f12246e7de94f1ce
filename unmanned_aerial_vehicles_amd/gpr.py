"""`GaussianProcessRegressor` with the estimator seam the reference calls, computed on MI355X.

Drop-in for the object the reference stores under `'gp_model'` (`src/px4/simple_gp.py:170-177`,
`src/px4/train_gp_offline.py:188-194`): same constructor arguments, `fit(X, y)`,
`predict(X, return_std=False)`, `log_marginal_likelihood(theta, eval_gradient)`, attributes
`kernel_`, `X_train_`, `y_train_`, `alpha_`, `L_`, `n_features_in_`,
`log_marginal_likelihood_value_`, and it pickles.  The arithmetic follows scikit-learn 1.7.2's
`GaussianProcessRegressor` (Rasmussen & Williams Alg. 2.1; `sklearn/gaussian_process/_gpr.py`)
but every O(N^2)/O(N^3) step runs in the HIP kernels behind libgpk:

    K1 gram -> K2 potrf -> K3 potrs -> K6 lml / gradient      (fit, per optimiser evaluation)
    K4 fused mean, K5 variance (trsm + column norms)           (predict)

There is no CPU fallback: without the GPU library `fit`/`predict` raise.
"""
from __future__ import annotations

import copy
import warnings

import numpy as np
import scipy.optimize

from .device import DeviceGP, get_backend
from .kernels import RBF, ConstantKernel, Kernel, KernelComponents, WhiteKernel  # noqa: F401
from ._lib import NotPositiveDefinite

LOG_2PI = float(np.log(2.0 * np.pi))


def _rng_from(random_state):
    """`sklearn.utils.check_random_state` semantics (`sklearn/gaussian_process/_gpr.py:249`):
    None -> numpy's global RandomState (so `np.random.seed(0)` pins the restarts, as in the
    reference's trainer), int -> fresh RandomState, RandomState -> itself."""
    if random_state is None or random_state is np.random:
        return np.random.mtrand._rand
    if isinstance(random_state, (int, np.integer)):
        return np.random.RandomState(int(random_state))
    return random_state


class GaussianProcessRegressor:
    def __init__(self, kernel=None, *, alpha=1e-10, optimizer="fmin_l_bfgs_b", n_restarts_optimizer=0,
                 normalize_y=False, copy_X_train=True, n_targets=None, random_state=None, device=None,
                 predict_dtype="float64", var_method="auto"):
        self.kernel = kernel
        self.alpha = alpha
        self.optimizer = optimizer
        self.n_restarts_optimizer = n_restarts_optimizer
        self.normalize_y = normalize_y
        self.copy_X_train = copy_X_train
        self.n_targets = n_targets
        self.random_state = random_state
        self.device = device
        self.predict_dtype = predict_dtype
        self.var_method = var_method      # 'auto' | 'inverse' | 'solve' | 'inverse_split' | 'inverse_split2' (fp32 only): DeviceGP.predict_var_dev
        self.fp32_gate = True             # predict_dtype="float32": route ill-conditioned models / tiny variances to fp64
        self._dev = None

    # ------------------------------------------------------------------ fit
    def fit(self, X, y):
        """`sklearn/gaussian_process/_gpr.py:225-365`."""
        if self.kernel is None:  # sklearn default: C(1.0, fixed) * RBF(1.0, fixed)
            self.kernel_ = ConstantKernel(1.0, constant_value_bounds="fixed") * RBF(1.0, length_scale_bounds="fixed")
        else:
            self.kernel_ = copy.deepcopy(self.kernel)
        self._rng = _rng_from(self.random_state)
        X = np.array(X, dtype=np.float64, ndmin=2)
        y = np.asarray(y, dtype=np.float64)
        if X.shape[0] != y.shape[0]:
            raise ValueError("X and y have inconsistent numbers of samples")
        if not (np.isfinite(X).all() and np.isfinite(y).all()):
            raise ValueError("Input contains NaN or infinity")
        if np.iterable(self.alpha):
            raise ValueError("per-sample alpha is not supported by the MI355X path")
        self._y_1d = y.ndim == 1
        y2 = y.reshape(X.shape[0], -1)
        if self.n_targets is not None and y2.shape[1] != self.n_targets:
            raise ValueError("The number of targets seen in `y` is different from the parameter `n_targets`.")
        # normalise targets: population std, zero std -> 1   (_gpr.py:271-282)
        if self.normalize_y:
            self._y_train_mean = np.mean(y2, axis=0)
            std = np.std(y2, axis=0)
            self._y_train_std = np.where(std < 10 * np.finfo(np.float64).eps, 1.0, std)
            yn = (y2 - self._y_train_mean) / self._y_train_std
        else:
            self._y_train_mean = np.zeros(y2.shape[1])
            self._y_train_std = np.ones(y2.shape[1])
            yn = y2.copy()
        self.X_train_ = X.copy() if self.copy_X_train else X
        self._yn = yn
        self.y_train_ = yn[:, 0].copy() if self._y_1d else yn
        self.n_features_in_ = X.shape[1]
        self._comp_check()
        self._dev = DeviceGP(self.X_train_, yn, get_backend(self.device))

        if self.optimizer is not None and self.kernel_.n_dims > 0:
            def obj(theta):
                lml, grad = self._lml_on_device(theta, eval_gradient=True)
                return -lml, -grad

            bounds = self.kernel_.bounds
            starts = [self.kernel_.theta]
            if self.n_restarts_optimizer > 0:
                if not np.isfinite(bounds).all():
                    raise ValueError("Multiple optimizer restarts (n_restarts_optimizer>0) requires that all "
                                     "bounds are finite.")
                # (scikit-learn draws a restart's start after the previous run, _gpr.py:316-327 - from a generator the runs do
                # not touch: drawing them up front gives the same starts)
                starts += [self._rng.uniform(bounds[:, 0], bounds[:, 1]) for _ in range(self.n_restarts_optimizer)]
            optima = self._optimise_from(starts, bounds, obj, yn)
            vals = [o[1] for o in optima]
            self.kernel_.theta = optima[int(np.argmin(vals))][0]
            self.log_marginal_likelihood_value_ = -float(np.min(vals))
            self._dev.release_grad_buffers()
            lml_from_final_factor = False
        else:
            # fixed theta: the reference evaluates the LML (one Cholesky) and then factorises again for
            # prediction (_gpr.py:339-349); here the one factorisation below serves both
            lml_from_final_factor = True

        # final factorisation at the selected theta (_gpr.py:343-364); not-PD raises here
        dev = self._dev
        try:
            logdet_half, quad = self._refactor()
        except NotPositiveDefinite as exc:
            exc.args = (f"The kernel, {self.kernel_}, is not returning a positive definite matrix. Try gradually "
                        "increasing the 'alpha' parameter of your GaussianProcessRegressor estimator.",) + exc.args
            raise
        if lml_from_final_factor:
            self.log_marginal_likelihood_value_ = float(
                np.sum(-0.5 * quad - logdet_half - 0.5 * dev.N * LOG_2PI))
        self._alpha_host = None
        self._L_host = None
        return self

    # the restarts run side by side while all their device models together stay below this many bytes ...
    CONCURRENT_RESTART_BYTES = 48e9
    # ... and while one evaluation leaves most of the chip idle: from ~8000 rows on it fills it and two chains side by side only
    # take each other's matrix pipe (fit with one restart, side by side / one after the other: N = 6000 244 / 255 ms,
    # 8000 562 / 555, 10 000 1091 / 1058: profiles/r05_train_restarts_ab.log)
    CONCURRENT_RESTART_MAX_NP = 7168

    def _optimise_from(self, starts, bounds, obj, yn):
        """One optimiser run per start, results in the order of `starts`.  The runs are independent (same data, their own
        start): with more than one they run side by side, each on its own libgpk handle, stream and scratch model from its own
        host thread - at the reference's sizes an evaluation is a chain of latency-bound launches that leaves most of the
        chip idle, and two chains interleave.  Same results as one after the other (every run is deterministic)."""
        n = len(starts)
        per_run = 3.3 * self._dev.Np * self._dev.Np * 8.0
        if (n == 1 or not getattr(self, "concurrent_restarts", True) or n * per_run > self.CONCURRENT_RESTART_BYTES
                or self._dev.Np > self.CONCURRENT_RESTART_MAX_NP):
            return [self._constrained_optimization(obj, t0, bounds) for t0 in starts]
        import torch
        from concurrent.futures import ThreadPoolExecutor
        from .device import worker_backends
        workers = worker_backends(self._dev.be.device_index, n)
        torch.cuda.current_stream(self._dev.be.device).synchronize()

        def run(i):
            be, stream = workers[i]
            with torch.cuda.stream(stream):
                dev = DeviceGP(self.X_train_, yn, be)

                def obj_i(theta):
                    lml, grad = self._lml_on_device(theta, True, dev)
                    return -lml, -grad

                out = self._constrained_optimization(obj_i, starts[i], bounds)
                stream.synchronize()
            return out

        with ThreadPoolExecutor(max_workers=n) as ex:
            return list(ex.map(run, range(n)))

    def _comp_check(self):
        comp = self.kernel_.components()
        comp.ls_vector(self.n_features_in_)
        return comp

    def _constrained_optimization(self, obj, theta0, bounds):
        """`sklearn/gaussian_process/_gpr.py:654-670`."""
        if self.optimizer == "fmin_l_bfgs_b":
            res = scipy.optimize.minimize(obj, theta0, method="L-BFGS-B", jac=True, bounds=bounds)
            if res.status != 0 and "CONVERGENCE" not in str(res.message):
                warnings.warn(f"lbfgs failed to converge (status={res.status}): {res.message}")
            return res.x, res.fun
        if callable(self.optimizer):
            return self.optimizer(obj, theta0, bounds=bounds)
        raise ValueError(f"Unknown optimizer {self.optimizer}.")

    # ------------------------------------------------------------------ LML
    def _lml_on_device(self, theta, eval_gradient, dev=None):
        """One evaluation of `_gpr.py:537-652` on the GPU: K1, K2, K3, K6a (+ K^-1 and K6b).  Works on `dev`
        (default: the estimator's own device model, whose factor it overwrites - what `fit` wants)."""
        dev = dev if dev is not None else self._dev
        kern = self.kernel_.clone_with_theta(theta)
        comp = kern.components()
        D = self.n_features_in_
        N = dev.N
        if dev.Np <= dev.INVERSE_EAGER_NP and D <= 16:
            # the whole evaluation as one chain of launches with one synchronisation (gpk_lml_eval)
            try:
                logdet_half, quad, g = dev.lml_eval(comp.ls_vector(D), comp.sf2, (comp.noise or 0.0) + float(self.alpha),
                                                    comp.noise or 0.0, eval_gradient)
            except NotPositiveDefinite:
                return (-np.inf, np.zeros_like(theta)) if eval_gradient else -np.inf
            lml = float(np.sum(-0.5 * quad - logdet_half - 0.5 * N * LOG_2PI))
            return (lml, comp.map_gradient(g, D)) if eval_gradient else lml
        try:
            dev.factorize(comp.ls_vector(D), comp.sf2, (comp.noise or 0.0) + float(self.alpha))
        except NotPositiveDefinite:
            return (-np.inf, np.zeros_like(theta)) if eval_gradient else -np.inf
        dev.solve_alpha()
        logdet_half, quad = dev.lml_terms()
        lml = float(np.sum(-0.5 * quad - logdet_half - 0.5 * N * LOG_2PI))
        if not eval_gradient:
            return lml
        g = dev.lml_grad(comp.noise or 0.0)
        return lml, comp.map_gradient(g, D)

    def log_marginal_likelihood(self, theta=None, eval_gradient=False, clone_kernel=True):
        if theta is None:
            if eval_gradient:
                raise ValueError("Gradient can only be evaluated for theta!=None")
            return self.log_marginal_likelihood_value_
        if not hasattr(self, "X_train_"):
            raise RuntimeError("This GaussianProcessRegressor instance is not fitted yet.")
        theta = np.asarray(theta, dtype=np.float64)
        # Non-mutating, as in scikit-learn (_gpr.py:537-652 works on fresh arrays): the trial theta is factorised
        # in a scratch device model, so the fitted factor in HBM - which a predict() on another thread may be
        # reading - is never touched.  The scratch (a second N x N matrix) is kept for the next evaluation;
        # `release_lml_scratch()` frees it.
        scratch = getattr(self, "_lml_dev", None)
        if scratch is None or scratch.N != self.X_train_.shape[0]:
            scratch = self._lml_dev = DeviceGP(self.X_train_, self._yn, get_backend(self.device))
        out = self._lml_on_device(theta, eval_gradient, scratch)
        if not clone_kernel:
            self.kernel_.theta = theta
        return out

    def release_lml_scratch(self):
        self._lml_dev = None

    # ------------------------------------------------------------------ predict
    def predict(self, X, return_std=False, return_cov=False):
        """`sklearn/gaussian_process/_gpr.py:367-496`."""
        if return_cov:
            raise NotImplementedError("return_cov is not part of the reference's GP-MPC path")
        X = np.array(X, dtype=np.float64, ndmin=2)
        if not np.isfinite(X).all():       # scikit-learn's input validation (_gpr.py:404-412 -> check_array)
            raise ValueError("Input X contains NaN or infinity.")
        if not hasattr(self, "X_train_"):  # prior (unfitted) prediction, _gpr.py:416-440
            n_t = self.n_targets if self.n_targets is not None else 1
            kern = self.kernel if self.kernel is not None else (
                ConstantKernel(1.0, constant_value_bounds="fixed") * RBF(1.0, length_scale_bounds="fixed"))
            comp = kern.components()
            mean = np.zeros((X.shape[0], n_t)).squeeze()
            if return_std:
                var = np.full((X.shape[0], n_t), comp.sf2 + (comp.noise or 0.0)).squeeze()
                return mean, np.sqrt(var)
            return mean
        self._ensure_device()
        dev = self._dev
        if (self.predict_dtype != "float32" and self.var_method in ("auto", "inverse")
                and dev.host_path_ok(X.shape[0], return_std)):
            # small batches (the control loop's 1..25 rows): one C call, one synchronisation
            kss = None
            if return_std:
                comp = self.kernel_.components()
                kss = comp.sf2 + (comp.noise or 0.0)
            mean, var = dev.predict_host(X, self._y_train_mean, self._y_train_std, kss, 0.0)
            if not return_std:
                return mean[:, 0] if mean.shape[1] == 1 else mean
            var = np.outer(var, self._y_train_std ** 2)
            if mean.shape[1] == 1:
                mean, var = mean[:, 0], var[:, 0]
            return mean, np.sqrt(var)
        import torch
        # fp32 serving is gated (DeviceGP.predict_gated_dev): a model whose mean would leave the stated 1e-4 goes through
        # the fp64 kernels, and so do single queries whose fp32 variance is too small a fraction of the prior's to carry a
        # 1e-3-accurate standard deviation
        gated = getattr(self, "fp32_gate", True)
        kss = None
        if return_std:
            comp = self.kernel_.components()
            kss = comp.sf2 + (comp.noise or 0.0)        # kernel_.diag(X): RBF diag + WhiteKernel level
        mean_d, var_d = dev.predict_gated_dev(X, self._y_train_mean, self._y_train_std, kss, 0.0, self.predict_dtype,
                                              self.var_method, gated)   # variance clipped at 0 (_gpr.py:479-485)
        if not return_std:
            mean = mean_d.double().cpu().numpy()       # float64 whatever dtype served it, as scikit-learn returns
            return mean[:, 0] if mean.shape[1] == 1 else mean
        both = torch.cat([mean_d.double(), var_d[:, None]], dim=1).cpu().numpy()     # one device->host copy
        mean, var = both[:, :-1], both[:, -1]
        var = np.outer(var, self._y_train_std ** 2)
        if mean.shape[1] == 1:
            mean, var = mean[:, 0], var[:, 0]
        return mean, np.sqrt(var)

    # ------------------------------------------------------------------ host views / persistence
    @property
    def alpha_(self):
        self._ensure_device()
        if getattr(self, "_alpha_host", None) is None:
            a = self._dev.alpha_host()
            self._alpha_host = a[:, 0] if self._y_1d else a
        return self._alpha_host

    @property
    def L_(self):
        self._ensure_device()
        if getattr(self, "_L_host", None) is None:
            self._L_host = self._dev.L_host()
        return self._L_host

    def _refactor(self):
        """The model's factorisation at `kernel_` (final step of fit, and again after unpickling: same route, same bits):
        Gram, factor, inverse factor, alpha and the LML terms as ONE launch chain with one synchronisation (gpk_lml_eval, value
        only; up to 4608 rows factor and inverse factor are one launch) - or, beyond 32 768 rows and 16 features, call by call.
        Returns (log-det / 2, [y_p . alpha_p]).  Raises NotPositiveDefinite."""
        comp = self.kernel_.components()
        dev, D = self._dev, self.n_features_in_
        diag = (comp.noise or 0.0) + float(self.alpha)
        if dev.Np <= dev.INVERSE_EAGER_NP and D <= 16:
            logdet_half, quad, _ = dev.lml_eval(comp.ls_vector(D), comp.sf2, diag, comp.noise or 0.0, False)
            dev.release_grad_buffers()
            return logdet_half, quad
        dev.factorize(comp.ls_vector(D), comp.sf2, diag)
        dev.solve_alpha()
        return dev.lml_terms()

    def _ensure_device(self):
        """Re-create the HBM state after unpickling (deterministic refactorisation)."""
        if self._dev is None:
            if not hasattr(self, "X_train_"):
                raise RuntimeError("This GaussianProcessRegressor instance is not fitted yet.")
            self._dev = DeviceGP(self.X_train_, self._yn, get_backend(self.device))
            imported = getattr(self, "_imported_factor", None)
            comp = self.kernel_.components()
            if imported is not None:
                self._dev.load_factor(imported["L"], comp.ls_vector(self.n_features_in_), comp.sf2)
                self._dev.set_alpha(imported["alpha"])
            else:
                self._refactor()

    def __getstate__(self):
        st = self.__dict__.copy()
        st["_dev"] = None
        st.pop("_lml_dev", None)
        if not isinstance(st.get("device"), (int, type(None))):     # a private Backend (handle + stream): keep its index
            st["device"] = getattr(st["device"], "device_index", None)
        st.pop("_rng", None)
        st["_alpha_host"] = None
        st["_L_host"] = None
        return st

    def __setstate__(self, st):
        self.__dict__.update(st)
        self._dev = None

    @classmethod
    def from_sklearn(cls, skl, device=None, predict_dtype="float64"):
        """Ingest a fitted scikit-learn GaussianProcessRegressor (e.g. unpickled from the
        reference's `gp_models/*.pkl`, `src/px4/simple_gp.py:50-73`): X_train_, alpha_, L_,
        kernel_ parameters and the target normalisation are taken as they are."""
        kern = _convert_sklearn_kernel(skl.kernel_)
        self = cls(kernel=kern, alpha=float(np.ravel(skl.alpha)[0]) if np.ndim(skl.alpha) else float(skl.alpha),
                   optimizer=None, normalize_y=bool(skl.normalize_y), device=device, predict_dtype=predict_dtype)
        self.kernel_ = copy.deepcopy(kern)
        self.X_train_ = np.array(skl.X_train_, dtype=np.float64)
        yn = np.asarray(skl.y_train_, dtype=np.float64)
        self._y_1d = yn.ndim == 1
        self._yn = yn.reshape(self.X_train_.shape[0], -1)
        self.y_train_ = yn
        self._y_train_mean = np.asarray(skl._y_train_mean, dtype=np.float64).reshape(-1)
        self._y_train_std = np.asarray(skl._y_train_std, dtype=np.float64).reshape(-1)
        self.n_features_in_ = self.X_train_.shape[1]
        self.log_marginal_likelihood_value_ = float(getattr(skl, "log_marginal_likelihood_value_", np.nan))
        self._imported_factor = {"L": np.array(skl.L_, dtype=np.float64),
                                 "alpha": np.asarray(skl.alpha_, dtype=np.float64).reshape(self._yn.shape)}
        self._alpha_host = None
        self._L_host = None
        return self


def _convert_sklearn_kernel(k):
    """Map a scikit-learn kernel object (duck-typed by class name) to this package's spec."""
    name = type(k).__name__
    if name == "RBF":
        return RBF(np.array(k.length_scale, dtype=np.float64) if np.iterable(k.length_scale)
                   else float(k.length_scale), k.length_scale_bounds)
    if name == "WhiteKernel":
        return WhiteKernel(float(k.noise_level), k.noise_level_bounds)
    if name == "ConstantKernel":
        return ConstantKernel(float(k.constant_value), k.constant_value_bounds)
    if name == "Sum":
        return _convert_sklearn_kernel(k.k1) + _convert_sklearn_kernel(k.k2)
    if name == "Product":
        return _convert_sklearn_kernel(k.k1) * _convert_sklearn_kernel(k.k2)
    raise ValueError(f"unsupported scikit-learn kernel {name}")
