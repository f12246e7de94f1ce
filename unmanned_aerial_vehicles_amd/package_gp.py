"""`GaussianProcess` of the `quadrotor_gp_mpc` ROS package, on MI355X.

Mirrors the non-ROS surface of `quadrotor_gp_mpc/quadrotor_gp_mpc/gaussian_process.py:63-394`
(`add_training_data`, `fit`, `predict`, `log_marginal_likelihood`, `optimize_hyperparameters`,
`train_gp`, `compute_kernel_matrix`, `training_data_callback`, `prediction_request_callback`, `save_model`,
`load_model`, the callable `kernel` with `.gradient`, attributes `kernel.length_scale`, `kernel.signal_variance`,
`noise_variance`, `X_train`, `Y_train`, `L`, `alpha`).  Only the `rclpy.Node` base class (publishers,
subscriptions, the timer) stays in the caller: a node wraps this object and subscribes its two callbacks.

Semantics kept from the reference: no target normalisation; K = sf2 exp(-d2 / (2 l^2)) + noise I;
predictive variance = sf2 - |L^-1 k*|^2 (no noise term), floored at 1e-10 and tiled over the
outputs; not-PD fit -> noise x10 (reset to 0.01 above 1.0), alpha = 0, L = None; unfitted predict
-> (zeros, sf2 * ones).  Squared distances use exact differences on the GPU rather than the
reference's norm expansion (`:38`); the two differ by ~3e-14 on the flight data.

Thread-safety: `fit()` builds a fresh device model and swaps it in atomically, so a predict running
on another executor thread keeps using the previous factor (the reference shares state unlocked,
`quadrotor_gp_mpc/quadrotor_gp_mpc/main.py:818-826`).
"""
from __future__ import annotations

import threading

import numpy as np
import scipy.optimize

from ._lib import NotPositiveDefinite
from .device import DeviceGP, get_backend

LOG_2PI = float(np.log(2.0 * np.pi))


class RBFKernel:
    """The package's kernel object (gaussian_process.py:19-60): hyper-parameter holder AND callable.

    `kernel(X1, X2)` and `kernel.gradient(X1, X2)` run on the GPU (`gpk_rbf_kernel_grad`: one launch, squared distances by
    exact differences - the reference's norm expansion `:38` differs from them by ~3e-14 on the flight data) and return host
    arrays like the reference: K (n1, n2), and (dK/dlength_scale, dK/dsignal_variance) = (K d2 / l^3, K / sf2)."""

    def __init__(self, length_scale=1.0, signal_variance=1.0, device=None):
        self.length_scale = length_scale
        self.signal_variance = signal_variance
        self.device = device

    def _evaluate(self, X1, X2, want_q):
        import ctypes as C

        from . import _lib
        from .device import _torch
        torch = _torch()
        X1 = np.ascontiguousarray(np.atleast_2d(X1), dtype=np.float64)
        X2 = np.ascontiguousarray(np.atleast_2d(X2), dtype=np.float64)
        if X1.shape[1] != X2.shape[1]:
            raise ValueError(f"feature dimensions differ: {X1.shape[1]} and {X2.shape[1]}")
        n1, D = X1.shape
        n2 = X2.shape[0]
        if n1 == 0 or n2 == 0:
            z = np.zeros((n1, n2))
            return z, (z.copy() if want_q else None)
        be = get_backend(self.device)
        ls = np.full(D, float(self.length_scale))
        with be.lock:
            be.bind_stream()
            a = be.upload(X1)
            b = a if X2 is X1 else be.upload(X2)
            K = be.empty((n1, n2), torch.float64)
            Q = be.empty((n1, n2), torch.float64) if want_q else None
            be.check(be.lib.gpk_rbf_kernel_grad(be.h, C.c_void_p(a.data_ptr()), n1, C.c_void_p(b.data_ptr()), n2, D,
                                                ls.ctypes.data_as(_lib._dp), float(self.signal_variance),
                                                C.c_void_p(K.data_ptr()), C.c_void_p(Q.data_ptr()) if want_q else None, n2))
            Kh = K.cpu().numpy()
            Qh = Q.cpu().numpy() if want_q else None
        return Kh, Qh

    def __call__(self, X1, X2):
        """RBF kernel matrix (n1 x n2), gaussian_process.py:26-41."""
        return self._evaluate(X1, X2, False)[0]

    def gradient(self, X1, X2):
        """(dK/dlength_scale, dK/dsignal_variance), gaussian_process.py:43-60."""
        K, Q = self._evaluate(X1, X2, True)
        return Q / float(self.length_scale), K / float(self.signal_variance)


class _Logger:
    def __getattr__(self, _name):
        return lambda *a, **k: None


class GaussianProcess:
    def __init__(self, input_dim=16, output_dim=12, device=None, predict_dtype="float64", logger=None):
        self.input_dim = input_dim
        self.output_dim = output_dim
        self.kernel = RBFKernel(length_scale=1.0, signal_variance=1.0, device=device)
        self.noise_variance = 0.01
        self.X_train = np.empty((0, input_dim))
        self.Y_train = np.empty((0, output_dim))
        self.K_inv = None
        self.L = None
        self.alpha = None
        self.max_data_points = 1000
        self.data_collection_active = True
        self.device = device
        self.predict_dtype = predict_dtype
        self._logger = logger or _Logger()
        self._model = None          # (DeviceGP, sf2) snapshot used by predict
        self._swap = threading.Lock()

    def get_logger(self):
        return self._logger

    # ---- data (gaussian_process.py:126-156) ---------------------------------------------------
    def add_training_data(self, X, Y):
        X = np.atleast_2d(X)
        Y = np.atleast_2d(Y)
        if X.shape[1] != self.input_dim or Y.shape[1] != self.output_dim:
            self.get_logger().error(f"Data dimension mismatch: X={X.shape}, Y={Y.shape}")
            return
        self.X_train = np.vstack([self.X_train, X])
        self.Y_train = np.vstack([self.Y_train, Y])
        if len(self.X_train) > self.max_data_points:      # FIFO eviction
            excess = len(self.X_train) - self.max_data_points
            self.X_train = self.X_train[excess:]
            self.Y_train = self.Y_train[excess:]
        self.K_inv = None
        self.L = None
        self.alpha = None

    # ---- kernel matrix (gaussian_process.py:158-171) ----------------------------------------------
    def compute_kernel_matrix(self, X1, X2=None):
        """`kernel(X1, X2)`; with X2 omitted the training-side matrix `kernel(X1, X1) + noise_variance I` - built by `gpk_gram`,
        the launch `fit()` factors (symmetric halves bit-identical, diagonal exactly sf2 + noise)."""
        if X2 is not None:
            return self.kernel(X1, X2)
        X1 = np.atleast_2d(X1)
        dev = DeviceGP(X1, np.zeros((len(X1), 1)), get_backend(self.device))
        return dev.gram_host(float(self.kernel.length_scale), float(self.kernel.signal_variance), float(self.noise_variance))

    # ---- fit (gaussian_process.py:173-201) -----------------------------------------------------
    def _factor(self, length_scale, signal_variance, noise_variance):
        dev = DeviceGP(self.X_train, self.Y_train, get_backend(self.device))
        dev.factorize(float(length_scale), float(signal_variance), float(noise_variance))
        dev.solve_alpha()
        return dev

    def fit(self):
        if len(self.X_train) < 2:
            self.get_logger().warning("Insufficient training data for GP fitting")
            return
        try:
            dev = self._factor(self.kernel.length_scale, self.kernel.signal_variance, self.noise_variance)
            with self._swap:      # (the zero / one target scaling is kept with the model: predict_host caches by identity)
                self._model = (dev, float(self.kernel.signal_variance), np.zeros(dev.P), np.ones(dev.P))
            self.alpha = dev.alpha_host()
            self.L = _LazyFactor(dev)
        except NotPositiveDefinite as e:
            self.get_logger().error(f"Numerical error in GP fitting: {e}")
            self.noise_variance *= 10
            if self.noise_variance > 1.0:
                self.noise_variance = 0.01
            self.alpha = np.zeros((len(self.X_train), self.output_dim))
            self.L = None
            with self._swap:
                self._model = None

    # ---- predict (gaussian_process.py:203-241) ---------------------------------------------------
    def predict(self, X_test):
        X_test = np.atleast_2d(X_test)
        prior = (np.zeros((len(X_test), self.output_dim)),
                 np.ones((len(X_test), self.output_dim)) * self.kernel.signal_variance)
        if len(self.X_train) == 0 or self.alpha is None:
            return prior
        with self._swap:
            model = self._model
        if model is None:           # failed fit: alpha = 0, L missing -> the reference's except branch
            return prior
        try:
            dev, sf2, zeros, ones = model
            if self.predict_dtype != "float32" and dev.host_path_ok(len(X_test), True):
                # small batches: one C call and one synchronisation (gpk_predict_host)
                mean, var = dev.predict_host(X_test, zeros, ones, sf2, 1e-10)
                return mean, np.tile(var.reshape(-1, 1), (1, self.output_dim))
            # (predict_dtype "float32" is a request: DeviceGP's serving gates may route the model, or single rows'
            # variances, to the fp64 kernels)
            mean, var = dev.predict_gated_dev(X_test, zeros, ones, sf2, 1e-10, self.predict_dtype)
            mean = mean.double().cpu().numpy()
            var = np.tile(var.cpu().numpy().reshape(-1, 1), (1, self.output_dim))
            return mean, var
        except Exception as e:  # noqa: BLE001
            self.get_logger().error(f"Error in GP prediction: {e}")
            return prior

    # ---- LML (gaussian_process.py:243-265) --------------------------------------------------------
    def log_marginal_likelihood(self):
        if len(self.X_train) < 2 or self.L is None:
            return -np.inf
        try:
            with self._swap:          # a refit on another thread replaces the snapshot; never read it half-swapped
                model = self._model
            if model is None:
                return -np.inf
            logdet_half, quad = model[0].lml_terms()
            n = len(self.X_train)
            return float(-0.5 * (2.0 * logdet_half + quad.sum() + n * self.output_dim * LOG_2PI))
        except Exception as e:  # noqa: BLE001
            self.get_logger().error(f"Error computing log marginal likelihood: {e}")
            return -np.inf

    # ---- hyper-parameters (gaussian_process.py:267-324) ----------------------------------------------
    def optimize_hyperparameters(self, use_gradient=True):
        """L-BFGS-B over log(length_scale, signal_variance, noise_variance), maxiter 50.  As in the
        reference every objective evaluation installs the trial hyper-parameters and refits (so after an
        unsuccessful run the model sits at the last trial point); non-finite likelihood restores the
        previous values and returns 1e6.  The reference differentiates numerically; here the analytic
        gradient from the fused K6b kernel is used unless `use_gradient=False`."""
        if len(self.X_train) < 10:
            return
        D = self.X_train.shape[1]

        def objective(params):
            old = (self.kernel.length_scale, self.kernel.signal_variance, self.noise_variance)
            self.kernel.length_scale = float(np.exp(params[0]))
            self.kernel.signal_variance = float(np.exp(params[1]))
            self.noise_variance = float(np.exp(params[2]))
            noise = self.noise_variance
            self.fit()
            nll = -self.log_marginal_likelihood()
            if not np.isfinite(nll):
                self.kernel.length_scale, self.kernel.signal_variance, self.noise_variance = old
                return (1e6, np.zeros(3)) if use_gradient else 1e6
            if not use_gradient:
                return float(nll)
            with self._swap:
                model = self._model
            g = model[0].lml_grad(noise)                  # d LML / d log [ls_d..., noise, sf2]
            return float(nll), -np.array([np.sum(g[:D]), g[D + 1], g[D]])

        x0 = np.log([self.kernel.length_scale, self.kernel.signal_variance, self.noise_variance])
        try:
            result = scipy.optimize.minimize(objective, x0, method="L-BFGS-B", jac=bool(use_gradient),
                                             options={"maxiter": 50})
            self.last_optimize_result = result
            if result.success:
                self.kernel.length_scale = float(np.exp(result.x[0]))
                self.kernel.signal_variance = float(np.exp(result.x[1]))
                self.noise_variance = float(np.exp(result.x[2]))
                self.fit()
        except Exception as e:  # noqa: BLE001
            self.get_logger().error(f"Hyperparameter optimization failed: {e}")

    # ---- topic callbacks (gaussian_process.py:326-358), Node-less ------------------------------------
    # A ROS node that wraps this object subscribes these two methods as they stand: they only touch `msg.data` (any sequence
    # of floats - `std_msgs/Float64MultiArray` in the reference) and hand results to `prediction_pub` / `uncertainty_pub`,
    # anything with a `publish(msg)` method (rclpy publishers; unset: the results are only returned).  `message_type` builds the
    # outgoing message (default: a plain object with a `.data` list, so the path runs without rclpy).
    prediction_pub = None
    uncertainty_pub = None

    class _Msg:
        def __init__(self):
            self.data = []

    message_type = _Msg

    def training_data_callback(self, msg):
        data = np.array(msg.data, dtype=np.float64)
        expected_size = self.input_dim + self.output_dim
        if len(data) != expected_size:
            self.get_logger().error(f"Invalid training data size: {len(data)} != {expected_size}")
            return
        self.add_training_data(data[: self.input_dim].reshape(1, -1), data[self.input_dim:].reshape(1, -1))

    def prediction_request_callback(self, msg):
        X_test = np.array(msg.data, dtype=np.float64).reshape(1, -1)
        if X_test.shape[1] != self.input_dim:
            self.get_logger().error(f"Invalid prediction input size: {X_test.shape[1]} != {self.input_dim}")
            return None
        mean, var = self.predict(X_test)
        pred_msg = self.message_type()
        pred_msg.data = mean.flatten().tolist()
        unc_msg = self.message_type()
        unc_msg.data = np.sqrt(var).flatten().tolist()
        if self.prediction_pub is not None:
            self.prediction_pub.publish(pred_msg)
        if self.uncertainty_pub is not None:
            self.uncertainty_pub.publish(unc_msg)
        return pred_msg, unc_msg

    def train_gp(self):
        """Periodic training callback (gaussian_process.py:360-367)."""
        if len(self.X_train) >= 10:
            if len(self.X_train) % 50 == 0:
                self.optimize_hyperparameters()
            else:
                self.fit()

    # ---- persistence (gaussian_process.py:369-394) ---------------------------------------------------
    def save_model(self, filename):
        np.savez(filename, X_train=self.X_train, Y_train=self.Y_train, length_scale=self.kernel.length_scale,
                 signal_variance=self.kernel.signal_variance, noise_variance=self.noise_variance)

    def load_model(self, filename):
        try:
            data = np.load(filename)
            self.X_train = data["X_train"]
            self.Y_train = data["Y_train"]
            self.kernel.length_scale = float(data["length_scale"])
            self.kernel.signal_variance = float(data["signal_variance"])
            self.noise_variance = float(data["noise_variance"])
            self.fit()
        except Exception as e:  # noqa: BLE001
            self.get_logger().error(f"Failed to load GP model: {e}")


class _LazyFactor:
    """`gp.L` placeholder: truthy once fitted, materialises the (N, N) host copy on demand."""

    def __init__(self, dev):
        self._dev = dev
        self._host = None

    def __array__(self, dtype=None, copy=None):
        if self._host is None:
            self._host = self._dev.L_host()
        return self._host if dtype is None else self._host.astype(dtype)
