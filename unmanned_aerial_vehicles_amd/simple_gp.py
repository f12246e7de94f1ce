"""`SimpleQuadrotorGP` — the model seam the GP-MPC ROS 2 nodes call, on MI355X.

Mirrors the public surface of the reference class (`src/px4/simple_gp.py:24-223`):
`add_training_data`, `train_gp`, `predict_residual`, `get_uncertainty`,
`predict_enhanced_dynamics`, `load_model`, `save_dataset`, `get_stats`, the attributes
`X_train`, `Y_train`, `gp_model`, `is_trained`, `training_count`, `prediction_count`, and the
"never raise into the control loop" fallbacks.  Added for the GPU: batched variants
(`predict_residual_batch`, `build_gp_residuals`) that replace the per-horizon-point Python
loop of `src/px4/mpc.py:1490-1506` by one fused kernel call.
"""
from __future__ import annotations

import os
import pickle
from collections import deque

import numpy as np

from .gpr import GaussianProcessRegressor
from .kernels import RBF, WhiteKernel

GPU_BACKEND_AVAILABLE = True  # the class fails loudly at train/predict time if the GPU is absent


class SimpleQuadrotorGP:
    """Multi-output GP on [x,y,z,vx,vy,vz, ax,ay,az,yaw_rate] -> 6 state residuals."""

    def __init__(self, max_data_points=1000, device=None, predict_dtype="float64"):
        self.max_data_points = max_data_points
        self.X_train = deque(maxlen=max_data_points)   # oldest rows are evicted (simple_gp.py:31-32)
        self.Y_train = deque(maxlen=max_data_points)
        self.gp_model = None
        self.is_trained = False
        self.training_count = 0
        self.prediction_count = 0
        self.device = device
        self.predict_dtype = predict_dtype

    # ---- data ---------------------------------------------------------------------------------
    def _to_numpy(self):
        if len(self.X_train) == 0:
            return np.empty((0, 10)), np.empty((0, 6))
        return np.array(self.X_train), np.array(self.Y_train)

    def _nominal_dynamics(self, state, control, dt):
        """Double integrator: x+ = x + dt * [v, a]  (simple_gp.py:146-154)."""
        state = np.asarray(state, dtype=float)
        control = np.asarray(control, dtype=float)
        return state + dt * np.concatenate([state[3:6], control[:3]])

    def add_training_data(self, state, control, state_next, dt=0.02):
        """Quality filters of simple_gp.py:118-140: |v| <= 5, |a_cmd| <= 3, |residual| <= 2."""
        if len(state) < 6 or len(state_next) < 6:
            return
        state = np.asarray(state, dtype=float)
        control = np.asarray(control, dtype=float)
        if np.linalg.norm(state[3:6]) > 5.0 or np.linalg.norm(control[:3]) > 3.0:
            return
        residual = np.asarray(state_next, dtype=float) - self._nominal_dynamics(state, control, dt)
        if np.linalg.norm(residual) > 2.0:
            return
        self.X_train.append(np.concatenate([state[:6], control[:4]]))
        self.Y_train.append(residual.copy())

    def save_dataset(self, csv_path, include_header=True, overwrite=True):
        """16-column `%.18e` CSV (simple_gp.py:75-115)."""
        X, Y = self._to_numpy()
        if X.shape[0] == 0:
            print("No training data to save.")
            return
        from .data import save_dataset_csv
        save_dataset_csv(csv_path, X, Y, include_header=include_header, overwrite=overwrite)

    # ---- training -----------------------------------------------------------------------------
    def train_gp(self):
        """RBF(0.5) + WhiteKernel(0.1), alpha=1e-4, normalize_y, one optimiser restart
        (simple_gp.py:156-185); failures leave the model untrained instead of raising."""
        if len(self.X_train) < 30:
            return
        try:
            X = np.array(list(self.X_train))
            Y = np.array(list(self.Y_train))
            kernel = RBF(length_scale=0.5) + WhiteKernel(noise_level=0.1)
            self.gp_model = GaussianProcessRegressor(kernel=kernel, alpha=1e-4, normalize_y=True,
                                                     n_restarts_optimizer=1, device=self.device,
                                                     predict_dtype=self.predict_dtype)
            self.gp_model.fit(X, Y)
            self.is_trained = True
            self.training_count += 1
            print(f"Simple GP trained with {len(X)} samples (iteration {self.training_count})")
        except Exception as e:  # noqa: BLE001 - reference behaviour: report and stay untrained
            print(f"GP training failed: {e}")
            self.is_trained = False

    def load_model(self, model_path):
        """Reads the `{'gp_model', 'training_count', ...}` pickle of train_gp_offline.py:188-194.
        A scikit-learn regressor inside it is ingested onto the GPU as is."""
        try:
            if not os.path.exists(model_path):
                print(f"Model file not found: {model_path}")
                return False
            with open(model_path, "rb") as f:
                model_data = pickle.load(f)
            model = model_data["gp_model"]
            if not isinstance(model, GaussianProcessRegressor) and hasattr(model, "L_") and hasattr(model, "kernel_"):
                model = GaussianProcessRegressor.from_sklearn(model, device=self.device,
                                                              predict_dtype=self.predict_dtype)
            self.gp_model = model
            self.is_trained = True
            self.training_count = model_data.get("training_count", 0)
            print(f"GP model loaded: {model_path} (samples: {model_data.get('data_points_used', 'unknown')}, "
                  f"created: {model_data.get('timestamp', 'unknown')})")
            return True
        except Exception as e:  # noqa: BLE001
            print(f"Failed to load model: {e}")
            return False

    # ---- prediction ---------------------------------------------------------------------------
    def predict_residual(self, state, control):
        """One row -> (mean (6,), variance = std^2 (6,)); untrained/failed -> (zeros, ones)
        (simple_gp.py:187-201)."""
        if not self.is_trained:
            return np.zeros(6), np.ones(6)
        try:
            x = np.concatenate([state, control]).reshape(1, -1)
            mean, std = self.gp_model.predict(x, return_std=True)
            self.prediction_count += 1
            return mean.flatten(), std.flatten() ** 2
        except Exception as e:  # noqa: BLE001
            print(f"GP prediction failed: {e}")
            return np.zeros(6), np.ones(6)

    def predict_residual_batch(self, X, return_var=True):
        """M rows [state(6), control(4)] in one kernel call -> (mean (M,P), variance (M,P))."""
        X = np.atleast_2d(np.asarray(X, dtype=float))
        P = 6 if self.gp_model is None else self.gp_model._yn.shape[1]
        if not self.is_trained:
            return np.zeros((len(X), P)), np.ones((len(X), P))
        try:
            if return_var:
                mean, std = self.gp_model.predict(X, return_std=True)
                self.prediction_count += len(X)
                return mean.reshape(len(X), -1), std.reshape(len(X), -1) ** 2
            mean = self.gp_model.predict(X)
            self.prediction_count += len(X)
            return mean.reshape(len(X), -1), None
        except Exception as e:  # noqa: BLE001
            print(f"GP prediction failed: {e}")
            return np.zeros((len(X), P)), np.ones((len(X), P))

    def build_gp_residuals(self, X_guess, U_guess, dt, gain=0.1, n_states=6):
        """Horizon-batched counterpart of `QuadrotorMPC._build_gp_residuals`
        (src/px4/mpc.py:1475-1511): D[3:6, k] = gain * mean_k[3:6] / dt for every stage k, from
        ONE batched mean prediction.  X_guess (6, N+1) or (R, 6, N+1) for R rollouts;
        U_guess (4, N) / (R, 4, N).  Returns D (6, N) / (R, 6, N)."""
        X_guess = np.asarray(X_guess, dtype=float)
        U_guess = np.asarray(U_guess, dtype=float)
        single = X_guess.ndim == 2
        if single:
            X_guess, U_guess = X_guess[None], U_guess[None]
        R, _, N = U_guess.shape
        D = np.zeros((R, n_states, N))
        if self.is_trained:
            rows = np.concatenate([X_guess[:, :6, :N], U_guess[:, :4, :]], axis=1)    # (R, 10, N)
            rows = rows.transpose(0, 2, 1).reshape(R * N, -1)
            mean, _ = self.predict_residual_batch(rows, return_var=False)
            if mean.shape[1] >= n_states:
                acc = gain * (mean / dt)[:, 3:6].reshape(R, N, 3)
                D[:, 3:6, :] = acc.transpose(0, 2, 1)
        return D[0] if single else D

    def predict_horizon_gated(self, X_guess, U_guess, confidence_threshold, n_states=6):
        """Batched counterpart of `_get_gp_predictions_for_horizon`
        (src/px4/mpc_direct_rates.py:317-355): one mean+variance call for the whole horizon; stage k keeps
        the GP residual only if sqrt(sum_p var_kp) < confidence_threshold, otherwise zeros (nominal
        dynamics).  Returns (N, n_states)."""
        X_guess = np.asarray(X_guess, dtype=float)
        U_guess = np.asarray(U_guess, dtype=float)
        N = U_guess.shape[1]
        out = np.zeros((N, n_states))
        if not self.is_trained:
            return out
        rows = np.concatenate([X_guess[:6, :N], U_guess[:4, :N]], axis=0).T
        mean, var = self.predict_residual_batch(rows, return_var=True)
        keep = np.sqrt(np.sum(var, axis=1)) < confidence_threshold
        out[keep] = mean[keep, :n_states]
        return out

    def get_uncertainty(self, state, control):
        _, variance = self.predict_residual(state, control)
        return np.mean(np.sqrt(variance))

    def predict_enhanced_dynamics(self, state, control, dt):
        residual_mean, _ = self.predict_residual(state, control)
        return self._nominal_dynamics(state, control, dt) + residual_mean

    def get_stats(self):
        return {
            "is_trained": self.is_trained,
            "data_points": len(self.X_train),
            "training_iterations": self.training_count,
            "predictions_made": self.prediction_count,
            "sklearn_available": False,      # key kept for the reference's log lines; sklearn is not used
            "backend": "mi355x-hip",
        }


class SimpleGPEnhancedMPC:
    """Confidence-gated dynamics wrapper (simple_gp.py:226-260)."""

    def __init__(self, gp_model, confidence_threshold=0.5):
        self.gp_model = gp_model
        self.confidence_threshold = confidence_threshold
        self.gp_usage = 0
        self.nominal_usage = 0

    def enhanced_dynamics_function(self, state, control, dt):
        if not self.gp_model.is_trained:
            self.nominal_usage += 1
            return self.gp_model._nominal_dynamics(state, control, dt)
        if self.gp_model.get_uncertainty(state, control) < self.confidence_threshold:
            self.gp_usage += 1
            return self.gp_model.predict_enhanced_dynamics(state, control, dt)
        self.nominal_usage += 1
        return self.gp_model._nominal_dynamics(state, control, dt)

    def get_usage_ratio(self):
        total = self.gp_usage + self.nominal_usage
        return 0.0 if total == 0 else self.gp_usage / total * 100
