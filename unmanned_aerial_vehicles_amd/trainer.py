"""Per-output ARD GP trainer / loader — the reference's alternative offline trainer on MI355X.

`GPTrainer.load_training_data` mirrors `src/px4/gp_trainer.py:49-119` (flight_data_*.npz -> [state, control] rows and
double-integrator residuals); `GPTrainer.train_gp_models` mirrors `src/px4/gp_trainer.py:121-205`: 80/20 split with seed 42,
one independent GP per residual component, z-scored inputs and target,
`C(1, fixed) * RBF(ls in R^D, bounds 0.1..10) + WhiteKernel(0.01, 1e-5..10)`, `alpha=1e-6`,
3 optimiser restarts.  `PreTrainedGP.predict_residual` mirrors `src/px4/pretrained_gp.py:52-98`
(returns the *standard deviation*, un-scaled by the target scaler), plus a batched variant.
The pickle layout `{'gp_models','scalers_X','scalers_y','training_stats','model_name',
'creation_time'}` is the reference's (`gp_trainer.py:214-221`).
"""
from __future__ import annotations

import os
import pickle
import time
from datetime import datetime

import numpy as np

from .gpr import GaussianProcessRegressor
from .kernels import RBF, ConstantKernel, WhiteKernel

OUTPUT_NAMES = ["x_residual", "y_residual", "z_residual", "vx_residual", "vy_residual", "vz_residual"]


class StandardScaler:
    """z-scoring with the population variance, as the reference gets it from scikit-learn 1.7.2
    (`gp_trainer.py:152-156` -> `sklearn/preprocessing/_data.py` StandardScaler.partial_fit): the variance is the
    corrected two-pass form of `_incremental_mean_and_var`, and a feature counts as constant (scale 1) when its variance
    is not above `n eps var + (n mean eps)^2` (`_is_constant_feature`) - NOT when its std is below 10 eps: the yaw-rate
    column of the flight CSVs has a std of 1e-21 and is scaled to unit variance like every other column."""

    def fit(self, A):
        A = np.asarray(A, dtype=np.float64)
        n = A.shape[0]
        self.mean_ = A.sum(axis=0) / n
        temp = A - self.mean_
        correction = temp.sum(axis=0)
        self.var_ = ((temp ** 2).sum(axis=0) - correction ** 2 / n) / n
        eps = np.finfo(np.float64).eps
        constant = self.var_ <= n * eps * self.var_ + (n * self.mean_ * eps) ** 2
        scale = np.sqrt(self.var_)
        self.scale_ = np.where(constant | (scale == 0.0), 1.0, scale)
        return self

    def transform(self, A):
        return (np.asarray(A, dtype=np.float64) - self.mean_) / self.scale_

    def fit_transform(self, A):
        return self.fit(A).transform(A)

    def inverse_transform(self, A):
        return np.asarray(A, dtype=np.float64) * self.scale_ + self.mean_


def _as_scaler(obj):
    """This package's `StandardScaler` from any object carrying `mean_` and `scale_` (e.g. the scikit-learn
    scalers inside a pickle written by the reference's `gp_trainer.py:214-221`)."""
    if isinstance(obj, StandardScaler):
        return obj
    s = StandardScaler()
    s.mean_ = np.asarray(obj.mean_, dtype=np.float64).copy()
    s.scale_ = np.asarray(obj.scale_, dtype=np.float64).copy()
    return s


def train_test_split(X, y, test_size=0.2, random_state=42):
    """Shuffled split: permutation from RandomState(seed); the first ceil(test_size * n) indices are
    the test set, the rest the training set (the ShuffleSplit rule the reference relies on)."""
    n = len(X)
    n_test = int(np.ceil(test_size * n))
    n_train = int(np.floor((1.0 - test_size) * n))
    perm = np.random.RandomState(random_state).permutation(n)
    te, tr = perm[:n_test], perm[n_test:n_test + n_train]
    return X[tr], X[te], y[tr], y[te]


def _r2(y_true, y_pred):
    ss_res = np.sum((y_true - y_pred) ** 2)
    ss_tot = np.sum((y_true - np.mean(y_true)) ** 2)
    return 1.0 - ss_res / ss_tot if ss_tot > 0 else 0.0


class GPTrainer:
    def __init__(self, data_dir="gp_data", model_dir="gp_models", device=None):
        self.data_dir = data_dir
        self.model_dir = model_dir
        self.device = device
        self.gp_models, self.scalers_X, self.scalers_y, self.training_stats = {}, {}, {}, {}

    def load_training_data(self, max_samples=None):
        """`flight_data_*.npz` files of `data_dir` -> (X (n, 10) = [state_prev | control], y (n, 6) = residuals against the
        double-integrator nominal model).  Mirrors `src/px4/gp_trainer.py:49-102`: every file carries `states_prev`
        (n, 6), `controls` (n, 4), `states_next` (n, 6), `dt_values` (n,); residual = next - nominal(prev, control, dt);
        `max_samples` draws rows without replacement from the GLOBAL NumPy RNG (`np.random.choice`, :94-97), as the
        reference does.  The files are visited in sorted order (the reference's unsorted glob makes the row order
        depend on the file system)."""
        import glob
        data_files = sorted(glob.glob(os.path.join(self.data_dir, "flight_data_*.npz")))
        if not data_files:
            raise FileNotFoundError(f"No training data found in {self.data_dir}")
        all_X, all_y = [], []
        for data_file in data_files:
            with np.load(data_file) as data:
                states_prev = np.asarray(data["states_prev"], dtype=np.float64)
                controls = np.asarray(data["controls"], dtype=np.float64)
                states_next = np.asarray(data["states_next"], dtype=np.float64)
                dt_values = np.asarray(data["dt_values"], dtype=np.float64).reshape(-1)
            n = len(states_prev)            # (the reference walks range(len(states_prev)))
            nominal = self._nominal_dynamics(states_prev[:n], controls[:n], dt_values[:n, None])
            all_X.append(np.concatenate([states_prev[:n], controls[:n]], axis=1))
            all_y.append(states_next[:n] - nominal)
        X, y = np.concatenate(all_X, axis=0), np.concatenate(all_y, axis=0)
        if max_samples and len(X) > max_samples:
            indices = np.random.choice(len(X), max_samples, replace=False)
            X, y = X[indices], y[indices]
        return X, y

    @staticmethod
    def _nominal_dynamics(state, control, dt):
        """Double integrator (`gp_trainer.py:104-119`): pos + vel dt, vel + accel dt; rows at once (the same fp64
        operations per element as the reference's per-row form)."""
        state, control = np.asarray(state, dtype=np.float64), np.asarray(control, dtype=np.float64)
        pos, vel, accel = state[..., :3], state[..., 3:6], control[..., :3]
        return np.concatenate([pos + vel * dt, vel + accel * dt], axis=-1)

    def train_gp_models(self, X, y, test_size=0.2, n_restarts_optimizer=3, optimizer="fmin_l_bfgs_b", batched=True):
        """batched=True (default): the per-output models share X, so they are trained together — one fused
        launch chain per optimiser evaluation (`BatchedARDGP`); batched=False trains them one by one as the
        reference does."""
        X = np.asarray(X, dtype=np.float64)
        y = np.asarray(y, dtype=np.float64)
        X_tr, X_te, y_tr, y_te = train_test_split(X, y, test_size=test_size, random_state=42)
        results = {}
        names = [n for i, n in enumerate(OUTPUT_NAMES[: y.shape[1]]) if np.std(y_tr[:, i]) >= 1e-6]
        if batched and 2 <= len(names) <= 8 and optimizer == "fmin_l_bfgs_b":
            from .batched import BatchedARDGP
            idx = [OUTPUT_NAMES.index(n) for n in names]
            sx = StandardScaler().fit(X_tr)
            sys_ = [StandardScaler().fit(y_tr[:, i].reshape(-1, 1)) for i in idx]
            Ys = np.column_stack([s_.transform(y_tr[:, i].reshape(-1, 1)).ravel() for s_, i in zip(sys_, idx)])
            bg = BatchedARDGP(length_scale=1.0, length_scale_bounds=(0.1, 10.0), noise_level=0.01,
                              noise_level_bounds=(1e-5, 1e1), alpha=1e-6, normalize_y=False, optimizer=optimizer,
                              n_restarts_optimizer=n_restarts_optimizer, device=self.device).fit(sx.transform(X_tr), Ys)
            pred_s = bg.predict(sx.transform(X_te))
            for j, (name, i) in enumerate(zip(names, idx)):
                gp, sy = bg.models[j], sys_[j]
                pred = sy.inverse_transform(pred_s[:, j].reshape(-1, 1)).flatten()
                mse = float(np.mean((y_te[:, i] - pred) ** 2))
                self.gp_models[name], self.scalers_X[name], self.scalers_y[name] = gp, sx, sy
                results[name] = {"mse": mse, "rmse": float(np.sqrt(mse)), "r2": float(_r2(y_te[:, i], pred)),
                                 "kernel": str(gp.kernel_), "log_marginal_likelihood": gp.log_marginal_likelihood()}
            self.training_stats = results
            return results
        for i, name in enumerate(OUTPUT_NAMES[: y.shape[1]]):
            yi_tr, yi_te = y_tr[:, i], y_te[:, i]
            if np.std(yi_tr) < 1e-6:          # gp_trainer.py:148-150
                continue
            sx, sy = StandardScaler(), StandardScaler()
            Xs = sx.fit_transform(X_tr)
            ys = sy.fit_transform(yi_tr.reshape(-1, 1)).flatten()
            kernel = (ConstantKernel(1.0, constant_value_bounds="fixed")
                      * RBF(length_scale=[1.0] * X.shape[1], length_scale_bounds=(0.1, 10.0))
                      + WhiteKernel(noise_level=0.01, noise_level_bounds=(1e-5, 1e1)))
            gp = GaussianProcessRegressor(kernel=kernel, n_restarts_optimizer=n_restarts_optimizer, alpha=1e-6,
                                          normalize_y=False, optimizer=optimizer, device=self.device)
            gp.fit(Xs, ys)
            pred = sy.inverse_transform(gp.predict(sx.transform(X_te)).reshape(-1, 1)).flatten()
            mse = float(np.mean((yi_te - pred) ** 2))
            self.gp_models[name], self.scalers_X[name], self.scalers_y[name] = gp, sx, sy
            results[name] = {"mse": mse, "rmse": float(np.sqrt(mse)), "r2": float(_r2(yi_te, pred)),
                             "kernel": str(gp.kernel_), "log_marginal_likelihood": gp.log_marginal_likelihood()}
        self.training_stats = results
        return results

    def save_models(self, model_name=None):
        if model_name is None:
            model_name = f"gp_model_{datetime.now().strftime('%Y%m%d_%H%M%S')}"
        os.makedirs(self.model_dir, exist_ok=True)
        path = os.path.join(self.model_dir, f"{model_name}.pkl")
        with open(path, "wb") as f:
            pickle.dump({"gp_models": self.gp_models, "scalers_X": self.scalers_X, "scalers_y": self.scalers_y,
                         "training_stats": self.training_stats, "model_name": model_name,
                         "creation_time": time.time()}, f)
        return path

    def load_models(self, model_path):
        with open(model_path, "rb") as f:
            d = pickle.load(f)
        self.gp_models = {n: (m if isinstance(m, GaussianProcessRegressor)
                              else GaussianProcessRegressor.from_sklearn(m, device=self.device))
                          for n, m in d["gp_models"].items()}
        self.scalers_X = {n: _as_scaler(v) for n, v in d["scalers_X"].items()}
        self.scalers_y = {n: _as_scaler(v) for n, v in d["scalers_y"].items()}
        self.training_stats = d["training_stats"]


class PreTrainedGP:
    def __init__(self, model_path):
        self.model_path = model_path
        self.gp_models, self.scalers_X, self.scalers_y, self.training_stats = {}, {}, {}, {}
        self.is_loaded = False
        self.load_models()

    def load_models(self):
        if not os.path.exists(self.model_path):
            print(f"GP model file not found: {self.model_path}")
            return False
        try:
            with open(self.model_path, "rb") as f:
                d = pickle.load(f)
            return self.load_dict(d)
        except Exception as e:  # noqa: BLE001
            print(f"Failed to load GP models: {e}")
            return False

    def load_dict(self, d, device=None):
        """Install the content of a model file.  The file may have been written by this package's `GPTrainer` or
        by the reference's (`src/px4/gp_trainer.py:214-221`: scikit-learn regressors and scikit-learn
        `StandardScaler`s): foreign regressors are ingested with `GaussianProcessRegressor.from_sklearn`
        (X_train_, alpha_, L_, kernel_ and the target normalisation taken as they are), and any object carrying
        `mean_` / `scale_` serves as a scaler.  A component that cannot be converted is dropped (its prediction is
        then the reference's (0, 1e6) fallback, `pretrained_gp.py:93-96`)."""
        models, sxs, sys_ = {}, {}, {}
        for name, m in d["gp_models"].items():
            try:
                if not isinstance(m, GaussianProcessRegressor):
                    m = GaussianProcessRegressor.from_sklearn(m, device=device)
                models[name] = m
                sxs[name] = _as_scaler(d["scalers_X"][name])
                sys_[name] = _as_scaler(d["scalers_y"][name])
            except Exception as e:  # noqa: BLE001
                print(f"GP model {name} could not be loaded: {e}")
                models.pop(name, None)
        if d["gp_models"] and not models:
            # nothing survived the conversion: as in the reference, where a failed load leaves `is_loaded` False
            # (`pretrained_gp.py:34-50`) and every prediction is the (0, 1e6) fallback
            self.gp_models, self.scalers_X, self.scalers_y, self.training_stats = {}, {}, {}, {}
            self._fused_bg = None
            self.is_loaded = False
            return False
        self.gp_models, self.scalers_X, self.scalers_y = models, sxs, sys_
        self.training_stats = d.get("training_stats", {})
        self._fused_bg = None
        self.is_loaded = True
        return True

    def predict_residual(self, state, control):
        """One query -> (mean (6,), std (6,)); missing / failing components -> (0, 1e6).  Never raises
        (`pretrained_gp.py:52-98`: the control loop calls this every step)."""
        try:
            mean, std = self.predict_residual_batch(
                np.concatenate([np.asarray(state, float)[:6], np.asarray(control, float)[:4]]).reshape(1, -1))
            return mean[0], std[0]
        except Exception as e:  # noqa: BLE001
            print(f"GP prediction failed: {e}")
            return np.zeros(6), np.ones(6) * 1e6

    def _fused(self):
        """All loaded models share the input scaler and training inputs (they do when written by `GPTrainer`):
        evaluate their means with one fused launch."""
        if getattr(self, "_fused_bg", None) is None:
            self._fused_bg = False
            try:
                names = [n for n in OUTPUT_NAMES if n in self.gp_models]
                if 2 <= len(names) <= 8:
                    sx0 = self.scalers_X[names[0]]
                    m0 = self.gp_models[names[0]]
                    x0 = getattr(m0, "X_train_", None)
                    same = x0 is not None and all(
                        np.array_equal(self.scalers_X[n].mean_, sx0.mean_) and
                        np.array_equal(self.scalers_X[n].scale_, sx0.scale_) and
                        getattr(self.gp_models[n], "X_train_", np.empty(0)).shape == x0.shape and
                        np.array_equal(self.gp_models[n].X_train_, x0) and
                        getattr(self.gp_models[n], "_yn", np.empty((0, 0))).shape[1:] == (1,) for n in names)
                    if same:
                        from .batched import BatchedARDGP
                        bg = BatchedARDGP(optimizer=None)
                        bg.models = [self.gp_models[n] for n in names]
                        self._fused_bg = (bg, names)
            except Exception as e:  # noqa: BLE001 - never into the control loop: the per-model path below still serves
                print(f"fused per-axis prediction unavailable: {e}")
        return self._fused_bg

    def predict_residual_batch(self, X, return_std=True):
        X = np.atleast_2d(np.asarray(X, dtype=np.float64))
        if not self.is_loaded:
            return np.zeros((len(X), 6)), np.ones((len(X), 6)) * 1e6
        fused = self._fused()
        if fused:
            # models written by `GPTrainer` share scaler and inputs: all of them in one call (<= 32 rows: one C call
            # and two launches for means and stds, `gpk_predict_host_multi`; larger batches: one fused mean launch)
            bg, names = fused
            try:
                out = bg.predict(self.scalers_X[names[0]].transform(X), return_std=return_std)
                ms, ss = out if return_std else (out, None)
                mean = np.zeros((len(X), 6))
                std = np.full((len(X), 6), 1e6) if return_std else None
                for j, n in enumerate(names):
                    i = OUTPUT_NAMES.index(n)
                    mean[:, i] = self.scalers_y[n].inverse_transform(ms[:, j].reshape(-1, 1)).ravel()
                    if return_std:
                        std[:, i] = np.abs(ss[:, j] * self.scalers_y[n].scale_[0])
                return mean, std
            except Exception as e:  # noqa: BLE001
                print(f"GP prediction failed: {e}")
                if not return_std:
                    return np.zeros((len(X), 6)), None
        mean = np.zeros((len(X), 6))
        std = np.full((len(X), 6), 1e6)
        for i, name in enumerate(OUTPUT_NAMES):
            if name not in self.gp_models:
                continue
            try:
                Xs = self.scalers_X[name].transform(X)
                m, s = self.gp_models[name].predict(Xs, return_std=True)
                mean[:, i] = self.scalers_y[name].inverse_transform(m.reshape(-1, 1)).flatten()
                std[:, i] = np.abs(s * self.scalers_y[name].scale_[0])
            except Exception as e:  # noqa: BLE001
                print(f"GP prediction failed for {name}: {e}")
                mean[:, i], std[:, i] = 0.0, 1e6
        return mean, std

    def get_uncertainty(self, state, control):
        return float(np.mean(self.predict_residual(state, control)[1]))

    def get_stats(self):
        return {"is_loaded": self.is_loaded, "model_path": self.model_path,
                "available_models": list(self.gp_models.keys()), "training_stats": self.training_stats}
