"""Offline evaluation table: nominal vs GP-corrected residual error.

Restates `src/px4/evaluate_gp_offline.py:163-365` around ONE batched `gp.predict(X_feat)`:
global and acceleration-only MSE/RMSE/improvement, fractions of samples improved, and the
per-component table (mse, rmse, improvement %, R^2 with the true residual as target, fraction
better) written by the reference to `<csv>_metrics.csv`.
"""
from __future__ import annotations

import numpy as np

COMPONENTS = ["dx", "dy", "dz", "dvx", "dvy", "dvz"]
COLUMNS = ["mse_nom", "mse_gp", "rmse_nom", "rmse_gp", "improvement_%", "r2_nom", "r2_gp", "frac_better"]


def r2_score(y_true, y_pred):
    """evaluate_gp_offline.py:146-157 (NaN when the target has no variance)."""
    ss_res = np.sum((y_true - y_pred) ** 2)
    ss_tot = np.sum((y_true - np.mean(y_true)) ** 2)
    return np.nan if ss_tot <= 1e-12 else 1.0 - ss_res / ss_tot


def _block(err_nom, err_gp):
    se_nom, se_gp = np.sum(err_nom ** 2, axis=1), np.sum(err_gp ** 2, axis=1)
    mse_nom, mse_gp = np.mean(se_nom), np.mean(se_gp)
    return {"mse_nom": mse_nom, "mse_gp": mse_gp, "rmse_nom": np.sqrt(mse_nom), "rmse_gp": np.sqrt(mse_gp),
            "improvement_%": (mse_nom - mse_gp) / max(mse_nom, 1e-12) * 100.0}, se_nom, se_gp


def f_nominal(x, u):
    """Double-integrator nominal model (evaluate_gp_offline.py:49-69): x = [p, v] (6,), u = [a_cmd, yaw_rate_cmd] (4,)
    -> xdot = [v, a_cmd] (6,)."""
    xdot = np.zeros(6, dtype=float)
    xdot[0:3] = x[3:6]
    xdot[3:6] = u[0:3]
    return xdot


def load_dataset(csv_path):
    """(X_feat (N,10), R_true (N,6), X_state (N,6), U_ctrl (N,4)) of a dataset CSV, rows with NaN dropped - the arrays
    `load_dataset` of evaluate_gp_offline.py:105-141 returns behind its DataFrame."""
    from .data import read_csv
    X, Y = read_csv(csv_path)
    ok = ~(np.isnan(X).any(axis=1) | np.isnan(Y).any(axis=1))
    X, Y = X[ok], Y[ok]
    return X, Y, X[:, 0:6].copy(), X[:, 6:10].copy()


def evaluate_gp(gp, X_feat, R_true, X_state=None, U_ctrl=None, save_prefix=None):
    """evaluate_gp_offline.py:163-365 with its signature: gp is anything with `.predict(X) -> (N, 6)`; the nominal derivative
    `f_nominal(X_state[i], U_ctrl[i])` is formed for every row, the true derivative reconstructed as nominal + R_true and the
    errors taken on the derivatives, exactly as the reference does (so the table carries its roundings, not just its
    formulas); X_state / U_ctrl default to the state and control columns of X_feat, which is what the reference's loader
    passes.  ONE batched predict.  save_prefix (a path without suffix): writes `<prefix>_metrics.csv` and
    `<prefix>_metrics.tex` like `:322-345`."""
    X_feat = np.asarray(X_feat, dtype=float)
    R_true = np.asarray(R_true, dtype=float)
    N = X_feat.shape[0]
    X_state = X_feat[:, 0:6] if X_state is None else np.asarray(X_state, dtype=float)
    U_ctrl = X_feat[:, 6:10] if U_ctrl is None else np.asarray(U_ctrl, dtype=float)
    xdot_nom = np.zeros((N, 6), dtype=float)
    xdot_nom[:, 0:3] = X_state[:, 3:6]          # f_nominal, row by row in the reference
    xdot_nom[:, 3:6] = U_ctrl[:, 0:3]
    xdot_true = xdot_nom + R_true
    R_pred = np.asarray(gp.predict(X_feat))
    if R_pred.ndim == 1:
        if R_pred.shape[0] != 6:
            raise RuntimeError(f"GP predicted shape {R_pred.shape}, expected (N, 6) or (6,).")
        R_pred = np.tile(R_pred, (N, 1))
    if R_pred.shape[1] > 6:
        R_pred = R_pred[:, :6]
    elif R_pred.shape[1] < 6:
        R_pred = np.hstack([R_pred, np.zeros((N, 6 - R_pred.shape[1]))])
    xdot_gp = xdot_nom + R_pred
    err_nom, err_gp = xdot_true - xdot_nom, xdot_true - xdot_gp
    glob, se_nom, se_gp = _block(err_nom, err_gp)
    acc, _, _ = _block(err_nom[:, 3:6], err_gp[:, 3:6])
    imp = se_nom - se_gp
    fb, fw = float(np.mean(imp > 0.0)), float(np.mean(imp < 0.0))
    table = np.zeros((6, len(COLUMNS)))
    for j in range(6):
        en, eg = err_nom[:, j], err_gp[:, j]
        mn, mg = np.mean(en ** 2), np.mean(eg ** 2)
        table[j] = [mn, mg, np.sqrt(mn), np.sqrt(mg), (mn - mg) / max(mn, 1e-12) * 100.0,
                    r2_score(R_true[:, j], np.zeros(N)), r2_score(R_true[:, j], R_pred[:, j]),
                    np.mean(en ** 2 > eg ** 2)]
    res = {"global": glob, "acc_only": acc,
           "fractions": {"frac_better": fb, "frac_worse": fw, "frac_equal": 1.0 - fb - fw},
           "per_component": table, "components": COMPONENTS, "columns": COLUMNS, "pred": R_pred}
    if save_prefix is not None:
        base = str(save_prefix)
        write_metrics_csv(res, base + "_metrics.csv")
        write_metrics_tex(res, base + "_metrics.tex")
    return res


def write_metrics_tex(result, path):
    """The LaTeX table of evaluate_gp_offline.py:338-343 (`%.3e` entries)."""
    cols = ["component"] + COLUMNS
    with open(path, "w") as f:
        f.write("\\begin{tabular}{l" + "r" * len(COLUMNS) + "}\n\\toprule\n")
        f.write(" & ".join(c.replace("_", "\\_").replace("%", "\\%") for c in cols) + " \\\\\n\\midrule\n")
        for name, row in zip(result["components"], result["per_component"]):
            f.write(name + " & " + " & ".join("%.3e" % float(v) for v in row) + " \\\\\n")
        f.write("\\bottomrule\n\\end{tabular}\n")


def write_metrics_csv(result, path):
    """Same schema as gp_datasets/*_metrics.csv (evaluate_gp_offline.py:322-333)."""
    with open(path, "w") as f:
        f.write("component," + ",".join(COLUMNS) + "\n")
        for name, row in zip(result["components"], result["per_component"]):
            f.write(name + "," + ",".join(repr(float(v)) for v in row) + "\n")


# ---------------------------------------------------------------------------------------------------------------------
# `GPModelEvaluator` (src/px4/gp_evaluation.py:54-549): the seeded physical test grid, ONE batched predict per model and the
# summary numbers of `analyze_gp_performance`.  Plotting is out of scope.
# ---------------------------------------------------------------------------------------------------------------------
GRID_COLUMNS = ["x", "y", "z", "vx", "vy", "vz", "ax", "ay", "az", "yaw_rate"]
# draw order and ranges of the uniform block (gp_evaluation.py:163-174), then the hover block's scales (:177-188)
_UNIFORM = [("x", -10, 10), ("y", -10, 10), ("z", -2, 15), ("vx", -5, 5), ("vy", -5, 5), ("vz", -3, 3),
            ("ax", -8, 8), ("ay", -8, 8), ("az", 1, 18), ("yaw_rate", -1, 1)]
_HOVER = [("vx", 0.0, 0.5), ("vy", 0.0, 0.5), ("vz", 0.0, 0.2), ("ax", 0.0, 2.0), ("ay", 0.0, 2.0), ("az", 9.81, 1.0),
          ("yaw_rate", 0.0, 0.3)]


class _Identity:
    scale_ = np.array([1.0])

    def transform(self, X):
        return X

    def inverse_transform(self, X):
        return X


class GPModelEvaluator:
    """Loads either model file of the reference - `{'gp_model': estimator, ...}` (train_gp_offline.py:188-194) or the
    per-output `{'models', 'scalers_input', 'scalers_output'}` dict (gp_evaluation.py:83-104) - and evaluates it on the
    reference's own synthetic grid.  Each model is predicted with ONE batched `predict(X, return_std=True)` (K4 + K5)."""

    def __init__(self, model_path=None, model_data=None):
        self.model_path = model_path
        self.mode, self.gp_models, self.scalers_X, self.scalers_y = "", {}, {}, {}
        self.gp_model, self.n_features, self.training_stats, self.test_data = None, 0, {}, {}
        if model_data is None:
            import pickle
            with open(model_path, "rb") as f:
                model_data = pickle.load(f)
        self.load_model(model_data)

    def load_model(self, model_data):
        if "models" in model_data:
            self.mode = "multi"
            self.gp_models = model_data["models"]
            self.scalers_X = model_data.get("scalers_input") or {k: _Identity() for k in self.gp_models}
            self.scalers_y = model_data.get("scalers_output") or {k: _Identity() for k in self.gp_models}
            self.training_stats = model_data.get("training_stats", {})
            return
        if "gp_model" in model_data:
            g = model_data["gp_model"]
            if not hasattr(g, "predict"):
                raise KeyError("Unsupported gp_model type inside pickle")
            self.mode, self.gp_model = "single", g
            self.training_stats = {k: model_data.get(k) for k in ("training_count", "data_points_used", "timestamp", "is_trained")}
            if hasattr(g, "n_features_in_"):
                self.n_features = g.n_features_in_
            elif hasattr(g, "X_train_"):
                self.n_features = g.X_train_.shape[1]
            else:
                raise ValueError("gp_model has no 'n_features_in_' or 'X_train_'")
            return
        raise KeyError("Model file does not contain 'models' or 'gp_model'")

    @staticmethod
    def generate_physical_test_data(n_samples=1000):
        """gp_evaluation.py:150-207: `n_samples` uniform rows, 100 hover-like rows, a 200-point figure-8, from
        np.random.seed(42) with the reference's order of draws (2 300 rows for the evaluation's n_samples = 2000)."""
        np.random.seed(42)
        data = {name: np.random.uniform(lo, hi, n_samples) for name, lo, hi in _UNIFORM}
        pos = np.random.uniform(-5, 5, (100, 3))
        hover = {"x": pos[:, 0], "y": pos[:, 1], "z": pos[:, 2] + 5.0}
        for name, mu, sd in _HOVER:
            hover[name] = np.random.normal(mu, sd, 100)
        t = np.linspace(0, 10, 200)
        traj = {"x": 3 * np.sin(0.5 * t), "y": 3 * np.sin(t), "z": 5 + 2 * np.sin(0.3 * t),
                "vx": 1.5 * np.cos(0.5 * t), "vy": 3.0 * np.cos(t), "vz": 0.6 * np.cos(0.3 * t)}
        traj["ax"] = -0.75 * np.sin(0.5 * t) + np.random.normal(0, 1, 200)
        traj["ay"] = -3.0 * np.sin(t) + np.random.normal(0, 1, 200)
        traj["az"] = -0.18 * np.sin(0.3 * t) + 9.81 + np.random.normal(0, 0.5, 200)
        traj["yaw_rate"] = np.random.normal(0, 0.2, 200)
        return {k: np.concatenate([data[k], hover[k], traj[k]]) for k in GRID_COLUMNS}

    def generate_generic_test_data(self, n_samples=2000):
        np.random.seed(42)
        X = np.random.uniform(-1, 1, size=(n_samples, self.n_features))
        return {f"feature_{i}": X[:, i] for i in range(self.n_features)}

    def _grid_matrix(self, test_data):
        if all(k in test_data for k in GRID_COLUMNS) and (self.mode == "multi" or self.n_features == 10):
            return np.column_stack([test_data[k] for k in GRID_COLUMNS])
        return np.column_stack([test_data[k] for k in sorted(k for k in test_data if k.startswith("feature_"))])

    @staticmethod
    def _bands(mean, std):
        std = np.abs(std)
        return {"mean": mean, "std": std, "upper": mean + 2 * std, "lower": mean - 2 * std}

    def predict_on_test_data(self, test_data):
        """gp_evaluation.py:222-330.  A multi-output single estimator is flattened row-major, as the reference does."""
        X = self._grid_matrix(test_data)
        if self.mode == "multi":
            out = {}
            for name, model in self.gp_models.items():
                sx, sy = self.scalers_X.get(name, _Identity()), self.scalers_y.get(name, _Identity())
                ys, ss = model.predict(sx.transform(X), return_std=True)
                ys, ss = np.asarray(ys).reshape(-1), np.asarray(ss).reshape(-1)
                y = np.asarray(sy.inverse_transform(ys.reshape(-1, 1))).flatten()
                out[name] = self._bands(y, ss * sy.scale_[0] if hasattr(sy, "scale_") else ss)
            return out
        if self.mode == "single":
            y, s = self.gp_model.predict(X, return_std=True)
            return {"output": self._bands(np.asarray(y).reshape(-1), np.asarray(s).reshape(-1))}
        return {}

    @staticmethod
    def analyze_gp_performance(predictions):
        """The numbers gp_evaluation.py:503-549 prints: per output (mean prediction, average and largest sigma), the pooled
        uncertainty's mean / max / 90th percentile and the shares of the three confidence regions."""
        if not predictions:
            return {}
        per = {name: {"mean_pred": float(np.mean(np.ravel(p["mean"]))), "sigma_avg": float(np.mean(np.ravel(p["std"]))),
                      "sigma_max": float(np.max(np.ravel(p["std"])))} for name, p in predictions.items()}
        unc = np.concatenate([np.ravel(p["std"]) for p in predictions.values()])
        return {"outputs": list(predictions), "per_output": per, "mean_uncertainty": float(np.mean(unc)),
                "max_uncertainty": float(np.max(unc)), "p90_uncertainty": float(np.percentile(unc, 90)),
                "high_confidence": float(np.mean(unc < 0.1)), "medium_confidence": float(np.mean((unc >= 0.1) & (unc < 0.5))),
                "low_confidence": float(np.mean(unc >= 0.5))}

    def run_complete_evaluation(self):
        physical = self.mode == "multi" or (self.mode == "single" and self.n_features == 10)
        self.test_data = self.generate_physical_test_data(2000) if physical else self.generate_generic_test_data(2000)
        predictions = self.predict_on_test_data(self.test_data)
        return {"test_data": self.test_data, "predictions": predictions, "summary": self.analyze_gp_performance(predictions)}
