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


def evaluate_gp(gp, X_feat, R_true):
    """gp: anything with `.predict(X) -> (N, 6)`.  The nominal model predicts zero residual."""
    X_feat = np.asarray(X_feat, dtype=float)
    R_true = np.asarray(R_true, dtype=float)
    N = X_feat.shape[0]
    R_pred = np.asarray(gp.predict(X_feat))
    if R_pred.ndim == 1:
        if R_pred.shape[0] != 6:
            raise RuntimeError(f"GP predicted shape {R_pred.shape}, expected (N, 6) or (6,).")
        R_pred = np.tile(R_pred, (N, 1))
    if R_pred.shape[1] > 6:
        R_pred = R_pred[:, :6]
    elif R_pred.shape[1] < 6:
        R_pred = np.hstack([R_pred, np.zeros((N, 6 - R_pred.shape[1]))])
    err_nom, err_gp = R_true, R_true - R_pred
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
    return {"global": glob, "acc_only": acc,
            "fractions": {"frac_better": fb, "frac_worse": fw, "frac_equal": 1.0 - fb - fw},
            "per_component": table, "components": COMPONENTS, "columns": COLUMNS, "pred": R_pred}


def write_metrics_csv(result, path):
    """Same schema as gp_datasets/*_metrics.csv (evaluate_gp_offline.py:322-333)."""
    with open(path, "w") as f:
        f.write("component," + ",".join(COLUMNS) + "\n")
        for name, row in zip(result["components"], result["per_component"]):
            f.write(name + "," + ",".join(repr(float(v)) for v in row) + "\n")
