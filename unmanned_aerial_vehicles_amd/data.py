"""CSV <-> training-set pipeline of the reference's offline tools.

`load_csv_rows` applies the acceptance rule of `src/px4/train_gp_offline.py:22-76` (all 16
columns present, every value finite, ||residual||_2 < 5); `load_dataset_dir` walks a directory in
*sorted* order (the reference globs unsorted, `train_gp_offline.py:95`, which makes pooled runs
order-dependent); `save_dataset_csv` writes the 16-column `%.18e` file of
`src/px4/simple_gp.py:75-115`.
"""
from __future__ import annotations

import glob
import os

import numpy as np

INPUT_COLS = ["x", "y", "z", "vx", "vy", "vz", "ax", "ay", "az", "yaw_rate"]
OUTPUT_COLS = ["res_dx", "res_dy", "res_dz", "res_dvx", "res_dvy", "res_dvz"]
HEADER = ",".join(INPUT_COLS + OUTPUT_COLS)


def read_csv(path):
    """Returns (X (n,10), Y (n,6)) for a dataset CSV, columns located by header name."""
    with open(path, "r") as f:
        header = f.readline().strip().split(",")
    missing = [c for c in INPUT_COLS + OUTPUT_COLS if c not in header]
    if missing:
        raise ValueError(f"{path}: missing columns {missing}")
    arr = np.atleast_2d(np.genfromtxt(path, delimiter=",", skip_header=1))
    if arr.size == 0:
        return np.empty((0, 10)), np.empty((0, 6))
    idx_in = [header.index(c) for c in INPUT_COLS]
    idx_out = [header.index(c) for c in OUTPUT_COLS]
    return arr[:, idx_in], arr[:, idx_out]


def filter_rows(X, Y, max_residual_norm=5.0):
    """Keep a row iff all 16 values are finite and ||y||_2 < 5 (train_gp_offline.py:60-65)."""
    ok = np.isfinite(X).all(axis=1) & np.isfinite(Y).all(axis=1)
    with np.errstate(invalid="ignore"):
        ok &= np.linalg.norm(np.where(np.isfinite(Y), Y, 0.0), axis=1) < max_residual_norm
    return X[ok], Y[ok]


def load_csv_rows(gp, csv_path):
    """Append the accepted rows of one CSV to `gp.X_train` / `gp.Y_train`; returns the count
    (the role of `load_csv_data_simple`, train_gp_offline.py:22-76)."""
    try:
        X, Y = read_csv(csv_path)
    except Exception as e:  # noqa: BLE001 - reference returns 0 on any failure
        print(f"Failed to load CSV: {e}")
        return 0
    X, Y = filter_rows(X, Y)
    for xi, yi in zip(X, Y):
        gp.X_train.append(xi)
        gp.Y_train.append(yi)
    return len(X)


def load_dataset_dir(gp, data_dir, pattern="*.csv"):
    """All CSVs of a directory in sorted order (metrics tables are skipped)."""
    total = 0
    for path in sorted(glob.glob(os.path.join(data_dir, pattern))):
        if path.endswith("_metrics.csv"):
            continue
        total += load_csv_rows(gp, path)
    return total


def save_dataset_csv(csv_path, X, Y, include_header=True, overwrite=True):
    data = np.hstack([X, Y])
    os.makedirs(os.path.dirname(csv_path) or ".", exist_ok=True)
    if overwrite or not os.path.exists(csv_path):
        np.savetxt(csv_path, data, delimiter=",", header=HEADER if include_header else "", comments="")
    else:
        with open(csv_path, "ab") as f:
            np.savetxt(f, data, delimiter=",")
