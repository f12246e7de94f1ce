"""Offline evaluation CLI — the role of `src/px4/evaluate_gp_offline.py:371-406`:

    python -m unmanned_aerial_vehicles_amd.evaluate_offline --model-path gp_models/gp_model_latest.pkl \\
        --data-path gp_datasets/gp_mpc_data_20251124_225535.csv

loads the pickle (dict with 'gp_model', or a bare estimator; scikit-learn estimators are moved to the
GPU), runs ONE batched predict over the CSV and writes `<csv>_metrics.csv` with the reference's schema.
"""
from __future__ import annotations

import argparse
import os
import pickle

from .evaluate import evaluate_gp, load_dataset
from .gpr import GaussianProcessRegressor


def load_gp_model(path):
    with open(path, "rb") as f:
        obj = pickle.load(f)
    gp = obj["gp_model"] if isinstance(obj, dict) and "gp_model" in obj else obj
    if not isinstance(gp, GaussianProcessRegressor) and hasattr(gp, "L_") and hasattr(gp, "kernel_"):
        gp = GaussianProcessRegressor.from_sklearn(gp)
    if not hasattr(gp, "predict"):
        raise RuntimeError(f"Loaded object of type {type(gp)} has no .predict() method.")
    return gp


def main(argv=None):
    ap = argparse.ArgumentParser(description="Evaluate a trained GP against a flight CSV (MI355X)")
    ap.add_argument("--model-path", required=True)
    ap.add_argument("--data-path", required=True)
    args = ap.parse_args(argv)
    gp = load_gp_model(args.model_path)
    X, Y, X_state, U_ctrl = load_dataset(args.data_path)
    prefix = os.path.splitext(args.data_path)[0]          # <csv stem>_metrics.csv / .tex, as evaluate_gp_offline.py:395-399
    res = evaluate_gp(gp, X, Y, X_state, U_ctrl, save_prefix=prefix)
    out = prefix + "_metrics.csv"
    g = res["global"]
    print(f"N={len(X)}  MSE nominal {g['mse_nom']:.4e} -> GP {g['mse_gp']:.4e} ({g['improvement_%']:.2f} %)  "
          f"frac better {res['fractions']['frac_better']:.3f}  -> {out}")
    return 0


if __name__ == "__main__":
    raise SystemExit(main())
