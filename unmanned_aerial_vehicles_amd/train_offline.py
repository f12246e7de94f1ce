"""Offline trainer CLI — the role of `src/px4/train_gp_offline.py` (same arguments and artefacts):

    python -m unmanned_aerial_vehicles_amd.train_offline --data_dir gp_datasets --output_dir gp_models

globs the CSVs (sorted — the reference's unsorted glob makes pooled runs order-dependent), applies the
row filters of `train_gp_offline.py:60-65`, fits `SimpleQuadrotorGP(max_data_points=10000)` on the GPU,
pickles `{'gp_model', 'training_count', 'data_points_used', 'timestamp', 'is_trained'}`
(`train_gp_offline.py:188-194`), refreshes the `gp_model_latest.pkl` symlink (`:206-212`) and runs the hover
smoke prediction (`:226-232`).
"""
from __future__ import annotations

import argparse
import glob
import os
import pickle
from datetime import datetime

import numpy as np

from .data import load_csv_rows
from .simple_gp import SimpleQuadrotorGP


def main(argv=None):
    ap = argparse.ArgumentParser(description="Train GP from collected flight data (MI355X)")
    ap.add_argument("--data_dir", type=str, default="gp_datasets", help="Directory with CSV flight data")
    ap.add_argument("--output_dir", type=str, default="gp_models", help="Directory to save trained models")
    ap.add_argument("--model_name", type=str, default=None, help="Custom model name")
    ap.add_argument("--pattern", type=str, default="*.csv", help="CSV file pattern to match")
    ap.add_argument("--max_data_points", type=int, default=10000)
    args = ap.parse_args(argv)

    data_dir = os.path.expanduser(args.data_dir)
    csv_files = sorted(f for f in glob.glob(os.path.join(data_dir, args.pattern)) if not f.endswith("_metrics.csv"))
    if not csv_files:
        print(f"No flight data found in {data_dir} (pattern {args.pattern})")
        return 1
    gp = SimpleQuadrotorGP(max_data_points=args.max_data_points)
    total = 0
    for f in csv_files:
        n = load_csv_rows(gp, f)
        total += n
        print(f"  {os.path.basename(f)}: {n} rows")
    print(f"Loaded {total} rows from {len(csv_files)} files; training on the last {len(gp.X_train)}")
    if total < 30:
        print(f"Insufficient training data: {total} < 30")
        return 1
    gp.train_gp()
    if not gp.is_trained:
        print("GP training failed")
        return 1
    out_dir = os.path.expanduser(args.output_dir)
    os.makedirs(out_dir, exist_ok=True)
    name = args.model_name or f"gp_model_{datetime.now().strftime('%Y%m%d_%H%M%S')}"
    path = os.path.join(out_dir, f"{name}.pkl")
    with open(path, "wb") as fh:
        pickle.dump({"gp_model": gp.gp_model, "training_count": gp.training_count,
                     "data_points_used": len(gp.X_train), "timestamp": datetime.now().isoformat(),
                     "is_trained": gp.is_trained}, fh)
    latest = os.path.join(out_dir, "gp_model_latest.pkl")
    try:
        if os.path.lexists(latest):
            os.remove(latest)
        os.symlink(os.path.basename(path), latest)
    except OSError as e:
        print(f"Could not create symlink: {e}")
    residual, _ = gp.predict_residual(np.array([0, 0, -3, 0, 0, 0.0]), np.zeros(4))
    print(f"Model saved: {path}  kernel: {gp.gp_model.kernel_}  "
          f"LML: {gp.gp_model.log_marginal_likelihood_value_:.6f}  hover residual norm: {np.linalg.norm(residual):.4f}")
    return 0


if __name__ == "__main__":
    raise SystemExit(main())
