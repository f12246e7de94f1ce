#!/usr/bin/env python3
"""Round-4 golden fixtures (run ONCE in the build container; the files of the earlier make_golden*.py stay as they are).

Imports scikit-learn 1.7.2 and the reference's own modules from /root/reference (read-only) and freezes numbers:

  c5_ref.npz         BASELINE configs[4] (C5) as SURVEY.md section 8(d) defines it: N = 4096 synthetic rows, D = 9, three
                     single-output GPs (one per target column) with `C(1, fixed) * RBF(ARD l_d = 2.0 (1 + 0.1 d), (0.1, 10))
                     + White(0.1, (1e-5, 10))`, `alpha = 1e-4`, `normalize_y=True`, `optimizer=None`
                     (the per-axis model of src/px4/gp_trainer.py:163-174): scikit-learn's log-marginal likelihood AND its
                     gradient (sklearn/gaussian_process/_gpr.py:537-652, kernels.py:1571-1580) at that theta.
  train_ref.npz      the reference's offline-training workload at N = 4096 (src/px4/train_gp_offline.py:124-140 ->
                     src/px4/simple_gp.py:156-185): `RBF(0.5) + WhiteKernel(0.1)`, `alpha=1e-4`, `normalize_y=True`,
                     `n_restarts_optimizer=1`, `np.random.seed(0)` on the synthetic flight-like rows of
                     `oracle.gp_oracle.synthetic_flight_problem` (D = 10, P = 6): final theta and LML (the optimiser path is
                     not bit-stable: the test compares the final LML only).
  evaluator_ref.npz  the reference's `GPModelEvaluator` (src/px4/gp_evaluation.py:54-549) on a pickle of the KA3 model
                     (the reference's SimpleQuadrotorGP.train_gp() on gp_mpc_data_20251129_170501.csv, np.random.seed(0)):
                     the seeded 2 300-row physical test grid (`generate_physical_test_data`, :150-207), the arrays
                     `predict_on_test_data` returns (:222-330) and the numbers `analyze_gp_performance` prints (:503-549).

    python tests/golden/make_golden_r4.py [c5] [train] [evaluator]

Nothing here is reference source: the fixtures hold seeds, generator parameters and the values the reference computed.
"""
import contextlib
import io
import os
import pickle
import re
import sys
import tempfile
import time
import types
import warnings

import numpy as np

sys.dont_write_bytecode = True
REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
CSV_TRAIN = f"{REF}/gp_datasets/gp_mpc_data_20251129_170501.csv"


def c5_fixture():
    from sklearn.gaussian_process import GaussianProcessRegressor
    from sklearn.gaussian_process.kernels import RBF, ConstantKernel, WhiteKernel
    from oracle.gp_oracle import synthetic_problem

    N = 4096
    X, Y, _ = synthetic_problem(N, 1)
    ls = 2.0 * (1.0 + 0.1 * np.arange(9))
    out = {"N": np.array(N), "length_scale": ls, "noise_level": np.array(0.1), "alpha": np.array(1e-4)}
    lml, grad, theta = [], [], []
    for b in range(3):
        kern = ConstantKernel(1.0, "fixed") * RBF(ls, (0.1, 10.0)) + WhiteKernel(0.1, (1e-5, 1e1))
        g = GaussianProcessRegressor(kernel=kern, alpha=1e-4, normalize_y=True, optimizer=None).fit(X, Y[:, b])
        t0 = time.perf_counter()
        l, gr = g.log_marginal_likelihood(g.kernel_.theta, eval_gradient=True)
        print(f"c5 gp {b}: lml {l:.6f}  ({time.perf_counter() - t0:.1f} s)", flush=True)
        lml.append(l), grad.append(gr), theta.append(g.kernel_.theta.copy())
    out.update(lml=np.array(lml), grad=np.array(grad), theta=np.array(theta))
    np.savez_compressed(os.path.join(HERE, "c5_ref.npz"), **out)


def train_fixture():
    from sklearn.gaussian_process import GaussianProcessRegressor
    from sklearn.gaussian_process.kernels import RBF, WhiteKernel
    from oracle.gp_oracle import synthetic_flight_problem

    N = 4096
    X, Y = synthetic_flight_problem(N)
    np.random.seed(0)
    t0 = time.perf_counter()
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        g = GaussianProcessRegressor(kernel=RBF(0.5) + WhiteKernel(0.1), alpha=1e-4, normalize_y=True,
                                     n_restarts_optimizer=1).fit(X, Y)
    dt = time.perf_counter() - t0
    print(f"train N={N}: {g.kernel_}  lml {g.log_marginal_likelihood_value_:.6f}  ({dt:.0f} s)", flush=True)
    np.savez_compressed(os.path.join(HERE, "train_ref.npz"), N=np.array(N), seed=np.array(0),
                        theta=g.kernel_.theta.copy(), lml=np.array(g.log_marginal_likelihood_value_),
                        kernel=np.array(str(g.kernel_)), cpu_fit_s=np.array(dt),
                        cpu_threads=np.array(len(os.sched_getaffinity(0))))


def evaluator_fixture():
    # gp_evaluation.py imports seaborn (absent here: an ordinary ModuleNotFoundError) and opens figures: a module stub and
    # the non-interactive backend; none of the plotting is part of the fixture
    import matplotlib
    matplotlib.use("Agg")
    sns = types.ModuleType("seaborn")
    sns.set_palette = lambda *a, **k: None
    sns.heatmap = lambda *a, **k: None
    sys.modules.setdefault("seaborn", sns)
    sys.path.insert(0, f"{REF}/src/px4")
    import gp_evaluation    # the reference module
    import simple_gp        # the reference module

    arr = np.loadtxt(CSV_TRAIN, delimiter=",", skiprows=1)
    X10, Y6 = arr[:, :10].copy(), arr[:, 10:16].copy()
    np.random.seed(0)
    with contextlib.redirect_stdout(io.StringIO()):
        sgp = simple_gp.SimpleQuadrotorGP(max_data_points=10000)
        for xi, yi in zip(X10, Y6):
            sgp.X_train.append(xi)
            sgp.Y_train.append(yi)
        sgp.train_gp()
    gm = sgp.gp_model
    out = {"theta": gm.kernel_.theta.copy(), "lml": np.array(gm.log_marginal_likelihood_value_)}
    with tempfile.TemporaryDirectory() as tmp:
        path = os.path.join(tmp, "model.pkl")
        with open(path, "wb") as f:
            pickle.dump({"gp_model": gm, "training_count": 1000, "data_points_used": 1000,
                         "timestamp": "20251129_170501", "is_trained": True}, f)
        buf = io.StringIO()
        with contextlib.redirect_stdout(buf):
            ev = gp_evaluation.GPModelEvaluator(path)
            grid = ev.generate_physical_test_data(n_samples=2000)
            pred = ev.predict_on_test_data(grid)
            ev.analyze_gp_performance(pred)
        text = buf.getvalue()
    cols = ["x", "y", "z", "vx", "vy", "vz", "ax", "ay", "az", "yaw_rate"]
    out["mode"] = np.array(ev.mode)
    out["n_features"] = np.array(ev.n_features)
    out["grid"] = np.column_stack([grid[c] for c in cols])
    out["grid_columns"] = np.array(cols)
    out["pred_names"] = np.array(list(pred.keys()))
    for name, p in pred.items():
        for k in ("mean", "std", "upper", "lower"):
            out[f"pred_{name}_{k}"] = np.asarray(p[k])

    # the numbers analyze_gp_performance printed (4 decimals / 0.1 %)
    def num(pattern):
        m = re.search(pattern, text)
        assert m, pattern
        return float(m.group(1))
    out["printed_mean_uncertainty"] = np.array(num(r"Mean uncertainty:\s+([-+0-9.eE]+)"))
    out["printed_max_uncertainty"] = np.array(num(r"Max  uncertainty:\s+([-+0-9.eE]+)"))
    out["printed_p90_uncertainty"] = np.array(num(r"90th percentile:\s+([-+0-9.eE]+)"))
    out["printed_high_pct"] = np.array(num(r"High \(.*?\):\s+([0-9.]+)%"))
    out["printed_medium_pct"] = np.array(num(r"Medium \(.*?\):\s+([0-9.]+)%"))
    out["printed_low_pct"] = np.array(num(r"Low \(.*?\):\s+([0-9.]+)%"))
    m = re.search(r"output\s*: μ=([-+0-9.]+), σ_avg=([0-9.]+), σ_max=([0-9.]+)", text)
    assert m, text
    out["printed_output_stats"] = np.array([float(m.group(i)) for i in (1, 2, 3)])
    np.savez_compressed(os.path.join(HERE, "evaluator_ref.npz"), **out)
    print("evaluator:", ev.mode, out["grid"].shape, {k: np.asarray(v["mean"]).shape for k, v in pred.items()},
          "printed:", out["printed_output_stats"], float(out["printed_mean_uncertainty"]))


if __name__ == "__main__":
    which = sys.argv[1:] or ["c5", "train", "evaluator"]
    if "evaluator" in which:
        evaluator_fixture()
    if "c5" in which:
        c5_fixture()
    if "train" in which:
        train_fixture()
