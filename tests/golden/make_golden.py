#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ (run ONCE in the build container).

Imports scikit-learn 1.7.2 (the library the reference delegates its GP arithmetic to) and
the reference's own Python modules from /root/reference, runs them on the reference's
flight CSVs, and freezes inputs + outputs as small .npz files.  The reference tree never
travels to the GPU box; only these data files do.  Nothing here is reference source: the
fixtures hold numbers only (CSV columns and the values the reference computed from them).

    python tests/golden/make_golden.py

Known answers follow SURVEY.md §8(c) KA1..KA6.
"""
import os
import sys
import types
import warnings

import numpy as np

sys.dont_write_bytecode = True
REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))
CSV_TRAIN = f"{REF}/gp_datasets/gp_mpc_data_20251129_170501.csv"
CSV_QUERY = f"{REF}/gp_datasets/gp_mpc_data_20251129_221039.csv"
CSV_METRICS_DATA = f"{REF}/gp_datasets/gp_mpc_data_20251124_225535.csv"

from sklearn.gaussian_process import GaussianProcessRegressor  # noqa: E402
from sklearn.gaussian_process.kernels import RBF, ConstantKernel, WhiteKernel  # noqa: E402


def load_csv(path):
    arr = np.loadtxt(path, delimiter=",", skiprows=1)
    return arr[:, :10].copy(), arr[:, 10:16].copy()


def model_record(gpr, Xq, prefix, out, rows=(0, 1, 7, 100, 333, 512, 998, 999)):
    L = gpr.L_
    out[f"{prefix}_theta"] = gpr.kernel_.theta.copy()
    out[f"{prefix}_alpha"] = gpr.alpha_.copy()
    out[f"{prefix}_Ldiag"] = np.diag(L).copy()
    out[f"{prefix}_Lfro"] = np.array(np.linalg.norm(np.tril(L)))
    out[f"{prefix}_Lrows_idx"] = np.array(rows)
    out[f"{prefix}_Lrows"] = np.tril(L)[list(rows), :].copy()
    out[f"{prefix}_ymean"] = np.asarray(gpr._y_train_mean, dtype=float).reshape(-1).copy()
    out[f"{prefix}_ystd"] = np.asarray(gpr._y_train_std, dtype=float).reshape(-1).copy()
    mean, std = gpr.predict(Xq, return_std=True)
    out[f"{prefix}_mean"] = mean
    out[f"{prefix}_std"] = std
    out[f"{prefix}_lml"] = np.array(gpr.log_marginal_likelihood(gpr.kernel_.theta))


def main():
    warnings.simplefilter("ignore")
    X10, Y6 = load_csv(CSV_TRAIN)
    Xq_other, _ = load_csv(CSV_QUERY)
    assert X10.shape == (1000, 10)
    # 64 fixed query rows: 32 training rows + 32 rows of another flight
    q_train_idx = np.arange(0, 1000, 1000 // 32)[:32]
    Xq10 = np.vstack([X10[q_train_idx], Xq_other[:32]])
    data = {"X10": X10, "Y6": Y6, "Xq10": Xq10, "q_train_idx": q_train_idx}
    np.savez_compressed(os.path.join(HERE, "csv_170501.npz"), **data)

    out = {}
    # ---- KA1: sklearn fixed theta, D=10, P=6 ------------------------------------------
    k = RBF(0.5) + WhiteKernel(0.1)
    g = GaussianProcessRegressor(kernel=k, alpha=1e-4, normalize_y=True, optimizer=None).fit(X10, Y6)
    model_record(g, Xq10, "ka1", out)
    # ---- KA2: D=9, P=3 + gradient ------------------------------------------------------
    X9, Y3, Xq9 = X10[:, :9], Y6[:, 3:6], Xq10[:, :9]
    g = GaussianProcessRegressor(kernel=RBF(0.5) + WhiteKernel(0.1), alpha=1e-4, normalize_y=True,
                                 optimizer=None).fit(X9, Y3)
    model_record(g, Xq9, "ka2", out)
    lml, grad = g.log_marginal_likelihood(g.kernel_.theta, eval_gradient=True)
    out["ka2_grad"] = grad
    K = g.kernel_(X9)
    K[np.diag_indices_from(K)] += 1e-4
    out["ka2_cond"] = np.array(np.linalg.cond(K))
    # Gram samples for the K1 kernel: a few full rows of kernel_(X) (incl. white noise on diag)
    out["ka2_Krows_idx"] = np.array([0, 1, 500, 999])
    out["ka2_Krows"] = g.kernel_(X9)[[0, 1, 500, 999], :]
    # a second fixed theta close to the optimum (better conditioned, shorter length-scale)
    g = GaussianProcessRegressor(kernel=RBF(0.114) + WhiteKernel(0.35), alpha=1e-4, normalize_y=True,
                                 optimizer=None).fit(X9, Y3)
    model_record(g, Xq9, "ka2b", out)
    lml, grad = g.log_marginal_likelihood(g.kernel_.theta, eval_gradient=True)
    out["ka2b_grad"] = grad

    # ---- KA3: the reference's own SimpleQuadrotorGP.train_gp() --------------------------
    sys.path.insert(0, f"{REF}/src/px4")
    import simple_gp  # the reference module (read-only import)

    np.random.seed(0)
    sgp = simple_gp.SimpleQuadrotorGP(max_data_points=10000)
    for xi, yi in zip(X10, Y6):
        # same acceptance rule as src/px4/train_gp_offline.py:60-65
        if np.isfinite(xi).all() and np.isfinite(yi).all() and np.linalg.norm(yi) < 5.0:
            sgp.X_train.append(xi)
            sgp.Y_train.append(yi)
    out["ka3_rows_kept"] = np.array(len(sgp.X_train))
    sgp.train_gp()
    assert sgp.is_trained
    gm = sgp.gp_model
    out["ka3_theta"] = gm.kernel_.theta.copy()
    out["ka3_lml"] = np.array(gm.log_marginal_likelihood_value_)
    out["ka3_ymean"] = gm._y_train_mean.copy()
    out["ka3_ystd"] = gm._y_train_std.copy()
    m, v = sgp.predict_residual(X10[24, :6], X10[24, 6:])
    out["ka3_pred_mean"], out["ka3_pred_var"] = m, v
    # batched means at the 64 queries through the trained reference model
    mean, std = gm.predict(Xq10, return_std=True)
    out["ka3_mean"], out["ka3_std"] = mean, std
    # horizon-batched residual builder (src/px4/mpc.py:1475-1511): N=25 serial predicts,
    # D[3:6,k] = 0.1 * mean[3:6] / dt ; warm start = 26 consecutive flight rows
    dt, gain, Nh = 0.1, 0.1, 25
    Xg = Xq_other[100:100 + Nh + 1, :6].T.copy()
    Ug = Xq_other[100:100 + Nh, 6:10].T.copy()
    Dm = np.zeros((6, Nh))
    for kk in range(Nh):
        mk, _ = sgp.predict_residual(Xg[:, kk], Ug[:, kk])
        Dm[3:6, kk] = gain * (np.asarray(mk).reshape(-1) / dt)[3:6]
    out["ka3_hor_X"], out["ka3_hor_U"], out["ka3_hor_D"] = Xg, Ug, Dm
    out["ka3_hor_dt"] = np.array(dt)
    # uncertainty gate input (simple_gp.py:203-207)
    out["ka3_uncertainty"] = np.array(sgp.get_uncertainty(X10[24, :6], X10[24, 6:]))

    # ---- KA4: sklearn optimised, D=9, P=3, no restarts ----------------------------------
    g = GaussianProcessRegressor(kernel=RBF(0.5) + WhiteKernel(0.1), alpha=1e-4, normalize_y=True,
                                 n_restarts_optimizer=0).fit(X9, Y3)
    out["ka4_theta"] = g.kernel_.theta.copy()
    out["ka4_lml"] = np.array(g.log_marginal_likelihood_value_)

    # ---- KA5: the ROS-package GaussianProcess -------------------------------------------
    for name in ("rclpy", "rclpy.node", "std_msgs", "std_msgs.msg"):
        sys.modules.setdefault(name, types.ModuleType(name))

    class _Node:  # stand-in for rclpy.node.Node (absent here): logger + no-op ROS plumbing
        def __init__(self, *a, **k):
            pass

        def get_logger(self):
            class _L:
                def __getattr__(self, _):
                    return lambda *a, **k: None
            return _L()

        def create_publisher(self, *a, **k):
            return None

        create_subscription = create_timer = create_publisher

    sys.modules["rclpy.node"].Node = _Node
    sys.modules["std_msgs.msg"].Float64MultiArray = type("Float64MultiArray", (), {})
    sys.path.insert(0, f"{REF}/quadrotor_gp_mpc/quadrotor_gp_mpc")
    import gaussian_process as pkg_gp  # the reference module

    pg = pkg_gp.GaussianProcess(input_dim=9, output_dim=3)
    pg.add_training_data(X9, Y3)
    pg.fit()
    out["ka5_lml"] = np.array(pg.log_marginal_likelihood())
    pm, pv = pg.predict(Xq9)
    out["ka5_mean"], out["ka5_var"] = pm, pv
    out["ka5_alpha"] = pg.alpha.copy()
    out["ka5_Ldiag"] = np.diag(pg.L).copy()

    # ---- KA6: per-output ARD model (gp_trainer.py:163-174), fixed theta ------------------
    y1 = Y6[:, 3]
    kern = (ConstantKernel(1.0, constant_value_bounds="fixed") * RBF([1.0] * 9, (0.1, 10.0))
            + WhiteKernel(0.01, (1e-5, 1e1)))
    g = GaussianProcessRegressor(kernel=kern, alpha=1e-6, normalize_y=False, optimizer=None).fit(X9, y1)
    lml, grad = g.log_marginal_likelihood(g.kernel_.theta, eval_gradient=True)
    out["ka6_lml"], out["ka6_grad"] = np.array(lml), grad
    out["ka6_theta"] = g.kernel_.theta.copy()
    mean, std = g.predict(Xq9, return_std=True)
    out["ka6_mean"], out["ka6_std"] = mean, std
    # same with non-trivial ARD length-scales
    ls = 0.3 * (1.0 + 0.25 * np.arange(9))
    kern = (ConstantKernel(1.0, constant_value_bounds="fixed") * RBF(ls, (0.1, 10.0))
            + WhiteKernel(0.05, (1e-5, 1e1)))
    g = GaussianProcessRegressor(kernel=kern, alpha=1e-6, normalize_y=False, optimizer=None).fit(X9, y1)
    lml, grad = g.log_marginal_likelihood(g.kernel_.theta, eval_gradient=True)
    out["ka6b_ls"] = ls
    out["ka6b_lml"], out["ka6b_grad"] = np.array(lml), grad
    mean, std = g.predict(Xq9, return_std=True)
    out["ka6b_mean"], out["ka6b_std"] = mean, std
    out["ka6b_alpha"] = g.alpha_.copy()

    # ---- C2-sized synthetic known answer (N=4096, M=1024, fp64) ---------------------------
    rng = np.random.default_rng(0)
    N, M, D, P = 4096, 1024, 9, 3
    Xs = rng.standard_normal((N, D))
    W = rng.standard_normal((D, P))
    Ys = np.sin(Xs @ W) + 0.1 * rng.standard_normal((N, P))
    Xqs = np.random.default_rng(1).standard_normal((M, D))
    g = GaussianProcessRegressor(kernel=RBF(2.0) + WhiteKernel(0.1), alpha=1e-4, normalize_y=True,
                                 optimizer=None).fit(Xs, Ys)
    mean, std = g.predict(Xqs, return_std=True)
    out["c2_mean"], out["c2_std"] = mean, std
    out["c2_lml"] = np.array(g.log_marginal_likelihood(g.kernel_.theta))
    out["c2_Ldiag"] = np.diag(g.L_).copy()
    out["c2_alpha"] = g.alpha_.copy()

    np.savez_compressed(os.path.join(HERE, "known_answers.npz"), **out)

    # ---- offline evaluation table: the reference's own evaluate_gp() --------------------
    # (src/px4/evaluate_gp_offline.py:163-365) on the first 400 rows of the CSV its published
    # metrics file belongs to, with the KA1 model as the predictor.
    import contextlib
    import io
    import evaluate_gp_offline as ref_eval  # the reference module

    Xe, Ye = load_csv(CSV_METRICS_DATA)
    Xe, Ye = Xe[:400], Ye[:400]
    g = GaussianProcessRegressor(kernel=RBF(0.5) + WhiteKernel(0.1), alpha=1e-4, normalize_y=True,
                                 optimizer=None).fit(X10, Y6)
    with contextlib.redirect_stdout(io.StringIO()):
        res = ref_eval.evaluate_gp(g, Xe, Ye, Xe[:, :6], Xe[:, 6:10], save_prefix=None)
    df = res["per_component"]
    cols = ["mse_nom", "mse_gp", "rmse_nom", "rmse_gp", "improvement_%", "r2_nom", "r2_gp", "frac_better"]
    np.savez_compressed(
        os.path.join(HERE, "eval_table.npz"), X=Xe, Y=Ye, pred=g.predict(Xe),
        table=df[cols].to_numpy(dtype=float), columns=np.array(cols), components=df["component"].to_numpy(str),
        global_=np.array([res["global"][k] for k in ("mse_nom", "mse_gp", "rmse_nom", "rmse_gp", "improvement_%")]),
        acc_only=np.array([res["acc_only"][k] for k in ("mse_nom", "mse_gp", "rmse_nom", "rmse_gp", "improvement_%")]),
        fractions=np.array([res["fractions"][k] for k in ("frac_better", "frac_worse", "frac_equal")]))
    for kname in sorted(out):
        v = out[kname]
        print(f"{kname:20s} shape={getattr(v, 'shape', ())}")
    print("KA1 lml", out["ka1_lml"], " KA2 lml", out["ka2_lml"], "grad", out["ka2_grad"])
    print("KA3 theta", out["ka3_theta"], "lml", out["ka3_lml"])
    print("KA4 theta", out["ka4_theta"], "lml", out["ka4_lml"])
    print("KA5 lml", out["ka5_lml"], " KA6 lml", out["ka6_lml"])


if __name__ == "__main__":
    main()
