#!/usr/bin/env python3
"""Round-2 golden fixtures (run ONCE in the build container; the round-1 files of make_golden.py stay as they are).

Imports the reference's own modules from /root/reference (read-only) together with scikit-learn 1.7.2 and freezes
inputs + outputs as numbers:

  trainer_ref.npz   `GPTrainer.train_gp_models` (src/px4/gp_trainer.py:121-205) on the first 300 rows of
                    gp_datasets/gp_mpc_data_20251129_170501.csv with np.random.seed(0): per-output theta, LML,
                    mse / rmse / r2, the numeric content of the pickle `save_models` writes (gp_trainer.py:214-221:
                    X_train_, alpha_, L_, theta, scaler mean_ / scale_), and `PreTrainedGP.predict_residual`
                    (src/px4/pretrained_gp.py:52-98) of that pickle at 8 rows.
  loader_ref.npz    `load_csv_data_simple` (src/px4/train_gp_offline.py:22-76) on a CSV with injected NaN / inf /
                    ||y|| >= 5 rows: the rows it kept; `SimpleQuadrotorGP.add_training_data`
                    (src/px4/simple_gp.py:118-140) on transitions with |v| > 5, |a| > 3 and ||residual|| > 2 cases:
                    kept indices and the (X, Y) rows it stored.

    python tests/golden/make_golden_r2.py

Nothing here is reference source: the fixtures hold CSV columns, seeds and the values the reference computed.
"""
import contextlib
import io
import os
import sys
import tempfile
import warnings

import numpy as np

sys.dont_write_bytecode = True
REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))
CSV_TRAIN = f"{REF}/gp_datasets/gp_mpc_data_20251129_170501.csv"
CSV_QUERY = f"{REF}/gp_datasets/gp_mpc_data_20251129_221039.csv"
NAMES = ["x_residual", "y_residual", "z_residual", "vx_residual", "vy_residual", "vz_residual"]


def load_csv(path):
    arr = np.loadtxt(path, delimiter=",", skiprows=1)
    return arr[:, :10].copy(), arr[:, 10:16].copy()


def trainer_fixture():
    sys.path.insert(0, f"{REF}/src/px4")
    import gp_trainer       # the reference module
    import pretrained_gp    # the reference module

    X10, Y6 = load_csv(CSV_TRAIN)
    Xo, _ = load_csv(CSV_QUERY)
    X, Y = X10[:300], Y6[:300]
    out = {"X": X, "Y": Y, "seed": np.array(0)}
    with tempfile.TemporaryDirectory() as tmp, contextlib.redirect_stdout(io.StringIO()):
        tr = gp_trainer.GPTrainer(data_dir=tmp, model_dir=tmp)
        np.random.seed(0)
        res = tr.train_gp_models(X, Y)
        path = tr.save_models("ref_model")
        pre = pretrained_gp.PreTrainedGP(path)
        assert pre.is_loaded
        # 8 query rows: 4 training rows of the 300 and 4 rows of another flight
        Xq = np.vstack([X[[0, 17, 150, 299]], Xo[[5, 50, 500, 900]]])
        means, stds = [], []
        for row in Xq:
            m, s = pre.predict_residual(row[:6], row[6:])
            means.append(m)
            stds.append(s)
        unc = pre.get_uncertainty(Xq[0, :6], Xq[0, 6:])
    out["Xq"] = Xq
    out["pred_mean"], out["pred_std"] = np.array(means), np.array(stds)
    out["uncertainty_row0"] = np.array(unc)
    out["names"] = np.array([n for n in NAMES if n in tr.gp_models])
    for n in NAMES:
        if n not in tr.gp_models:
            continue
        g, sx, sy = tr.gp_models[n], tr.scalers_X[n], tr.scalers_y[n]
        out[f"{n}_theta"] = g.kernel_.theta.copy()
        out[f"{n}_lml"] = np.array(g.log_marginal_likelihood_value_)
        out[f"{n}_stats"] = np.array([res[n]["mse"], res[n]["rmse"], res[n]["r2"], res[n]["log_marginal_likelihood"]])
        out[f"{n}_kernel_str"] = np.array(res[n]["kernel"])
        out[f"{n}_X_train"] = g.X_train_.copy()
        out[f"{n}_y_train"] = g.y_train_.copy()
        out[f"{n}_alpha"] = g.alpha_.copy()
        out[f"{n}_L"] = np.tril(g.L_).copy()
        out[f"{n}_sx_mean"], out[f"{n}_sx_scale"] = sx.mean_.copy(), sx.scale_.copy()
        out[f"{n}_sy_mean"], out[f"{n}_sy_scale"] = sy.mean_.copy(), sy.scale_.copy()
    np.savez_compressed(os.path.join(HERE, "trainer_ref.npz"), **out)
    for n in out["names"]:
        print(f"{n:12s} lml {float(out[f'{n}_lml']): .6f}  rmse {out[f'{n}_stats'][1]:.3e}  {out[f'{n}_kernel_str']}")
    print("predict_residual row 0:", out["pred_mean"][0], out["pred_std"][0])


def loader_fixture():
    import types
    sys.path.insert(0, f"{REF}/src/px4")
    import simple_gp  # the reference module
    # train_gp_offline.py imports `px4_offboard.simple_gp` (its ROS package name): an ordinary ImportError here,
    # resolved by registering the module that is already imported under that name
    pkg = types.ModuleType("px4_offboard")
    pkg.simple_gp = simple_gp
    sys.modules["px4_offboard"] = pkg
    sys.modules["px4_offboard.simple_gp"] = simple_gp
    import train_gp_offline as tgo  # the reference module

    X10, Y6 = load_csv(CSV_TRAIN)
    rng = np.random.default_rng(7)
    rows = np.hstack([X10[:120], Y6[:120]]).copy()
    # injected defects: NaN in an input, NaN in an output, +inf, -inf, residual norm just above / exactly at /
    # just below the threshold of 5 (the rule is a strict "<")
    rows[3, 2] = np.nan
    rows[10, 12] = np.nan
    rows[25, 7] = np.inf
    rows[40, 15] = -np.inf
    rows[55, 10:16] = [5.0, 0.0, 0.0, 0.0, 0.0, 0.0]                  # norm == 5: rejected
    rows[56, 10:16] = [3.0, 4.0, 0.0, 0.0, 0.0, 1e-9]                 # just above
    rows[57, 10:16] = [3.0, 3.9999, 0.0, 0.0, 0.0, 0.0]               # just below: kept
    rows[80, 10:16] = 4.0 * np.ones(6)                                # norm 9.8
    rows[81, 10:16] = rng.standard_normal(6) * 10.0
    header = "x,y,z,vx,vy,vz,ax,ay,az,yaw_rate,res_dx,res_dy,res_dz,res_dvx,res_dvy,res_dvz"
    out = {"csv_rows": rows, "csv_header": np.array(header)}
    with tempfile.TemporaryDirectory() as tmp, contextlib.redirect_stdout(io.StringIO()):
        p = os.path.join(tmp, "inj.csv")
        np.savetxt(p, rows, delimiter=",", header=header, comments="")
        gp = simple_gp.SimpleQuadrotorGP(max_data_points=10000)
        n = tgo.load_csv_data_simple(gp, p)
        # a CSV with shuffled columns (columns are looked up by name) and one with a column missing
        perm = rng.permutation(16)
        p2 = os.path.join(tmp, "perm.csv")
        np.savetxt(p2, rows[:, perm], delimiter=",", header=",".join(np.array(header.split(","))[perm]), comments="")
        gp2 = simple_gp.SimpleQuadrotorGP(max_data_points=10000)
        n2 = tgo.load_csv_data_simple(gp2, p2)
        p3 = os.path.join(tmp, "missing.csv")
        np.savetxt(p3, rows[:, :15], delimiter=",", header=",".join(header.split(",")[:15]), comments="")
        gp3 = simple_gp.SimpleQuadrotorGP(max_data_points=10000)
        n3 = tgo.load_csv_data_simple(gp3, p3)
    out["csv_kept_count"] = np.array(n)
    out["csv_kept_X"], out["csv_kept_Y"] = np.array(gp.X_train), np.array(gp.Y_train)
    out["csv_perm"] = perm
    out["csv_perm_kept_count"] = np.array(n2)
    out["csv_perm_kept_X"] = np.array(gp2.X_train)
    out["csv_missing_kept_count"] = np.array(n3)
    print("load_csv_data_simple kept", n, "of", len(rows), "| permuted columns:", n2, "| missing column:", n3)

    # ---- add_training_data (simple_gp.py:118-140): transitions built from consecutive flight rows
    T = 60
    states = X10[200:200 + T, :6].copy()
    controls = X10[200:200 + T, 6:10].copy()
    dts = np.full(T, 0.02)
    nxt = states + 0.02 * np.hstack([states[:, 3:6], controls[:, :3]]) + 0.01 * rng.standard_normal((T, 6))
    states[5, 3:6] = [4.0, 3.0, 0.1]            # |v| = 5.001 > 5: skipped
    states[6, 3:6] = [3.0, 4.0, 0.0]            # |v| = 5 exactly: kept ("> 5" is strict)
    nxt[6] = states[6] + 0.02 * np.hstack([states[6, 3:6], controls[6, :3]])
    controls[9, :3] = [3.0, 0.1, 0.0]           # |a| > 3: skipped
    controls[10, :3] = [0.0, 3.0, 0.0]          # |a| = 3 exactly: kept
    nxt[10] = states[10] + 0.02 * np.hstack([states[10, 3:6], controls[10, :3]])
    nxt[20] += np.array([2.5, 0, 0, 0, 0, 0])   # ||residual|| > 2: skipped
    nxt[21] = states[21] + 0.02 * np.hstack([states[21, 3:6], controls[21, :3]]) + np.array([0, 2.0, 0, 0, 0, 0])  # == 2: kept
    dts[30:40] = 0.1                            # another step size
    nxt[30:40] = states[30:40] + 0.1 * np.hstack([states[30:40, 3:6], controls[30:40, :3]]) + 0.02 * rng.standard_normal((10, 6))
    gp = simple_gp.SimpleQuadrotorGP(max_data_points=10000)
    kept = []
    for i in range(T):
        before = len(gp.X_train)
        gp.add_training_data(states[i], controls[i], nxt[i], dt=float(dts[i]))
        if len(gp.X_train) > before:
            kept.append(i)
    # short state vectors are ignored (simple_gp.py:120-121)
    gp.add_training_data(states[0, :5], controls[0], nxt[0])
    assert len(gp.X_train) == len(kept)
    out["atd_states"], out["atd_controls"], out["atd_next"], out["atd_dt"] = states, controls, nxt, dts
    out["atd_kept"] = np.array(kept)
    out["atd_X"], out["atd_Y"] = np.array(gp.X_train), np.array(gp.Y_train)
    # the deque bound (simple_gp.py:31-32): the oldest rows are evicted
    gp_small = simple_gp.SimpleQuadrotorGP(max_data_points=16)
    for i in range(T):
        gp_small.add_training_data(states[i], controls[i], nxt[i], dt=float(dts[i]))
    out["atd_small_X"] = np.array(gp_small.X_train)
    print("add_training_data kept", len(kept), "of", T, "rows; rejected:", sorted(set(range(T)) - set(kept)))
    np.savez_compressed(os.path.join(HERE, "loader_ref.npz"), **out)


if __name__ == "__main__":
    warnings.simplefilter("ignore")
    which = sys.argv[1:] or ["loader", "trainer"]
    if "loader" in which:
        loader_fixture()
    if "trainer" in which:
        trainer_fixture()
