#!/usr/bin/env python3
"""Round-5 golden fixtures (run ONCE in the build container; the files of the earlier make_golden*.py stay as they are).

Imports the reference's ROS-package GP module from /root/reference (read-only; `rclpy` and `std_msgs` are absent here and
are stood in for by ordinary `sys.modules` entries, exactly as tests/golden/make_golden.py does for KA5) and freezes numbers:

  package_kernel_ref.npz   `quadrotor_gp_mpc/quadrotor_gp_mpc/gaussian_process.py`:
      * `RBFKernel.__call__` (:26-41) and `RBFKernel.gradient` (:43-60) for length_scale 0.7, signal_variance 1.3 on 64 rows
        of gp_mpc_data_20251129_170501.csv against 48 rows of gp_mpc_data_20251129_221039.csv (9 features), and on the 64
        rows against themselves;
      * `GaussianProcess.compute_kernel_matrix` (:158-171) with and without the second argument (noise_variance 0.02);
      * the two topic callbacks (:326-358) driven with `msg.data` stand-ins: 40 training messages through
        `training_data_callback` (one of them with a wrong length, which the reference rejects), `fit()`, then 5 requests
        through `prediction_request_callback` (one with a wrong length) - the stored training set and the `.data` lists
        the reference publishes on /gp/prediction and /gp/uncertainty.

    python tests/golden/make_golden_r5.py

Nothing here is reference source: the fixture holds CSV rows and the values the reference computed from them.
"""
import os
import sys
import types

import numpy as np

sys.dont_write_bytecode = True
REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))
CSV_TRAIN = f"{REF}/gp_datasets/gp_mpc_data_20251129_170501.csv"
CSV_QUERY = f"{REF}/gp_datasets/gp_mpc_data_20251129_221039.csv"


def load_csv(path):
    arr = np.loadtxt(path, delimiter=",", skiprows=1)
    return arr[:, :10].copy(), arr[:, 10:16].copy()


class _Recorder:
    def __init__(self):
        self.sent = []

    def publish(self, msg):
        self.sent.append(list(msg.data))


def import_reference_package_gp():
    for name in ("rclpy", "rclpy.node", "std_msgs", "std_msgs.msg"):
        sys.modules.setdefault(name, types.ModuleType(name))

    class _Log:
        def __getattr__(self, _n):
            return lambda *a, **k: None

    class _Node:  # stand-in for rclpy.node.Node (absent here): logger, recording publishers, no-op subscriptions / timers
        def __init__(self, *a, **k):
            pass

        def get_logger(self):
            return _Log()

        def create_publisher(self, *a, **k):
            return _Recorder()

        def create_subscription(self, *a, **k):
            return None

        create_timer = create_subscription

    sys.modules["rclpy.node"].Node = _Node

    class Float64MultiArray:  # the message type: a bare object with a .data attribute
        def __init__(self):
            self.data = []

    sys.modules["std_msgs.msg"].Float64MultiArray = Float64MultiArray
    sys.path.insert(0, f"{REF}/quadrotor_gp_mpc/quadrotor_gp_mpc")
    import gaussian_process as pkg_gp  # the reference module
    return pkg_gp, Float64MultiArray


def main():
    pkg_gp, Msg = import_reference_package_gp()
    X10, Y6 = load_csv(CSV_TRAIN)
    Xo, _ = load_csv(CSV_QUERY)
    XA = X10[np.arange(0, 1000, 1000 // 64)[:64], :9].copy()
    XB = Xo[:48, :9].copy()
    out = {"XA": XA, "XB": XB, "length_scale": np.array(0.7), "signal_variance": np.array(1.3),
           "noise_variance": np.array(0.02)}
    k = pkg_gp.RBFKernel(length_scale=0.7, signal_variance=1.3)
    out["K_AB"] = k(XA, XB)
    out["K_AA"] = k(XA, XA)
    out["dKdl_AB"], out["dKds_AB"] = k.gradient(XA, XB)
    out["dKdl_AA"], out["dKds_AA"] = k.gradient(XA, XA)

    gp = pkg_gp.GaussianProcess(input_dim=9, output_dim=3)
    gp.kernel.length_scale, gp.kernel.signal_variance, gp.noise_variance = 0.7, 1.3, 0.02
    out["ckm_A"] = gp.compute_kernel_matrix(XA)
    out["ckm_AB"] = gp.compute_kernel_matrix(XA, XB)

    # ---- the topic callbacks ----------------------------------------------------------------------
    gp = pkg_gp.GaussianProcess(input_dim=9, output_dim=3)
    train_msgs = []
    for r in range(40):
        row = np.concatenate([X10[25 * r, :9], Y6[25 * r, 3:6]])
        if r == 17:
            row = row[:-1]                               # wrong length: rejected (gaussian_process.py:331-333)
        train_msgs.append(row)
        m = Msg()
        m.data = row.tolist()
        gp.training_data_callback(m)
    out["cb_train_msgs"] = np.array([np.pad(t, (0, 12 - len(t)), constant_values=np.nan) for t in train_msgs])
    out["cb_train_len"] = np.array([len(t) for t in train_msgs])
    out["cb_X_train"], out["cb_Y_train"] = gp.X_train.copy(), gp.Y_train.copy()
    gp.fit()
    reqs = [Xo[3, :9], Xo[4, :9], Xo[5, :10], X10[25, :9], Xo[200, :9]]        # the third has 10 entries: rejected (:344-346)
    for q in reqs:
        m = Msg()
        m.data = q.tolist()
        gp.prediction_request_callback(m)
    out["cb_req_msgs"] = np.array([np.pad(q, (0, 10 - len(q)), constant_values=np.nan) for q in reqs])
    out["cb_req_len"] = np.array([len(q) for q in reqs])
    out["cb_pred_pub"] = np.array(gp.prediction_pub.sent)
    out["cb_unc_pub"] = np.array(gp.uncertainty_pub.sent)
    np.savez_compressed(os.path.join(HERE, "package_kernel_ref.npz"), **out)
    print({k_: np.asarray(v).shape for k_, v in out.items()})


if __name__ == "__main__":
    main()
