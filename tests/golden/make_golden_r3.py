#!/usr/bin/env python3
"""Round-3 golden fixture (run ONCE in the build container; the files of make_golden.py / make_golden_r2.py stay as
they are).

  flight_ref.npz   `GPTrainer.load_training_data` (src/px4/gp_trainer.py:49-119, with `_nominal_dynamics` :104-119) on
                   two synthetic `flight_data_*.npz` files (states_prev, controls, states_next, dt_values): the
                   (X, y) it returns without `max_samples`, and with `max_samples=40` under np.random.seed(7) (the
                   reference sub-samples with the global NumPy RNG, gp_trainer.py:94-97).  The files' arrays are stored
                   in the order the reference's (unsorted) glob visited them, so a loader with a sorted glob sees the
                   same sequence when the test writes them back as flight_data_0.npz, flight_data_1.npz.

    python tests/golden/make_golden_r3.py

Nothing here is reference source: the fixture holds synthetic inputs, a seed and the values the reference computed.
"""
import contextlib
import glob
import io
import os
import sys
import tempfile

import numpy as np

sys.dont_write_bytecode = True
REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))


def flight_fixture():
    sys.path.insert(0, f"{REF}/src/px4")
    import gp_trainer       # the reference module

    rng = np.random.default_rng(2025)
    files = {}
    for name, n in (("flight_data_20250101_000000.npz", 37), ("flight_data_20250102_120000.npz", 51)):
        sp = rng.standard_normal((n, 6))
        u = rng.standard_normal((n, 4))
        dt = 0.02 + 0.01 * rng.random(n)
        # a "true" next state: double integrator plus a smooth disturbance and noise
        nxt = np.concatenate([sp[:, :3] + sp[:, 3:] * dt[:, None], sp[:, 3:] + u[:, :3] * dt[:, None]], axis=1)
        nxt = nxt + 0.05 * np.sin(sp) + 0.01 * rng.standard_normal((n, 6))
        files[name] = dict(states_prev=sp, controls=u, states_next=nxt, dt_values=dt)
    out = {}
    with tempfile.TemporaryDirectory() as tmp, contextlib.redirect_stdout(io.StringIO()):
        for name, arrs in files.items():
            np.savez(os.path.join(tmp, name), **arrs)
        order = [os.path.basename(p) for p in glob.glob(os.path.join(tmp, "flight_data_*.npz"))]
        tr = gp_trainer.GPTrainer(data_dir=tmp, model_dir=tmp)
        X, y = tr.load_training_data()
        np.random.seed(7)
        Xs, ys = tr.load_training_data(max_samples=40)
    for k, name in enumerate(order):
        for key, a in files[name].items():
            out[f"file{k}_{key}"] = a
    out.update(n_files=np.array(len(order)), X=X, y=y, seed=np.array(7), max_samples=np.array(40), X_sub=Xs, y_sub=ys)
    np.savez_compressed(os.path.join(HERE, "flight_ref.npz"), **out)
    print("load_training_data:", X.shape, y.shape, "sub-sampled:", Xs.shape, "glob order:", order)


if __name__ == "__main__":
    flight_fixture()
