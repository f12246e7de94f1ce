#!/usr/bin/env python3
"""Stability of the control-loop pattern: many refits interleaved with small predicts (package GP, threads) and
SimpleQuadrotorGP train/predict cycles; reports device memory before/after and the worst call latency."""
import os
import sys
import threading
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402
from unmanned_aerial_vehicles_amd import GaussianProcess, SimpleQuadrotorGP  # noqa: E402

rng = np.random.default_rng(0)
free0 = torch.cuda.mem_get_info()[0]
# --- SimpleQuadrotorGP: 30 train/predict cycles on growing data
gp = SimpleQuadrotorGP(max_data_points=600)
state = np.zeros(6); worst = 0.0
for cyc in range(30):
    for _ in range(40):
        ctrl = rng.standard_normal(4) * 0.5
        nxt = state + 0.02 * np.concatenate([state[3:6], ctrl[:3]]) + 0.002 * rng.standard_normal(6)
        gp.add_training_data(state, ctrl, nxt, 0.02)
        state = nxt * 0.98
    gp.train_gp()
    for _ in range(25):
        t0 = time.perf_counter(); m, v = gp.predict_residual(state, rng.standard_normal(4) * 0.5); worst = max(worst, time.perf_counter() - t0)
        assert np.all(np.isfinite(m)) and np.all(v >= 0)
print("SimpleQuadrotorGP: cycles 30, trained", gp.is_trained, "points", len(gp.X_train), "worst predict ms %.3f" % (worst * 1e3), flush=True)
# --- package GP: predict thread against refit thread
pg = GaussianProcess(input_dim=9, output_dim=3)
X = rng.standard_normal((400, 9)); Y = np.sin(X[:, :3])
for x, y in zip(X, Y):
    pg.add_training_data(x, y)
pg.fit()
stop = False; errs = []
def predictor():
    while not stop:
        try:
            m, v = pg.predict(rng.standard_normal((5, 9)))
            assert np.all(np.isfinite(m)) and np.all(v > 0)
        except Exception as e:  # noqa: BLE001
            errs.append(repr(e)); break
th = threading.Thread(target=predictor); th.start()
for i in range(40):
    pg.add_training_data(rng.standard_normal(9), rng.standard_normal(3) * 0.1)
    pg.fit()
stop = True; th.join()
print("package GP: 40 refits under a predicting thread, errors:", errs, flush=True)
torch.cuda.synchronize()
import gc; gc.collect(); torch.cuda.empty_cache()
free1 = torch.cuda.mem_get_info()[0]
print("device memory delta MB: %.1f" % ((free0 - free1) / 2**20))
assert not errs
