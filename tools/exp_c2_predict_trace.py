#!/usr/bin/env python3
"""BASELINE configs[1] serving step under `rocprofv3 --kernel-trace`: fit at N = 4096, then predict(mean + std) for 1024 fp64 queries,
launch by launch.    rocprofv3 --kernel-trace --output-format csv -d OUT -- python3 tools/exp_c2_predict_trace.py ; ... --join OUT"""
import csv
import glob
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def join(d):
    f = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)[0]
    rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
    marks = [i for i, r in enumerate(rows) if "cross_t_kernel" in r["Kernel_Name"]]
    last = marks[-1]
    # the launches of the last predict(return_std=True): from the mean kernel in front of the last cross-Gram to the finalise behind it
    i0 = last
    while i0 > 0 and "predict_mean" not in rows[i0]["Kernel_Name"]:
        i0 -= 1
    i1 = last
    while i1 + 1 < len(rows) and "finalize" not in rows[i1]["Kernel_Name"]:
        i1 += 1
    t0 = int(rows[i0]["Start_Timestamp"])
    prev_end = t0
    for r in rows[i0:i1 + 1]:
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        print(f"{(s - t0) / 1e3:9.1f} us  {(e - s) / 1e3:8.1f} us  gap {(s - prev_end) / 1e3:6.1f}  {r['Kernel_Name'][:90]}")
        prev_end = e
    print(f"span {(prev_end - t0) / 1e3:.1f} us, {i1 - i0 + 1} launches")


def main():
    if len(sys.argv) > 2 and sys.argv[1] == "--join":
        return join(sys.argv[2])
    import numpy as np
    from bench import synthetic_problem
    from unmanned_aerial_vehicles_amd import GaussianProcessRegressor, RBF, WhiteKernel
    X, Y, Xq = synthetic_problem(4096, 1024)
    gp = GaussianProcessRegressor(kernel=RBF(2.0) + WhiteKernel(0.1), alpha=1e-4, normalize_y=True, optimizer=None).fit(X, Y)
    for _ in range(5):
        t0 = time.perf_counter()
        gp.predict(Xq, return_std=True)
        print(f"predict(mean + std) {1e3 * (time.perf_counter() - t0):.3f} ms")


if __name__ == "__main__":
    main()
