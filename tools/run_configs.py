#!/usr/bin/env python3
"""Measure the BASELINE.json configurations on one MI355X, next to the reference CPU path
(scikit-learn on the host cores), and write gpurun_out/configs.{json,md}.

  C1  flight CSV (N=1000, D=10, P=6): SimpleQuadrotorGP.train_gp() with the optimiser, single and batched predict
  C2  N=4096, D=9, M=1024, fp64: gram / cholesky / alpha / predict, tolerance check against scikit-learn
  C3  = bench.py (N=65536, M=10000 mean+var, fp32) -- not repeated here
  C4  N=65536, 1 M queries, posterior means, fp32 (single-GPU leg of the scaling run)
  C5  3 per-axis ARD GPs: LML + analytic gradient evaluation at N=4096 and N=16384

    python tools/run_configs.py [--skip-cpu]
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def wall(fn, reps=3):
    import torch
    best = 1e30
    out = None
    for _ in range(reps):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        out = fn()
        torch.cuda.synchronize()
        best = min(best, time.perf_counter() - t0)
    return best, out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--skip-cpu", action="store_true")
    args = ap.parse_args()
    import torch
    from bench import synthetic_problem
    from unmanned_aerial_vehicles_amd import (RBF, ConstantKernel, GaussianProcessRegressor, SimpleQuadrotorGP,
                                              WhiteKernel)
    from unmanned_aerial_vehicles_amd.device import DeviceGP, get_backend

    be = get_backend(0)
    cores = len(os.sched_getaffinity(0))
    res = {"host_cores": cores, "gpu": torch.cuda.get_device_name(0)}
    try:
        from sklearn.gaussian_process import GaussianProcessRegressor as SkGPR
        from sklearn.gaussian_process.kernels import RBF as SkRBF, WhiteKernel as SkWhite
        have_skl = not args.skip_cpu
    except ImportError:
        have_skl = False

    # ------------------------------------------------------------------ C1
    d = np.load(os.path.join(ROOT, "tests", "golden", "csv_170501.npz"))
    ka = np.load(os.path.join(ROOT, "tests", "golden", "known_answers.npz"))
    X10, Y6 = d["X10"], d["Y6"]
    np.random.seed(0)
    gp = SimpleQuadrotorGP(max_data_points=10000)
    for xi, yi in zip(X10, Y6):
        gp.X_train.append(xi)
        gp.Y_train.append(yi)
    gp.train_gp()                       # warm-up (allocations, first launches)
    np.random.seed(0)
    t0 = time.perf_counter()
    gp.train_gp()
    t_train = time.perf_counter() - t0
    t_single, _ = wall(lambda: gp.predict_residual(X10[24, :6], X10[24, 6:]), reps=20)
    t_batch, _ = wall(lambda: gp.gp_model.predict(X10, return_std=True), reps=5)
    hor_X, hor_U = np.asarray(ka["ka3_hor_X"]), np.asarray(ka["ka3_hor_U"])     # (an .npz member is re-read on every access)
    t_hor, _ = wall(lambda: gp.build_gp_residuals(hor_X, hor_U, 0.1), reps=20)
    c1 = {"train_gp_s": t_train, "lml": gp.gp_model.log_marginal_likelihood_value_,
          "lml_reference": float(ka["ka3_lml"]), "kernel": str(gp.gp_model.kernel_),
          "predict_residual_single_ms": t_single * 1e3, "predict_1000_rows_mean_std_ms": t_batch * 1e3,
          "horizon25_residual_builder_ms": t_hor * 1e3}
    if have_skl:
        np.random.seed(0)
        t0 = time.perf_counter()
        sk = SkGPR(kernel=SkRBF(0.5) + SkWhite(0.1), alpha=1e-4, normalize_y=True, n_restarts_optimizer=1).fit(X10, Y6)
        c1["cpu_train_gp_s"] = time.perf_counter() - t0
        x1 = X10[24:25]
        t0 = time.perf_counter()
        for _ in range(20):
            sk.predict(x1, return_std=True)
        c1["cpu_predict_single_ms"] = (time.perf_counter() - t0) / 20 * 1e3
        t0 = time.perf_counter()
        sk.predict(X10, return_std=True)
        c1["cpu_predict_1000_rows_mean_std_ms"] = (time.perf_counter() - t0) * 1e3
        c1["cpu_horizon25_loop_ms"] = 25 * c1["cpu_predict_single_ms"]
    res["C1"] = c1
    print("C1", json.dumps(c1), flush=True)

    # ------------------------------------------------------------------ C2
    N, M = 4096, 1024
    X, Y, Xq = synthetic_problem(N, M)
    Yn = (Y - Y.mean(0)) / Y.std(0)
    dev = DeviceGP(X, Yn, be)
    dev.factorize(2.0, 1.0, 0.1001)
    t_gram, _ = wall(lambda: dev.gram(2.0, 1.0, 0.1001), reps=5)
    t_fac, _ = wall(lambda: dev.factorize(2.0, 1.0, 0.1001), reps=5)
    t_alpha, _ = wall(lambda: dev.solve_alpha(), reps=5)
    g = GaussianProcessRegressor(kernel=RBF(2.0) + WhiteKernel(0.1), alpha=1e-4, normalize_y=True, optimizer=None)
    t_fit, _ = wall(lambda: g.fit(X, Y), reps=3)
    t_mean, mean = wall(lambda: g.predict(Xq), reps=5)
    t_ms, (mean, std) = wall(lambda: g.predict(Xq, return_std=True), reps=5)
    c2 = {"gram_ms": t_gram * 1e3, "gram_GBps": N * N * 8 / t_gram / 1e9,
          "cholesky_ms": (t_fac - t_gram) * 1e3, "cholesky_GFLOPs": N ** 3 / 3 / (t_fac - t_gram) / 1e9,
          "alpha_ms": t_alpha * 1e3, "fit_total_ms": t_fit * 1e3,
          "predict_mean_1024_ms": t_mean * 1e3, "predict_mean_std_1024_ms": t_ms * 1e3,
          "mean_std_pred_per_s": M / t_ms,
          "rel_err_mean_vs_sklearn_golden": float(np.max(np.abs(mean - ka["c2_mean"])) / np.max(np.abs(ka["c2_mean"]))),
          "rel_err_std_vs_sklearn_golden": float(np.max(np.abs(std - ka["c2_std"])) / np.max(np.abs(ka["c2_std"])))}
    if have_skl:
        t0 = time.perf_counter()
        sk = SkGPR(kernel=SkRBF(2.0) + SkWhite(0.1), alpha=1e-4, normalize_y=True, optimizer=None).fit(X, Y)
        c2["cpu_fit_ms"] = (time.perf_counter() - t0) * 1e3
        t0 = time.perf_counter()
        sm, ss = sk.predict(Xq, return_std=True)
        c2["cpu_predict_mean_std_1024_ms"] = (time.perf_counter() - t0) * 1e3
        c2["rel_err_mean_vs_sklearn_live"] = float(np.max(np.abs(mean - sm)) / np.max(np.abs(sm)))
        c2["rel_err_std_vs_sklearn_live"] = float(np.max(np.abs(std - ss)) / np.max(np.abs(ss)))
    res["C2"] = c2
    print("C2", json.dumps(c2), flush=True)
    del dev, g

    # ------------------------------------------------------------------ C5 (before the big allocations)
    c5 = {}
    for N5 in (4096, 16384):
        X, Y, _ = synthetic_problem(N5, 1)
        ls = 2.0 * (1.0 + 0.1 * np.arange(9))
        times = []
        lmls = []
        for b in range(3):
            kern = ConstantKernel(1.0, "fixed") * RBF(ls, (0.1, 10.0)) + WhiteKernel(0.1, (1e-5, 1e1))
            gb = GaussianProcessRegressor(kernel=kern, alpha=1e-4, normalize_y=True, optimizer=None).fit(X, Y[:, b])
            t, (lml, grad) = wall(lambda: gb._lml_on_device(gb.kernel_.theta, True), reps=2)
            times.append(t)
            lmls.append(lml)
            gb._dev.release_grad_buffers()
            del gb
        # N^3/3 (potrf) + N^3/3 (trtri) + N^3/3 (W^T W) flops per evaluation
        c5[f"N{N5}"] = {"lml_grad_eval_s_per_gp": times, "three_gps_sequential_s": sum(times),
                        "TFLOPs_per_eval": N5 ** 3 / np.mean(times) / 1e12, "lml": lmls}
        # the batched surface: 3 models on 3 streams/handles, fused one-launch mean for 10 000 queries
        from unmanned_aerial_vehicles_amd import BatchedARDGP
        bg = BatchedARDGP(length_scale=ls, noise_level=0.1, alpha=1e-4, normalize_y=True, optimizer=None,
                          predict_dtype="float32").fit(X, Y)
        th = bg.thetas
        t, (bl, bgr) = wall(lambda: bg.log_marginal_likelihood(th, eval_gradient=True, fused=False), reps=2)
        c5[f"N{N5}"]["three_gps_concurrent_streams_s"] = t
        bg.log_marginal_likelihood(th, eval_gradient=True, fused=True)
        t, (bl, bgr) = wall(lambda: bg.log_marginal_likelihood(th, eval_gradient=True, fused=True), reps=2)
        c5[f"N{N5}"]["three_gps_fused_batch_s"] = t
        c5[f"N{N5}"]["lml_batched_matches"] = bool(np.allclose(bl, lmls, rtol=1e-12))
        bg.release_fused_buffers()
        Xq5 = np.random.default_rng(1).standard_normal((10000, 9))
        q5 = torch.as_tensor(Xq5, dtype=torch.float32, device=be.device)
        bg.predict_mean_dev(q5)
        t, _ = wall(lambda: bg.predict_mean_dev(q5), reps=5)
        c5[f"N{N5}"]["batched_3gp_mean_10k_queries_ms"] = t * 1e3       # what BatchedARDGP.predict_mean_dev does
        thr, bg.MFMA_MIN_QUERIES = bg.MFMA_MIN_QUERIES, 1 << 40
        t, _ = wall(lambda: bg.predict_mean_dev(q5), reps=5)            # forced: the fused vector-ALU kernel
        bg.MFMA_MIN_QUERIES = thr
        c5[f"N{N5}"]["fused_valu_3gp_mean_10k_queries_ms"] = t * 1e3
        t, _ = wall(lambda: [m._dev.predict_mean_dev(q5, m._y_train_mean, m._y_train_std, "float32", "valu")
                             for m in bg.models], reps=5)
        c5[f"N{N5}"]["per_model_valu_3gp_mean_10k_queries_ms"] = t * 1e3
        del bg
        if have_skl and N5 == 4096:
            from sklearn.gaussian_process.kernels import ConstantKernel as SkC
            kern = SkC(1.0, "fixed") * SkRBF(ls, (0.1, 10.0)) + SkWhite(0.1, (1e-5, 1e1))
            sk = SkGPR(kernel=kern, alpha=1e-4, normalize_y=True, optimizer=None).fit(X, Y[:, 0])
            t0 = time.perf_counter()
            l_ref, g_ref = sk.log_marginal_likelihood(sk.kernel_.theta, eval_gradient=True)
            c5[f"N{N5}"]["cpu_lml_grad_eval_s"] = time.perf_counter() - t0
            c5[f"N{N5}"]["lml_rel_err_vs_sklearn"] = abs(lmls[0] - l_ref) / abs(l_ref)
    res["C5"] = c5
    print("C5", json.dumps(c5), flush=True)

    # ------------------------------------------------------------------ C4 (single-GPU leg)
    N, M = 65536, 1 << 20
    X, Y, _ = synthetic_problem(N, 1)
    Yn = (Y - Y.mean(0)) / Y.std(0)
    dev = DeviceGP(X, Yn, be)
    dev.factorize(2.0, 1.0, 0.1001)
    dev.solve_alpha()
    q = torch.randn((M, 9), dtype=torch.float32, device=be.device)
    t, _ = wall(lambda: dev.predict_mean_dev(q, np.zeros(3), np.ones(3), "float32"), reps=3)
    tv, _ = wall(lambda: dev.predict_mean_dev(q, np.zeros(3), np.ones(3), "float32", "valu"), reps=3)
    flops = float(M) * N * (3 * 9 + 2 * 3 + 8)
    res["C4_single_gpu"] = {"queries": M, "n_train": N, "kernel": dev.mean_kernel_choice(), "mean_only_s": t,
                            "pred_per_s": M / t, "algorithmic_TFLOPs_all_pipes": flops / t / 1e12,
                            # the roofline fraction counts what runs on the vector ALU only (exp2 as 8 flops + 2P per pair); the
                            # 3D distance flops per pair run on the bf16 matrix pipe (bench.py --workload c4 reports the same)
                            "valu_TFLOPs": float(M) * N * (2 * 3 + 8) / t / 1e12,
                            "valu_frac_of_fp32_vector_peak": float(M) * N * (2 * 3 + 8) / t / 1e12 / 157.3,
                            "valu_kernel_s": tv, "valu_kernel_pred_per_s": M / tv}
    print("C4", json.dumps(res["C4_single_gpu"]), flush=True)

    os.makedirs("gpurun_out", exist_ok=True)
    with open("gpurun_out/configs.json", "w") as f:
        json.dump(res, f, indent=1)


if __name__ == "__main__":
    main()
