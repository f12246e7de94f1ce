#!/usr/bin/env python3
"""the reference CPU path (scikit-learn / SciPy, all host cores) and this repo's GPU path,
stage by stage, on the SURVEY.md section 8(d) synthetic inputs at N in {1024, 4096, 8192, 16384} (fp64, M = 1024 queries).

    python tools/cpu_gpu_stage_table.py [--sizes 1024,4096] [--out profiles/r01_cpu_gpu_stages.json]
"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402


def med(fn, reps):
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter(); fn(); ts.append(time.perf_counter() - t0)
    return sorted(ts)[len(ts) // 2]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--sizes", default="1024,4096,8192,16384")
    ap.add_argument("--out", default="")
    args = ap.parse_args()
    import torch
    from scipy.linalg import cho_solve, cholesky
    from sklearn.gaussian_process import GaussianProcessRegressor as SkGPR
    from sklearn.gaussian_process.kernels import RBF as SkRBF, WhiteKernel as SkWhite
    from unmanned_aerial_vehicles_amd import GaussianProcessRegressor, RBF, WhiteKernel
    from unmanned_aerial_vehicles_amd.device import DeviceGP, get_backend

    be = get_backend(0)
    info = {"host_threads": os.cpu_count()}
    try:
        from threadpoolctl import threadpool_info
        info["threadpools"] = [{k: p.get(k) for k in ("internal_api", "num_threads", "version")} for p in threadpool_info()]
    except Exception:  # noqa: BLE001
        pass
    try:
        with open("/proc/cpuinfo") as f:
            info["cpu_model"] = next(ln.split(":", 1)[1].strip() for ln in f if ln.startswith("model name"))
    except Exception:  # noqa: BLE001
        pass
    rows = []
    M, ls, sf2, noise, jitter = 1024, 2.0, 1.0, 0.1, 1e-4
    for N in [int(v) for v in args.sizes.split(",")]:
        rng = np.random.default_rng(0)
        X = rng.standard_normal((N, 9)); W = rng.standard_normal((9, 3))
        Y = np.sin(X @ W) + 0.1 * rng.standard_normal((N, 3))
        Xq = np.random.default_rng(1).standard_normal((M, 9))
        reps = 5 if N <= 4096 else 3
        r = {"N": N, "M": M}
        # ---- CPU: the calls scikit-learn's fit/predict make (sklearn/_gpr.py:343-364, 441-494)
        kern = SkRBF(ls) + SkWhite(noise)
        r["cpu_gram_s"] = med(lambda: kern(X), reps)
        K = kern(X); K[np.diag_indices_from(K)] += jitter
        r["cpu_cholesky_s"] = med(lambda: cholesky(K, lower=True, check_finite=False), reps)
        L = cholesky(K, lower=True, check_finite=False)
        Yn = (Y - Y.mean(0)) / Y.std(0)
        r["cpu_alpha_s"] = med(lambda: cho_solve((L, True), Yn, check_finite=False), reps)
        sk = SkGPR(kernel=kern, alpha=jitter, normalize_y=True, optimizer=None).fit(X, Y)
        r["cpu_fit_total_s"] = med(lambda: SkGPR(kernel=kern, alpha=jitter, normalize_y=True, optimizer=None).fit(X, Y), 1 if N > 8192 else 3)
        r["cpu_predict_mean_s"] = med(lambda: sk.predict(Xq), reps)
        r["cpu_predict_mean_std_s"] = med(lambda: sk.predict(Xq, return_std=True), reps)
        m_ref, s_ref = sk.predict(Xq, return_std=True)
        # ---- GPU: same stages through the C ABI
        dev = DeviceGP(X, Yn, be)
        sync = torch.cuda.synchronize
        dev.factorize(ls, sf2, noise + jitter); sync()
        r["gpu_gram_s"] = med(lambda: (dev.gram(ls, sf2, noise + jitter), sync()), reps)
        t = med(lambda: (dev.factorize(ls, sf2, noise + jitter), sync()), reps)
        r["gpu_cholesky_s"] = t - r["gpu_gram_s"]
        r["gpu_alpha_s"] = med(lambda: (dev.solve_alpha(), sync()), reps)
        gp = GaussianProcessRegressor(kernel=RBF(ls) + WhiteKernel(noise), alpha=jitter, normalize_y=True, optimizer=None)
        gp.fit(X, Y)
        r["gpu_fit_total_s"] = med(lambda: GaussianProcessRegressor(kernel=RBF(ls) + WhiteKernel(noise), alpha=jitter,
                                                                    normalize_y=True, optimizer=None).fit(X, Y), reps)
        gp.predict(Xq, return_std=True)
        r["gpu_predict_mean_s"] = med(lambda: gp.predict(Xq), reps)
        r["gpu_predict_mean_std_s"] = med(lambda: gp.predict(Xq, return_std=True), reps)
        m, s = gp.predict(Xq, return_std=True)
        r["rel_err_mean"] = float(np.max(np.abs(m - m_ref)) / np.max(np.abs(m_ref)))
        r["rel_err_std"] = float(np.max(np.abs(s - s_ref) / s_ref))
        r["cpu_cholesky_GFLOPs"] = N ** 3 / 3 / r["cpu_cholesky_s"] / 1e9
        r["gpu_cholesky_GFLOPs"] = N ** 3 / 3 / r["gpu_cholesky_s"] / 1e9
        r["cpu_pred_per_s"] = M / r["cpu_predict_mean_std_s"]
        r["gpu_pred_per_s"] = M / r["gpu_predict_mean_std_s"]
        rows.append(r)
        print(json.dumps(r), flush=True)
        del dev, gp, sk, K, L
    print("\n| N | stage | CPU (scikit-learn/SciPy, %d threads) | GPU (this repo, fp64) | ratio |" % (os.cpu_count() or 0))
    print("|---|---|---|---|---|")
    for r in rows:
        for st, lab in (("gram", "gram"), ("cholesky", "cholesky"), ("alpha", "alpha solve"), ("fit_total", "fit() total"),
                        ("predict_mean", "predict mean, 1024 q"), ("predict_mean_std", "predict mean+std, 1024 q")):
            c, g = r[f"cpu_{st}_s"], r[f"gpu_{st}_s"]
            print(f"| {r['N']} | {lab} | {c*1e3:.2f} ms | {g*1e3:.3f} ms | {c/g:.0f}x |")
    if args.out:
        os.makedirs(os.path.dirname(args.out) or ".", exist_ok=True)
        with open(args.out, "w") as f:
            json.dump({"info": info, "rows": rows}, f, indent=1)


if __name__ == "__main__":
    main()
