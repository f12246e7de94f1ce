#!/usr/bin/env python3
"""Headline benchmark: GP predictions/s (posterior mean + variance) at N_train = 65 536, D = 9.

    python bench.py --gpus N --steps K --warmup W

N > 1 runs one rank per GPU over RCCL.  Two ways in, same result: (a) the driver starts the ranks itself with
`python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...` (RANK / WORLD_SIZE in the
environment), or (b) plain `python bench.py --gpus N`: this process then - BEFORE importing torch or touching the
GPU - starts that same torch.distributed.run command as a child process, relays rank 0's single JSON line and exits
with the child's status (a process that has initialised the GPU is never re-executed).

Workload (BASELINE.json configs[2], SURVEY.md §8d "C3"): synthetic N_train x 9 training set with 3 outputs, fitted
once on the GPU in fp64 (Gram build + blocked Cholesky + alpha; untimed set-up, reported under "fit"), then every
timed step predicts mean AND variance for one batch of M = 10 000 query points (horizon 20 x 500 rollouts) in fp32
with the queries already resident in HBM.  N > 1 is weak scaling: every rank holds a replica of the model (each rank
fits redundantly: deterministic, no communication) and its own 10 000-query batch per step, and one RCCL all-gather
of the [mean | var] shards closes each step.  value = predictions of all ranks / max-over-ranks time.
`--workload c4` is BASELINE.json configs[3]: 1 048 576 queries in total sharded over the ranks, posterior means only,
all-gather of the means (strong scaling).

Prints ONE JSON line on rank 0 with the driver's contract fields plus
  "roofline":     the dominant kernel (the one GEMM launch behind |L^-1 k*|^2) against the matrix-pipe peak it
                  runs on, its duration measured live with HIP events on the launch stream over the timed steps;
  "cpu_baseline": the reference's CPU path (scikit-learn GaussianProcessRegressor, the library the reference
                  delegates to; falls back to the repo's NumPy oracle) on a bounded sample;
  "fit":          Gram GB/s vs the HBM roofline and Cholesky GFLOP/s of the set-up phase;
  "parity":       the timed path's outputs against the fp64 path on the same batch (checked after the timed loop).
"""
import argparse
import ctypes as C
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0        # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
MFMA_F32_PEAK_TF = 157.3      # dense fp32 MFMA (= fp32 vector peak)
MFMA_F64_PEAK_TF = 78.6       # fp64 matrix peak (MI355X datasheet; used for the Cholesky fraction only)
MFMA_BF16_PEAK_TF = 2516.6    # dense bf16 MFMA: 256 CUs x 4 SIMDs x 1024 flop/clk x 2.4 GHz (guide: "~2.5 PF dense")
PMC_TRAFFIC_FILE = os.path.join(ROOT, "profiles", "pmc_traffic.json")   # written from the --pmc passes of tools/pmc_k5.sh


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None, help="timed steps (default: 8; train: 2; lml: 5)")
    ap.add_argument("--warmup", type=int, default=None, help="untimed steps (default: 2; train: 1)")
    ap.add_argument("--n-train", type=int, default=None, help="default: 65536; train: 10000; lml: 16384")
    ap.add_argument("--queries", type=int, default=10000)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the extra timing of the other variance paths")
    ap.add_argument("--workload", default="c3", choices=["c3", "c4", "gram", "train", "lml"],
                    help="c3 (default, the headline): mean+var for 10 000 queries per GPU per step, weak scaling; "
                         "c4: BASELINE configs[3] - 1 048 576 queries in total sharded over the GPUs, posterior means "
                         "only, RCCL all-gather of the means (strong scaling); gram: the fp64 RBF Gram build at N_train, "
                         "row-sharded over the GPUs with no exchange (SURVEY 8e; strong scaling, GB/s); "
                         "train: the reference's offline-training workload (SimpleQuadrotorGP.train_gp(), N = 10 000, D = 10, "
                         "P = 6, L-BFGS-B + 1 restart); lml: BASELINE configs[4] - LML + gradient of three fused per-axis ARD "
                         "GPs at N = 16 384")
    ap.add_argument("--replicate", default="broadcast", choices=["broadcast", "refit"],
                    help="--gpus N > 1: how the ranks get the model - broadcast: rank 0 fits and broadcasts X, alpha and the "
                         "split inverse factor over RCCL (the other ranks hold no factor); refit: every rank fits redundantly")
    ap.add_argument("--var-method", default="auto", choices=["auto", "inverse_split", "inverse_split2", "inverse", "solve"],
                    help="auto = inverse_split2: |L^-1 k*|^2 with the explicit inverse factor, one fused GEMM launch on the "
                         "16-bit matrix pipe, every fp32 operand as two round-to-nearest fp16 parts (represented to 2^-23), "
                         "three products per block, fp32 accumulation, operands from L2 straight into registers; "
                         "inverse_split: three exact bf16 parts, six products; inverse: the same launch on the exact-fp32 "
                         "MFMA; solve: blocked triangular solve chain")
    args = ap.parse_args(argv)
    wl = args.workload
    if args.steps is None:
        args.steps = {"train": 2, "lml": 5}.get(wl, 8)
    if args.warmup is None:
        args.warmup = {"train": 1}.get(wl, 2)
    if args.n_train is None:
        args.n_train = {"train": 10000, "lml": 16384}.get(wl, 65536)
    return args


def launch_ranks(args):
    """Plain `python bench.py --gpus N` with N > 1: start the ranks as a child torch.distributed.run job.  Nothing in
    this process has imported torch or touched the GPU.  The rendezvous port is found by bind-and-close, which leaves a
    window in which another process can take it: a child that fails within a minute without a result line is started
    again on a fresh port (twice at most)."""
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # dmabuf IPC: required by RCCL on this host driver
    env["BENCH_LAUNCHED_BY_PARENT"] = "1"
    rc, lines = 1, []
    for attempt in range(3):
        s = socket.socket()
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
        s.close()
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
               "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        t0 = time.perf_counter()
        p = subprocess.Popen(cmd, stdout=subprocess.PIPE, env=env, text=True)
        lines = []
        for ln in p.stdout:                       # rank 0 prints exactly one JSON line; anything else goes to stderr
            if ln.lstrip().startswith('{"metric"'):
                lines.append(ln.strip())
            else:
                sys.stderr.write(ln)
        rc = p.wait()
        if rc == 0 or lines or time.perf_counter() - t0 > 60.0:
            break
        sys.stderr.write(f"bench.py: the ranks failed after {time.perf_counter() - t0:.0f} s (rc {rc}); starting them again on a new port\n")
    if lines:
        print(lines[-1], flush=True)
    if rc == 0 and not lines:
        sys.stderr.write("bench.py: the ranks finished without printing a result line\n")
        rc = 1
    sys.exit(rc)


class BoardSensors:
    """Board power and shader clock of this process's GPU while the timed steps run: the card's own hwmon files
    (/sys/class/drm/card*/device/hwmon/hwmon*/power1_input in microwatts, freq1_input in Hz), matched to the device by PCI
    bus id, sampled every 10 ms by a host thread.  Evidence for `roofline.pmc`: whether the matrix pipe's clock under the
    variance launch is the power management's doing.  Returns None where the files are not readable."""

    def __init__(self, device_index):
        import glob
        self.power = self.freq = self.cap = None
        self.samples = []
        try:
            import torch
            pr = torch.cuda.get_device_properties(device_index)
            want = f"{getattr(pr, 'pci_domain_id', 0):04x}:{pr.pci_bus_id:02x}:{getattr(pr, 'pci_device_id', 0):02x}."
        except Exception:                                   # noqa: BLE001
            want = None
        for hw in sorted(glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*")) if want else []:
            real = os.path.realpath(os.path.join(hw, "..", ".."))
            if want not in real:
                continue
            if os.path.exists(os.path.join(hw, "power1_input")):
                self.power = os.path.join(hw, "power1_input")
                self.freq = os.path.join(hw, "freq1_input")
                self.cap = os.path.join(hw, "power1_cap")
                self.pci = os.path.basename(real)
                break
        self._stop = None

    @staticmethod
    def _read(path):
        try:
            with open(path) as fh:
                return float(fh.read().strip())
        except Exception:                                   # noqa: BLE001
            return None

    def start(self):
        if not self.power:
            return
        import threading
        self._stop = threading.Event()

        def run():
            while not self._stop.is_set():
                p, fq = self._read(self.power), self._read(self.freq)
                if p is not None:
                    self.samples.append((p * 1e-6, (fq or 0.0) * 1e-6))
                self._stop.wait(0.01)
        self._thread = threading.Thread(target=run, daemon=True)
        self._thread.start()

    def stop(self):
        if not self._stop:
            return None
        self._stop.set()
        self._thread.join()
        if not self.samples:
            return None
        pw = np.array([a for a, _ in self.samples])
        fq = np.array([b for _, b in self.samples])
        cap = self._read(self.cap)
        return {"board_power_w_avg": float(pw.mean()), "board_power_w_max": float(pw.max()),
                "power_cap_w": cap * 1e-6 if cap else None,
                "sclk_mhz_avg": float(fq.mean()), "sclk_mhz_min": float(fq.min()), "sclk_mhz_max": float(fq.max()),
                "samples": int(len(pw)), "pci": self.pci,
                "source": "the card's hwmon power1_input / freq1_input, sampled every 10 ms by a host thread over the timed steps"}


def synthetic_problem(N, M, D=9, P=3, qseed=1):
    """SURVEY.md §8(d) deterministic inputs (same generator as oracle.gp_oracle.synthetic_problem)."""
    rng = np.random.default_rng(0)
    X = rng.standard_normal((N, D))
    W = rng.standard_normal((D, P))
    Y = np.sin(X @ W) + 0.1 * rng.standard_normal((N, P))
    Xq = np.random.default_rng(qseed).standard_normal((M, D))
    return X, Y, Xq


def cpu_baseline(M_sample=2000, N_sample=8192, n_full=65536):
    """Reference CPU path on a bounded sample: fit at N_sample (untimed), then time
    predict(return_std=True) for M_sample queries."""
    X, Y, Xq = synthetic_problem(N_sample, M_sample)
    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    kind = "reference"
    try:
        from sklearn.gaussian_process import GaussianProcessRegressor as SkGPR
        from sklearn.gaussian_process.kernels import RBF as SkRBF, WhiteKernel as SkWhite
        t0 = time.perf_counter()
        gp = SkGPR(kernel=SkRBF(2.0) + SkWhite(0.1), alpha=1e-4, normalize_y=True, optimizer=None).fit(X, Y)
        t_fit = time.perf_counter() - t0
        t0 = time.perf_counter()
        gp.predict(Xq, return_std=True)
        t_pred = time.perf_counter() - t0
        what = "scikit-learn GaussianProcessRegressor (the library the reference's simple_gp.py calls)"
    except ImportError:
        from oracle import gp_oracle as O
        kind = "port"
        t0 = time.perf_counter()
        st = O.fit_fixed(X, Y, 2.0, 1.0, 0.1, 1e-4)
        t_fit = time.perf_counter() - t0
        t0 = time.perf_counter()
        O.predict(st, Xq, return_std=True)
        t_pred = time.perf_counter() - t0
        what = "oracle/gp_oracle.py (NumPy/SciPy restatement)"
    v = M_sample / t_pred
    # `value` is the like-for-like figure at the headline N_train: predict cost grows as N_train^2 (the triangular solve), so
    # the sample's rate is scaled by (n_train_of_sample / n_train)^2 - and replaced by a MEASURED rate when the host has the
    # memory to run scikit-learn's predict on the full-size factor (main(): cpu_full_size_predict)
    return {"value": v * (N_sample / float(n_full)) ** 2, "unit": "predictions/s", "cores": cores, "kind": kind,
            "n_train": n_full, "basis": "extrapolated", "rule": "sample value x (n_train_of_sample / n_train)^2",
            "sample_at_smaller_n_train": {"value": v, "n_train": N_sample, "queries": M_sample, "predict_seconds": t_pred,
                                          "fit_seconds": t_fit},
            "sample": f"{what}: predict(return_std=True) of {M_sample} queries at N_train={N_sample} (fit {t_fit:.1f} s untimed, "
                      f"predict {t_pred:.2f} s), scaled to N_train={n_full}"}


def cpu_full_size_predict(dev, X, y_mean, y_std, alpha_host, ls, noise, M_cpu=200):
    """A measured (not extrapolated) CPU figure at the headline N_train: scikit-learn's predict(return_std=True) on a
    regressor object that carries the factor computed on the GPU (L_ downloaded once, untimed) - the reference's
    predict arithmetic (`_gpr.py:441-494`: kernel_(X, X_train_), K_trans @ alpha_, solve_triangular(L_, K_trans.T),
    einsum) on the host cores.  Only when the host has the memory for the 34 GB factor."""
    try:
        import psutil
        from sklearn.gaussian_process import GaussianProcessRegressor as SkGPR
        from sklearn.gaussian_process.kernels import RBF as SkRBF, WhiteKernel as SkWhite
    except ImportError as e:
        return {"skipped": f"{e}"}
    need = dev.N * dev.N * 8
    avail = psutil.virtual_memory().available
    if avail < 2.5 * need:
        return {"skipped": f"host memory: {avail / 1e9:.0f} GB available, {2.5 * need / 1e9:.0f} GB wanted"}
    import torch
    t0 = time.perf_counter()
    Lh = torch.empty((dev.N, dev.N), dtype=torch.float64)
    rows = 8192
    for r0 in range(0, dev.N, rows):           # (strided device rows -> contiguous host rows, block by block)
        Lh[r0:r0 + rows].copy_(dev.K[r0:r0 + rows, : dev.N])
    t_dl = time.perf_counter() - t0
    kern = SkRBF(ls) + SkWhite(noise)
    gp = SkGPR(kernel=kern, alpha=1e-4, normalize_y=True, optimizer=None)
    gp.kernel_ = kern
    gp.X_train_, gp.L_, gp.alpha_ = X, Lh.numpy(), alpha_host
    gp._y_train_mean, gp._y_train_std = y_mean, y_std
    gp.y_train_ = np.zeros((dev.N, alpha_host.shape[1]))
    gp.n_features_in_ = X.shape[1]
    Xq = np.random.default_rng(1).standard_normal((M_cpu, X.shape[1]))
    t0 = time.perf_counter()
    mean, std = gp.predict(Xq, return_std=True)
    t_pred = time.perf_counter() - t0
    return {"value": M_cpu / t_pred, "unit": "predictions/s", "n_train": dev.N, "queries": M_cpu,
            "predict_seconds": t_pred, "factor_download_seconds": t_dl, "mean": mean, "std": std, "Xq": Xq}


def pmc_traffic(kernel_key, N, M):
    """HBM/fabric bytes per launch of the dominant kernel from the committed --pmc passes (separate rocprofv3 runs:
    the counters cannot be collected inside this process), or None when no pass exists for this kernel and shape."""
    try:
        with open(PMC_TRAFFIC_FILE) as f:
            for e in json.load(f)["entries"]:
                if e["kernel_key"] == kernel_key and e["n_train"] == N and e["queries"] == M:
                    return e
    except (OSError, ValueError, KeyError):
        pass
    return None


def pmc_workload_traffic(workload, N):
    """Fabric/HBM bytes per STEP of a whole workload (tools/pmc_workload.sh: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this
    very command, set-up cancelled by differencing two step counts), or None when no pass exists for this workload and size."""
    try:
        with open(PMC_TRAFFIC_FILE) as f:
            for e in json.load(f).get("workloads", []):
                if e["workload"] == workload and e["n_train"] == N:
                    return e
    except (OSError, ValueError, KeyError):
        pass
    return None


def bench_gram(args, be, rank, world, use_dist, ranks_seen):
    """--workload gram: K1 alone, sharded by row slabs (each rank writes the rows it owns; nothing is exchanged)."""
    import torch
    import torch.distributed as dist
    from unmanned_aerial_vehicles_amd import _lib
    from unmanned_aerial_vehicles_amd.sharded import gram_slab_bounds, sharded_gram
    N, D = args.n_train, 9
    X, _, _ = synthetic_problem(N, 1)
    Xd = be.upload(X)
    row0, nrows = gram_slab_bounds(N, world, rank)
    Np = (N + 127) // 128 * 128
    slab = be.empty((nrows, Np), torch.float64)

    def sync_all():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    def step():
        sharded_gram(Xd, 2.0, 1.0, 0.1001, world, rank, be, out=slab)

    for _ in range(max(args.warmup, 1)):
        step()
    sync_all()
    evs = []
    t0 = time.perf_counter()
    for _ in range(args.steps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        step()
        b.record()
        evs.append((a, b))
    sync_all()
    dt = time.perf_counter() - t0
    mine = float(np.mean([a.elapsed_time(b) for a, b in evs])) * 1e-3
    if use_dist:
        t = torch.tensor([dt], dtype=torch.float64, device=be.device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    ok = bool(torch.isfinite(slab).all()) and (nrows == 0 or float(slab[0, row0]) == 1.0 + 0.1001)
    assert ok, "Gram slab check failed"
    if rank == 0:
        total_bytes = float(Np) * Np * 8 + float(N) * D * 8          # N^2 s + N D s
        my_bytes = float(nrows) * Np * 8 + float(N) * D * 8
        line = {"metric": "RBF Gram build GB/s (fp64, row-sharded over the GPUs) at N_train=65536, D=9",
                "value": total_bytes * args.steps / dt / 1e9, "unit": "GB/s", "n_gpus": world, "ranks_seen": ranks_seen,
                "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3,
                "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
                "config": {"workload": f"K1: N_train={N}, D={D}, fp64 Gram matrix ({Np} x {Np}), rows dealt to {world} GPU(s) "
                                       f"in 128-row tiles, no exchange", "n_train": N, "features": D,
                           "rows_of_rank0": nrows,
                           "kernel": "gram_strip_kernel (symmetric tiles computed once, written twice)" if world == 1
                                     else "cross_t_kernel per slab (every entry computed directly: no mirroring across ranks)"},
                "roofline": {"bound": "hbm" if world == 1 else "valu", "achieved": my_bytes / mine / 1e9, "peak": HBM_PEAK_GBPS,
                             "unit": "GB/s", "frac": my_bytes / mine / 1e9 / HBM_PEAK_GBPS,
                             "traffic": (pmc_workload_traffic("gram", N) or {}).get("bytes_per_step") if world == 1 else None,
                             "traffic_source": (pmc_workload_traffic("gram", N) or {}).get("source") if world == 1 else None,
                             "kernel_ms": mine * 1e3,
                             "note": "rank 0's slab bytes over its own launch time (HIP events); a slab kernel computes every "
                                     "entry (about 45 fp64 operations each), so from two ranks up it is bound by the "
                                     "fp64 vector rate rather than by HBM"}}
        print(json.dumps(line), flush=True)
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


def synthetic_flight_problem(N, D=10, P=6, seed=7):
    """Flight-like rows for the training workload (same generator as oracle.gp_oracle.synthetic_flight_problem)."""
    rng = np.random.default_rng(seed)
    X = 0.6 * rng.standard_normal((N, D))
    X[:, 2] -= 3.0
    if D >= 10:
        X[:, 9] *= 1e-3
    W = rng.standard_normal((6, P))
    va = X[:, 3:9]
    Y = 0.03 * np.sin(va @ W) - 0.01 * np.pad(va[:, :3] * np.abs(va[:, :3]), ((0, 0), (0, max(P - 3, 0))))[:, :P]
    Y = Y + 0.005 * rng.standard_normal((N, P))
    return X, Y


def bench_train(args, be):
    """--workload train: the reference's real offline-training job, src/px4/train_gp_offline.py:124-140 ->
    SimpleQuadrotorGP(max_data_points=10000).train_gp() (src/px4/simple_gp.py:156-185: RBF(0.5) + White(0.1), alpha 1e-4,
    normalize_y, L-BFGS-B + 1 restart) on N synthetic flight-like rows, D = 10, P = 6.  A step = one train_gp();
    every LML + gradient evaluation of the optimiser (sklearn/_gpr.py:537-652) is timed."""
    import contextlib
    import torch
    from unmanned_aerial_vehicles_amd import SimpleQuadrotorGP, _lib
    from unmanned_aerial_vehicles_amd import gpr as gpr_mod
    N, D, P = args.n_train, 10, 6
    X, Y = synthetic_flight_problem(N)
    evals = []
    orig = gpr_mod.GaussianProcessRegressor._lml_on_device

    def counted(self, theta, eval_gradient, dev=None):
        t0 = time.perf_counter()
        r = orig(self, theta, eval_gradient, dev)          # (ends with host reads of the reductions: synchronous)
        evals.append((time.perf_counter() - t0, bool(eval_gradient)))
        return r

    gpr_mod.GaussianProcessRegressor._lml_on_device = counted
    res = {}

    def step():
        np.random.seed(0)                                    # the restart point is drawn from the global RNG, as in the reference
        gp = SimpleQuadrotorGP(max_data_points=10000)
        gp.X_train.extend(X)
        gp.Y_train.extend(Y)
        with contextlib.redirect_stdout(sys.stderr):        # (train_gp reports like the reference: stdout carries ONE line)
            gp.train_gp()
        assert gp.is_trained
        res["gp"] = gp
        return gp

    try:
        for _ in range(args.warmup):
            step()
        torch.cuda.synchronize()
        evals.clear()
        be.check(be.lib.gpk_timing(be.h, 1))
        t0 = time.perf_counter()
        for _ in range(args.steps):
            step()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
    finally:
        gpr_mod.GaussianProcessRegressor._lml_on_device = orig
    ms = np.zeros(64)
    n = C.c_int(0)
    be.check(be.lib.gpk_kernel_times(be.h, _lib.GPK_TIMED_POTRF, ms.ctypes.data_as(_lib._dp), 64, C.byref(n)))
    potrf_ms = float(np.mean(ms[: n.value])) if n.value else None
    be.check(be.lib.gpk_timing(be.h, 0))
    gm = res["gp"].gp_model
    ge = [t for t, g in evals if g]
    gv = [t for t, g in evals if not g]
    # the optimiser's two runs (start + one restart) execute one after the other at this size (side by side on two handles /
    # streams up to 7168 rows, where an evaluation's own wall time includes its share of waiting for the other run's launches -
    # gpr.py: _optimise_from), and the rate is taken over the whole train_gp(): every gradient evaluation N^3 flops (potrf N^3/3 + inverse factor N^3/3 + W^T W N^3/3),
    # every value-only evaluation and the final fit 2 N^3 / 3
    flops = (len(ge) * 1.0 + (len(gv) + args.steps) * 2.0 / 3.0) * float(N) ** 3 / args.steps
    s_train = dt / args.steps
    s_eval = s_train / max(len(ge) / args.steps, 1)
    tr = pmc_workload_traffic("train", N)
    line = {
        "metric": "offline GP training wall time: SimpleQuadrotorGP.train_gp() (L-BFGS-B + 1 restart, LML + analytic gradient "
                  "per evaluation) on N_train rows, D=10, P=6",
        "value": dt / args.steps, "unit": "s", "n_gpus": 1, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": dt / args.steps * 1e3, "higher_is_better": False, "scaling": "weak", "vs_baseline": None,
        "dtype": "f64", "data": "synthetic flight-like rows (bench.synthetic_flight_problem), random restart seeded",
        "config": {"workload": f"train: N_train={N}, D={D}, P={P}, kernel RBF(0.5)+White(0.1), alpha=1e-4, normalize_y, "
                               "n_restarts_optimizer=1 (src/px4/train_gp_offline.py:124-140, simple_gp.py:156-185)"},
        "lml_grad_evaluations_per_train": len(ge) / args.steps,
        "s_per_lml_grad_evaluation": s_eval,
        "s_per_lml_grad_evaluation_note": "train_gp() wall time / gradient evaluations (the optimiser's runs: one after the other above 7168 rows)",
        "s_per_lml_grad_evaluation_own_wall_mean": float(np.mean(ge)), "s_per_lml_grad_evaluation_own_wall_min": float(np.min(ge)),
        "potrf_ms_final_fit": potrf_ms,
        "kernel": str(gm.kernel_), "lml": float(gm.log_marginal_likelihood_value_),
        "roofline": {"bound": "mfma", "achieved": flops / s_train / 1e12, "peak": MFMA_F64_PEAK_TF, "unit": "TFLOP/s",
                     "frac": flops / s_train / 1e12 / MFMA_F64_PEAK_TF,
                     "traffic": tr["bytes_per_step"] if tr else None,
                     "traffic_note": ("fabric bytes per train_gp() from the counter passes of this command; by kernel under pmc"
                                      if tr else "no counter pass for this size"),
                     "pmc": ({"by_kernel": tr["by_kernel"], "source": tr["source"]} if tr else None),
                     "what": "the fp64 flops of one train_gp() - N^3 per LML + gradient evaluation (factor, inverse factor, "
                             "K^-1 = W^T W; SURVEY 8d K6), 2 N^3 / 3 per value-only evaluation and for the final fit - over its "
                             "wall time, host side of the optimiser, Gram builds and reductions included"},
        "peak_hbm_bytes_per_rank": int(torch.cuda.max_memory_allocated(be.device)),
    }
    if not args.no_cpu_baseline:
        line["cpu_baseline"] = cpu_train_baseline(N)
    print(json.dumps(line), flush=True)


def cpu_train_baseline(n_full, n_sample=2000):
    """scikit-learn (the library the reference's train_gp() calls) on the same generator at a bounded N; the N^3
    extrapolation to n_full is labelled as such."""
    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        import warnings
        from sklearn.gaussian_process import GaussianProcessRegressor as SkGPR
        from sklearn.gaussian_process.kernels import RBF as SkRBF, WhiteKernel as SkWhite
    except ImportError as e:
        return {"skipped": repr(e)}
    X, Y = synthetic_flight_problem(n_sample)
    n_eval = [0]
    orig = SkGPR.log_marginal_likelihood

    def counted(self, *a, **k):
        n_eval[0] += 1
        return orig(self, *a, **k)

    SkGPR.log_marginal_likelihood = counted
    try:
        np.random.seed(0)
        t0 = time.perf_counter()
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            g = SkGPR(kernel=SkRBF(0.5) + SkWhite(0.1), alpha=1e-4, normalize_y=True, n_restarts_optimizer=1).fit(X, Y)
        dt = time.perf_counter() - t0
    finally:
        SkGPR.log_marginal_likelihood = orig
    scale = (n_full / n_sample) ** 3
    return {"value": dt * scale, "unit": "s", "cores": cores, "kind": "reference", "basis": "extrapolated",
            "rule": "measured seconds at the sample size x (N_train / sample N_train)^3",
            "measured_seconds": dt, "measured_at_n_train": n_sample, "evaluations": n_eval[0],
            "s_per_evaluation_measured": dt / max(n_eval[0], 1), "kernel": str(g.kernel_),
            "sample": f"scikit-learn GaussianProcessRegressor(RBF+White, n_restarts_optimizer=1).fit at N_train={n_sample} "
                      f"({dt:.1f} s, {n_eval[0]} evaluations, {cores} threads), N^3-extrapolated to N_train={n_full}"}


def bench_lml(args, be):
    """--workload lml: BASELINE configs[4] (C5) - LML + analytic gradient of three per-axis ARD GPs on shared inputs as ONE
    fused launch chain (sklearn/_gpr.py:537-652 per model; src/px4/gp_trainer.py:163-179).  A step = one evaluation of all
    three models.  `roofline`: the chain's N^3 fp64 flops per model against the fp64 matrix peak; `roofline_grad_kernel`: the
    streaming pass over K^-1 (`lml_grad_kernel`) against HBM, its duration from the library's event brackets."""
    import torch
    from unmanned_aerial_vehicles_amd import BatchedARDGP, _lib
    N, D, B = args.n_train, 9, 3
    X, Y, _ = synthetic_problem(N, 1)
    ls = 2.0 * (1.0 + 0.1 * np.arange(D))
    bg = BatchedARDGP(length_scale=ls, noise_level=0.1, alpha=1e-4, normalize_y=True, optimizer=None).fit(X, Y)
    th = np.array(bg.thetas)
    for _ in range(max(args.warmup, 1)):
        lml, grad = bg.log_marginal_likelihood(th, eval_gradient=True, fused=True)
    torch.cuda.synchronize()
    fbe = bg._fs["be"]
    fbe.check(fbe.lib.gpk_timing(fbe.h, 1))
    t0 = time.perf_counter()
    for _ in range(args.steps):
        lml, grad = bg.log_marginal_likelihood(th, eval_gradient=True, fused=True)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0

    def times(tag):
        ms = np.zeros(64)
        n = C.c_int(0)
        fbe.check(fbe.lib.gpk_kernel_times(fbe.h, tag, ms.ctypes.data_as(_lib._dp), 64, C.byref(n)))
        return ms[: n.value].copy()

    grad_ms, potrf_ms = times(_lib.GPK_TIMED_GRAD), times(_lib.GPK_TIMED_POTRF)
    fbe.check(fbe.lib.gpk_timing(fbe.h, 0))
    assert len(grad_ms) >= B and np.isfinite(lml).all() and np.isfinite(grad).all()
    step_s = dt / args.steps
    flops = B * float(N) ** 3
    Np = (N + 127) // 128 * 128
    gk = float(np.mean(grad_ms)) * 1e-3
    # SURVEY 8(d) prices the fused ARD pass at 2 N^2 s bytes (K and K^-1 once each).  This kernel needs less: K is recomputed
    # from X on the fly and K^-1 is symmetric, so what has to move is the lower tiles of K^-1 once - N^2 / 2 x 8 bytes (plus
    # the diagonal tiles' upper halves).  `achieved` counts THOSE bytes (priced at the survey's figure the rate would exceed
    # the HBM peak, which says nothing); the survey's count is kept beside it.
    survey_bytes = 2.0 * N * N * 8
    alg_bytes = Np * (Np + 128) / 2 * 8
    tr = pmc_workload_traffic("lml", N)
    gkern = (tr or {}).get("by_kernel", {}).get("lml_grad_kernel")
    line = {
        "metric": "LML + gradient evaluations/s, three per-axis ARD GPs fused into one launch chain (BASELINE configs[4])",
        "value": B / step_s, "unit": "GP evaluations/s", "n_gpus": 1, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": step_s * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64",
        "data": "synthetic (SURVEY 8d generator), ARD l_d = 2.0 (1 + 0.1 d), noise 0.1, jitter 1e-4",
        "config": {"workload": f"C5: N_train={N}, D={D}, B={B} single-output ARD GPs on shared inputs, LML + gradient"},
        "roofline": {"bound": "mfma", "achieved": flops / step_s / 1e12, "peak": MFMA_F64_PEAK_TF, "unit": "TFLOP/s",
                     "frac": flops / step_s / 1e12 / MFMA_F64_PEAK_TF,
                     "traffic": tr["bytes_per_step"] if tr else None,
                     "pmc": ({"by_kernel": tr["by_kernel"], "source": tr["source"]} if tr else None),
                     "what": "B x N^3 fp64 flops (factor + inverse factor + W^T W per model) over the step's wall time"},
        "roofline_grad_kernel": {"kernel": "lml_grad_kernel", "bound": "hbm", "achieved": alg_bytes / gk / 1e9,
                                 "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": alg_bytes / gk / 1e9 / HBM_PEAK_GBPS,
                                 "algorithmic_bytes_per_launch": alg_bytes,
                                 "what": "the lower tiles of K^-1 read once (N^2 / 2 x 8 bytes); K_ij and its D per-feature factors "
                                         "are recomputed per entry on the fp64 vector ALU, which is what the launch is bound by",
                                 "survey_bytes_per_launch": survey_bytes, "ms_per_launch": gk * 1e3,
                                 "launches_averaged": int(len(grad_ms)),
                                 "traffic": (gkern["bytes_per_step"] / max(gkern["launches_per_step"], 1)) if gkern else None},
        "potrf_ms_all_models_one_launch": float(np.mean(potrf_ms)) if len(potrf_ms) else None,
        "lml": [float(v) for v in lml],
        "peak_hbm_bytes_per_rank": int(torch.cuda.max_memory_allocated(be.device)),
    }
    if not args.no_cpu_baseline:
        line["cpu_baseline"] = cpu_lml_baseline(N)
    print(json.dumps(line), flush=True)


def cpu_lml_baseline(n_full, n_sample=2048):
    """scikit-learn's own LML + gradient (one ARD model) on a bounded sample, N^3-extrapolated (labelled)."""
    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        from sklearn.gaussian_process import GaussianProcessRegressor as SkGPR
        from sklearn.gaussian_process.kernels import RBF as SkRBF, ConstantKernel as SkC, WhiteKernel as SkWhite
    except ImportError as e:
        return {"skipped": repr(e)}
    X, Y, _ = synthetic_problem(n_sample, 1)
    ls = 2.0 * (1.0 + 0.1 * np.arange(9))
    kern = SkC(1.0, "fixed") * SkRBF(ls, (0.1, 10.0)) + SkWhite(0.1, (1e-5, 1e1))
    g = SkGPR(kernel=kern, alpha=1e-4, normalize_y=True, optimizer=None).fit(X, Y[:, 0])
    t0 = time.perf_counter()
    g.log_marginal_likelihood(g.kernel_.theta, eval_gradient=True)
    dt = time.perf_counter() - t0
    scale = (n_full / n_sample) ** 3
    return {"value": 1.0 / (dt * scale), "unit": "GP evaluations/s", "cores": cores, "kind": "reference",
            "basis": "extrapolated", "rule": "measured seconds at the sample size x (N_train / sample N_train)^3",
            "measured_seconds": dt, "measured_at_n_train": n_sample,
            "sample": f"scikit-learn log_marginal_likelihood(theta, eval_gradient=True) of one ARD model at N_train={n_sample} "
                      f"({dt:.2f} s, {cores} threads), N^3-extrapolated to N_train={n_full}"}


def main():
    args = parse_args()
    # (BENCH_FORCE_LAUNCH=1 takes the same parent -> torch.distributed.run -> rank route with one rank: the way to
    # rehearse the launcher on a one-GPU box)
    if (args.gpus > 1 or os.environ.get("BENCH_FORCE_LAUNCH") == "1") and "RANK" not in os.environ:
        launch_ranks(args)                     # does not return

    # dmabuf IPC: required by RCCL on this host driver - set before torch / HIP come up, on BOTH launch routes (the
    # driver's own torch.distributed.run never passes through launch_ranks)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if torch.cuda.device_count() < (local_rank + 1):
        raise SystemExit(f"rank {rank}: no GPU {local_rank} visible ({torch.cuda.device_count()} devices)")
    torch.cuda.set_device(local_rank)
    # under torch.distributed.run (RANK set) the process group is always created, also for one rank,
    # so the single-GPU box exercises the same RCCL all-gather path the 2/4/8-GPU runs use
    use_dist = "RANK" in os.environ and "WORLD_SIZE" in os.environ
    ranks_seen = 1
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        # RCCL prints a version banner on stdout when its first communicator comes up; stdout must carry exactly
        # one JSON line, so file descriptor 1 points at stderr until the communicator exists
        sys.stdout.flush()
        saved = os.dup(1)
        os.dup2(2, 1)
        try:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
            ones = torch.ones(1, device=torch.device("cuda", local_rank))
            dist.all_reduce(ones)                                   # RCCL all-reduce: every rank really is there
            ranks_seen = int(round(float(ones.item())))
            dist.barrier()
            torch.cuda.synchronize()
        finally:
            sys.stdout.flush()
            os.dup2(saved, 1)
            os.close(saved)
        assert ranks_seen == dist.get_world_size() == world, (ranks_seen, dist.get_world_size(), world)

    from unmanned_aerial_vehicles_amd import _lib
    from unmanned_aerial_vehicles_amd.device import DeviceGP, get_backend
    from unmanned_aerial_vehicles_amd.sharded import all_gather_rows, patch_low_rows

    be = get_backend(local_rank)
    if args.workload == "gram":
        return bench_gram(args, be, rank, world, use_dist, ranks_seen)
    if args.workload in ("train", "lml"):          # single-GPU workloads (the factorisation does not shard: "replicas only")
        if world != 1:
            raise SystemExit(f"--workload {args.workload} runs on one GPU")
        return bench_train(args, be) if args.workload == "train" else bench_lml(args, be)
    c4 = args.workload == "c4"
    if c4:
        args.queries = (1 << 20) // world          # strong scaling: the 1 M queries are split over the ranks
    method = "inverse_split2" if args.var_method == "auto" else args.var_method
    N, M, D, P = args.n_train, args.queries, 9, 3
    X, Y, _ = synthetic_problem(N, 1)
    Yn = (Y - Y.mean(0)) / Y.std(0)
    y_mean, y_std = Y.mean(0), Y.std(0)
    ls, sf2, noise, jitter = 2.0, 1.0, 0.1, 1e-4
    # per-rank query batch (different seed per rank), resident in HBM before the timed region
    Xq = np.random.default_rng(1 + rank).standard_normal((M, D))
    q32 = torch.as_tensor(Xq, dtype=torch.float32, device=be.device)

    def sync_all():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    # ---------------------------------------------------------------- fit (set-up, timed separately)
    # Multi-GPU: the model is REPLICATED.  --replicate broadcast (default for mean + variance serving): rank 0 fits and
    # broadcasts what fp32 serving needs - X, alpha, the split inverse factor + scales (17 GB at N = 65 536, RCCL over
    # xGMI) - the other ranks never hold L or the fp64 inverse factor; --replicate refit: every rank
    # fits redundantly (deterministic, no communication).
    use_w = not c4 and method in ("inverse", "inverse_split", "inverse_split2")
    by_broadcast = world > 1 and args.replicate == "broadcast" and method == "inverse_split2" and not c4

    def fit_model():
        dev = DeviceGP(X, Yn, be)
        dev.timing(True)
        dev.gram(ls, sf2, noise + jitter)            # warm-up of the Gram kernel + allocation
        torch.cuda.synchronize()
        for _ in range(6):
            dev.gram(ls, sf2, noise + jitter)
        gram_times = dev.kernel_times(_lib.GPK_TIMED_GRAM)[-6:] * 1e-3     # HIP events around the Gram kernel launches
        gram_s = float(np.mean(gram_times))                                 # the average (what a rocprofv3 --stats row shows)
        gram_bytes = dev.Np * dev.Np * 8 + N * D * 8          # N^2 s + N D s (s = 8)
        info = C.c_int(0)
        t0 = time.perf_counter()
        be.check(be.lib.gpk_potrf(be.h, C.c_void_p(dev.K.data_ptr()), dev.Np, dev.Np, C.c_void_p(dev.winv.data_ptr()),
                                  C.byref(info)))
        torch.cuda.synchronize()
        potrf_s = time.perf_counter() - t0
        dev.factored = True
        if use_w:
            # the inverse factor (34 GB) and its scratch come from torch's caching allocator: map them once outside the
            # timed region (a first hipMalloc of this size takes up to a second on some boxes and is not kernel time)
            # (the fp16 x 2 operand is split straight from the fp64 inverse factor: no fp32 copy exists on that route;
            # the trtri scratch is gone again before the split operand is allocated, as in DeviceGP)
            # (in the order and with the lifetimes the product path has: the split operand is allocated while the inverse
            # factor is alive and after its scratch has gone - freed FIRST here, the 34 GB block would be cut up to serve
            # the 17 GB request and the timed region would pay a fresh hipMalloc of 17 GB, 0.5 s, for the split operand)
            warm = [be.empty((dev.Np, dev.Np), torch.float64), be.empty(((dev.Np // 2 + 128) ** 2,), torch.float64)]
            if method != "inverse_split2":
                warm.append(be.empty((dev.Np, dev.Np), torch.float32))
            warm.pop(1)
            warm.append(be.empty((dev.Np * dev.Np * (6 if method == "inverse_split" else 4),), torch.uint8))
            del warm
            if method == "inverse_split2":
                # (and the split's two kernels once on one tile: a kernel's first launch in a process pays its lazy load,
                # ~0.7 ms here - not part of what the split costs)
                w1 = torch.eye(128, dtype=torch.float64, device=be.device)
                s1 = torch.ones((1,), dtype=torch.float32, device=be.device)
                d1 = be.empty((128 * 128 * 4,), torch.uint8)
                be.check(be.lib.gpk_split2_rows_f64_absmax(be.h, C.c_void_p(w1.data_ptr()), 128, 128, C.c_void_p(s1.data_ptr()),
                                                           C.c_void_p(d1.data_ptr())))
                del w1, s1, d1
            torch.cuda.synchronize()
        t0 = time.perf_counter()
        trtri_s = None
        if use_w:
            dev.inverse_factor(False)                  # W = L^-1 on the fp64 MFMA (N^3/3 flops) ...
            torch.cuda.synchronize()
            trtri_s = time.perf_counter() - t0
            if method == "inverse_split":
                dev.split_inverse_factor()             # ... served as three exact bf16 parts per entry (6 bytes)
            elif method == "inverse_split2":
                dev.split2_inverse_factor()            # ... served as two fp16 parts per entry (4 bytes)
            else:
                dev.inverse_factor(True)               # ... served as an fp32 copy
        elif not c4:
            dev._f32_factor()                          # fp32 copies of L / leaf inverses
        torch.cuda.synchronize()
        prep_s = time.perf_counter() - t0
        t0 = time.perf_counter()
        dev.solve_alpha()                              # two launches through W when it exists, else the solve chain
        torch.cuda.synchronize()
        alpha_s = time.perf_counter() - t0
        dev._f32_data()
        if method == "inverse_split2":
            # alpha and the fp16 x 2 serving operand exist: the fp64 inverse factor (34 GB at N = 65 536) has done its work.
            # (The fp64 variance of the parity check below then runs as the blocked solve with L - slower, untimed.)
            dev._Winv.pop("f64", None)
        torch.cuda.synchronize()
        fit = {"n_train": N, "dtype": "f64",
               "gram_ms": gram_s * 1e3, "gram_ms_min_avg_max": [float(gram_times.min() * 1e3), gram_s * 1e3, float(gram_times.max() * 1e3)],
               "gram_launches_timed": int(len(gram_times)), "gram_GBps": gram_bytes / gram_s / 1e9,
               "gram_frac_of_hbm_peak": gram_bytes / gram_s / 1e9 / HBM_PEAK_GBPS,
               "cholesky_s": potrf_s, "cholesky_GFLOPs": N ** 3 / 3.0 / potrf_s / 1e9,
               "cholesky_frac_of_f64_mfma_peak": N ** 3 / 3.0 / potrf_s / 1e12 / MFMA_F64_PEAK_TF,
               "alpha_solve_ms": alpha_s * 1e3,
               "variance_prep": "none (means only)" if c4 else
                                {"inverse_split": "explicit inverse factor W = L^-1 (N^3/3 flops, fp64 MFMA) + exact bf16x3 split",
                                 "inverse_split2": "explicit inverse factor W = L^-1 (N^3/3 flops, fp64 MFMA) + fp16x2 split",
                                 "inverse": "explicit inverse factor W = L^-1 (N^3/3 flops, fp64 MFMA) + fp32 copy",
                                 "solve": "fp32 copy of L"}[method],
               "variance_prep_s": prep_s, "trtri_s": trtri_s,
               "variance_split_ms": ((prep_s - trtri_s) * 1e3) if trtri_s else None,    # the split: ONE pass over W (the block maxima come out of the inverse factor's product epilogues)
               "trtri_GFLOPs": (N ** 3 / 3.0 / trtri_s / 1e9) if trtri_s else None,
               "replicated_per_rank": world > 1}
        return dev, fit

    if by_broadcast:
        from unmanned_aerial_vehicles_amd.sharded import broadcast_state
        dev, fit = fit_model() if rank == 0 else (None, None)
        torch.cuda.synchronize()
        dist.barrier()
        t0 = time.perf_counter()
        meta, tens = dev.serving_state() if rank == 0 else (None, None)
        meta, tens = broadcast_state(meta, tens, DeviceGP.SERVING_TENSORS, 0, None,
                                     make_empty=lambda shape, dt: be.empty(shape, dt))
        torch.cuda.synchronize()
        dist.barrier()
        bcast_s = time.perf_counter() - t0
        if rank != 0:
            dev = DeviceGP.from_serving_state(meta, tens, be)
            dev._f32_data()
        else:
            fit["replication"] = "broadcast"
            fit["broadcast_s"] = bcast_s
            fit["broadcast_bytes"] = int(sum(t.numel() * t.element_size() for t in tens.values()))
    else:
        dev, fit = fit_model()
        fit["replication"] = "refit" if world > 1 else "none (one rank)"

    # ---------------------------------------------------------------- the timed hot path
    kss = sf2 + noise
    ystd2 = torch.as_tensor(y_std ** 2, device=be.device, dtype=torch.float64)
    assert c4 or dev.fp32_mean_ok(q32), "the benchmark batch must pass the fp32 mean gate (it is served in fp32)"

    ag_events = []                                 # (start, end) event pairs around every all-gather of the timed steps
    low_patched = [0]
    src_rank = 0 if by_broadcast else None
    thr_low = dev.FP32_VAR_RECHECK_FRACTION * kss * float(y_std[0] ** 2)

    def gathered(rows):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        g = all_gather_rows(rows, M * world)                                                # RCCL all-gather
        e1.record()
        ag_events.append((e0, e1))
        return g

    rank_queries = {rank: q32}

    def queries_of(r):                             # the resident fp32 batch of rank r, regenerated from its seed (rank 0's checks)
        if r not in rank_queries:
            rank_queries[r] = torch.as_tensor(np.random.default_rng(1 + r).standard_normal((M, D)), dtype=torch.float32, device=be.device)
        return rank_queries[r]

    def recompute_low(rows):                       # rank 0 only (it holds the factor): fp64 variances of the gathered rows `rows`
        q = torch.stack([queries_of(int(g) // M)[int(g) % M] for g in rows.tolist()]).double().contiguous()
        v = dev.predict_var_dev(q, kss, 0.0, "float64", dev._fp64_var_method())
        return v[:, None] * ystd2[None, :]

    def step_c4():
        mean = dev.predict_mean_dev(q32, y_mean, y_std, "float32")                 # K4 only
        return gathered(mean) if use_dist else mean

    def step_c3():
        # K4 + K* + K5 + finalise (un-normalise, pack [mean | var], count the rows the fp32 variance gate must recompute):
        # libgpk launches only - the production serving path, gates included.  One rank: DeviceGP.predict_packed_dev (mean
        # gate on the batch, variance re-check).  Several ranks: what ShardedPredictor.predict_mean_var does - the mean gate
        # decided for ALL ranks (own shard + all-reduce MIN), the packed serving call, ONE all-gather, then the cross-rank
        # variance gate (patch_low_rows: no collective while no row is low).
        if not use_dist:
            return dev.predict_packed_dev(q32, y_mean, y_std, kss, 0.0, "float32", method)  # (M, 2P) float64
        flag = torch.tensor([1 if dev.fp32_mean_ok(q32) else 0], dtype=torch.int32, device=be.device)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        gate = bool(int(flag.item()))
        assert gate, "the benchmark batch must pass the fp32 mean gate on every rank"
        out = gathered(dev.predict_packed_dev(q32, y_mean, y_std, kss, 0.0, "float32", method, True, gate))
        if by_broadcast:
            out, n = patch_low_rows(out, P, thr_low, recompute_low, src_rank, None)
            low_patched[0] += n
        return out

    step = step_c4 if c4 else step_c3
    for _ in range(args.warmup):
        step()
    sync_all()
    dev.timing(True)                               # restart the event ring: it now covers exactly the timed steps
    ag_events.clear()
    low_patched[0] = 0
    sensors = BoardSensors(be.device.index if be.device.index is not None else 0) if rank == 0 else None
    if sensors:
        sensors.start()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = step()
    sync_all()
    dt = time.perf_counter() - t0
    board = sensors.stop() if sensors else None
    k5_ms = dev.kernel_times(_lib.GPK_TIMED_K5)    # the dominant launch of every timed step (the last 64 of them)
    dev.timing(False)
    if use_dist:
        t = torch.tensor([dt], dtype=torch.float64, device=be.device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    assert bool(torch.isfinite(out).all()), "non-finite predictions"
    # the collective's own share of a step: HIP events around every all-gather of the timed steps (this rank's view; the MAX
    # over ranks is what bounds the step)
    allgather_ms = None
    if use_dist and ag_events:
        mine_ms = float(np.mean([a.elapsed_time(b) for a, b in ag_events]))
        t = torch.tensor([mine_ms], dtype=torch.float64, device=be.device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        allgather_ms = {"rank0_avg": mine_ms, "max_over_ranks_avg": float(t.item()), "collectives_per_step": 1,
                        "bytes_per_rank_sent": int(out.shape[1] * M * out.element_size()),
                        "low_variance_rows_patched": int(low_patched[0])}

    # ---------------------------------------------------------------- parity of the timed path (outside the timed region)
    # rank 0's own shard of the last step against the fp64 kernels on the same batch: the stated fp32 bars
    # (mean 1e-4, std 1e-3, relative) at the headline shape itself
    parity = None
    v64 = None
    if rank == 0:
        mine = out[:M].double() if use_dist else out.double()
        m64 = dev.predict_mean_dev(q32.double(), y_mean, y_std, "float64")
        if c4:
            e_mean = float((mine - m64).abs().max() / m64.abs().max())
            parity = {"mean_max_rel_err_vs_fp64": e_mean, "mean_tol": 1e-4, "queries_checked": M, "ok": e_mean < 1e-4}
        else:
            v64 = dev.predict_var_dev(q32.double(), kss, 0.0, "float64", "inverse" if (use_w and "f64" in dev._Winv) else "solve")
            s64 = torch.sqrt(v64[:, None] * ystd2[None, :])
            e_mean = float((mine[:, :P] - m64).abs().max() / m64.abs().max())
            e_std = float(((torch.sqrt(mine[:, P:]) - s64).abs() / s64).max())
            parity = {"mean_max_rel_err_vs_fp64": e_mean, "std_max_rel_err_vs_fp64": e_std, "mean_tol": 1e-4,
                      "std_tol": 1e-3, "queries_checked": M, "ok": e_mean < 1e-4 and e_std < 1e-3}
        assert parity["ok"], f"timed path disagrees with the fp64 path: {parity}"
        if use_dist and world > 1 and not c4:
            # one shard that ANOTHER rank served (after a broadcast: a replica without factor) against rank 0's fp64 kernels
            q1 = queries_of(1).double()
            theirs = out[M:2 * M].double()
            m1 = dev.predict_mean_dev(q1, y_mean, y_std, "float64")
            v1 = dev.predict_var_dev(q1, kss, 0.0, "float64", "inverse" if (use_w and "f64" in dev._Winv) else "solve")
            s1 = torch.sqrt(v1[:, None] * ystd2[None, :])
            e1m = float((theirs[:, :P] - m1).abs().max() / m1.abs().max())
            e1s = float(((torch.sqrt(theirs[:, P:]) - s1).abs() / s1).max())
            parity["rank1_shard"] = {"mean_max_rel_err_vs_fp64": e1m, "std_max_rel_err_vs_fp64": e1s,
                                     "served_by": "replica (broadcast)" if by_broadcast else "refit"}
            assert e1m < 1e-4 and e1s < 1e-3, f"rank 1's shard disagrees with the fp64 path: {parity['rank1_shard']}"
    if not c4:
        dev._Winv.pop("f64", None)                 # 34 GB back before the extras

    # ---------------------------------------------------------------- roofline of the dominant kernel
    roof = None
    if rank == 0 and c4:
        # K4: algorithmic flops M N (3D + 2P + 8) against the fp32 vector peak.  The MFMA kernel moves the 3D
        # distance flops to the (bf16) matrix pipe, so the ratio is NOT a roofline fraction of one pipe: what bounds
        # the kernel is the vector ALU's exp + P FMAs per pair; `frac` is therefore null.
        flops = float(M) * N * (3 * D + 2 * P + 8)
        kern = dev.mean_kernel_choice()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        dev.predict_mean_dev(q32, y_mean, y_std, "float32")      # (the parity check above ran fp64 kernels: warm again)
        a.record()
        for _ in range(5):
            dev.predict_mean_dev(q32, y_mean, y_std, "float32")
        b.record()
        torch.cuda.synchronize()
        k4_s = a.elapsed_time(b) * 1e-3 / 5
        valu_flops = float(M) * N * (2 * P + 8)          # exp2 (counted 8) + P FMAs per pair stay on the vector ALU
        roof = {"bound": "valu",
                "kernel": "mean_bf16_kernel<3,2> (distances: 6 x v_mfma_f32_32x32x16_bf16 per 32x32 block, exact bf16x3 "
                          "operand split, training operand prepared once per call; exp2 + P FMAs per pair on the VALU)" if kern == "mfma"
                else "predict_mean_kernel<float,3,1> (exact differences on the VALU)",
                "achieved": (valu_flops if kern == "mfma" else flops) / k4_s / 1e12, "peak": MFMA_F32_PEAK_TF,
                "unit": "TFLOP/s",
                "frac": (valu_flops if kern == "mfma" else flops) / k4_s / 1e12 / MFMA_F32_PEAK_TF,
                "note": "vector-ALU flops only (exp2 counted as 8 + 2P per pair); the 3D distance flops per pair run "
                        "on the bf16 matrix pipe" if kern == "mfma" else "all flops on the vector ALU",
                "algorithmic_TFLOPs_all_pipes": flops / k4_s / 1e12,
                "traffic": (pmc_workload_traffic("c4", N) or {}).get("bytes_per_step") if world == 1 else None,
                "traffic_source": (pmc_workload_traffic("c4", N) or {}).get("source") if world == 1 else None,
                "k4_ms": k4_s * 1e3, "algorithmic_flops_per_step": flops}
    if rank == 0 and not c4:
        flops = float(N) * float(N) * float(M)      # N^2 M algorithmic (fp32-equivalent) flops per step (K5)
        # per-STEP kernel time: the bracketed launches of one step summed (a batch larger than the variance panel is
        # several launches; the ring keeps the last 64 launches, so only whole steps still in it are used)
        lps = max(1, -(-M // max(128, min(dev.VAR_PANEL_MAX, (dev.VAR_PANEL_BYTES // (dev.Np * 4)) // 128 * 128))))
        if method == "solve":                       # the chain's launches are not bracketed one by one: time the call
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            dev.predict_var_dev(q32, kss, 0.0, "float32", "solve")
            b.record()
            torch.cuda.synchronize()
            step_ms = np.array([a.elapsed_time(b)])
        else:
            whole = (len(k5_ms) // lps) * lps
            assert whole >= lps and len(k5_ms) == min(64, args.steps * lps), (len(k5_ms), args.steps, lps)
            step_ms = k5_ms[len(k5_ms) - whole:].reshape(-1, lps).sum(axis=1)
        k5_s = float(np.mean(step_ms)) * 1e-3
        if method == "inverse_split":
            # every fp32-equivalent multiply-add is six bf16 MFMA multiply-adds (a0b0, a0b1, a1b0, a1b1, a0b2, a2b0)
            key, peak, per_product = "k5_split_kernel", MFMA_BF16_PEAK_TF, 6.0
            kernel = ("k5_split_kernel<4> (V = W K*^T, fp32 operands as 3 exact bf16 parts, 6 x v_mfma_f32_32x32x16_bf16 per "
                      "32x32x16 block product, fp32 accumulation, fused column-norm epilogue, 1 launch/step)")
        elif method == "inverse_split2":
            key, peak, per_product = "k5_direct_kernel", MFMA_BF16_PEAK_TF, 3.0        # fp16 MFMA: the same rate as bf16
            kernel = ("k5_direct_kernel<4> (512 x 128 tiles, 128 x 128 per wave, one wave per SIMD; V = W K*^T, fp32 operands as 2 "
                      "round-to-nearest fp16 parts in fragment order, loaded from L2 straight into registers (no LDS); 3 x "
                      "v_mfma_f32_32x32x16_f16 per 32x32x16 block product, fp32 accumulation, fused column-norm epilogue, "
                      "1 launch/step)")
        else:
            key, peak, per_product = "gemm_kernel_f32_epi1", MFMA_F32_PEAK_TF, 1.0
            kernel = ("gemm_kernel<float,false,false,1> (V = W K*^T on v_mfma_f32_32x32x2_f32 with fused column-norm "
                      "epilogue, 1 launch/step)" if method == "inverse" else
                      "gemm_kernel<float,false,true,0> (all launches of the triangular solve)")
        tr = pmc_traffic(key, N, M)
        roof = {"bound": "mfma", "kernel": kernel,
                # `achieved` / `frac` are ALGORITHMIC: N^2 M fp32-equivalent flops over the kernel time, against the dense peak
                # of the pipe the kernel runs on; `frac_issued` counts the MFMA products actually issued (x 3 / x 6)
                "achieved": flops / k5_s / 1e12, "peak": peak, "unit": "TFLOP/s",
                "frac": flops / k5_s / 1e12 / peak,
                "frac_algorithmic": flops / k5_s / 1e12 / peak,
                "frac_issued": per_product * flops / k5_s / 1e12 / peak,
                "frac_of_fp32_mfma_peak": flops / k5_s / 1e12 / MFMA_F32_PEAK_TF,
                "pipe": {"inverse_split": "bf16 MFMA", "inverse_split2": "fp16 MFMA"}.get(method, "fp32 MFMA"),
                "algorithmic_flops_per_launch": flops / lps,
                "issued_flops_per_launch": per_product * flops / lps,
                "mfma_products_per_fp32_product": per_product,
                "algorithmic_flops_note": "N^2 M fp32-equivalent flops per step (SURVEY 8d: N^2 per prediction)",
                "traffic": tr["bytes_per_launch"] if tr else None,
                "traffic_source": tr["source"] if tr else None,
                "operand_bytes_per_launch": (float(N) * N * 2 + float(N) * M * 4) if method == "inverse_split2" else None,
                # committed counter passes of this kernel at this shape (not collected in this process): how busy the matrix
                # pipe is and the clock the chip holds under it - the product of the two is what `frac_issued` can reach
                "pmc": ({k: tr[k] for k in ("pmc_mfma_busy", "pmc_clock_ghz", "pmc_l2_hit_rate", "pmc_source") if k in tr}
                        or None) if tr else None,
                # measured in THIS run: the board's power and shader clock over the timed steps
                "board": board,
                "launches_per_step": lps if method != "solve" else 2 * (dev.Np // 128) - 1,
                "kernel_ms": k5_s * 1e3, "kernel_ms_min_max": [float(np.min(step_ms)), float(np.max(step_ms))],
                "timed_launches": int(len(k5_ms)), "timed_steps": int(len(step_ms)),
                "timing": "HIP events recorded by the library around the launch on its stream, over the timed steps"}

    # host-boundary rate (not the headline): queries start in host memory, results end in host memory
    host_api = None
    if rank == 0 and world == 1 and not c4:
        xq_host = np.ascontiguousarray(Xq, dtype=np.float32)
        ts = []
        for _ in range(3):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            qd = torch.from_numpy(xq_host).to(be.device)
            res = dev.predict_packed_dev(qd, y_mean, y_std, kss, 0.0, "float32", method).cpu().numpy()
            ts.append(time.perf_counter() - t0)
        host_api = {"ms_per_batch": min(ts) * 1e3, "predictions_per_s": M / min(ts),
                    "note": "PCIe-inclusive: 10 000 x 9 fp32 queries host->HBM, (10 000 x 6) fp64 results HBM->host"}
        assert np.isfinite(res).all()

    # ---------------------------------------------------------------- extras: the other fp32 forms of the same launch
    # (never the headline; same batch, same factor; each with its own error against the fp64 kernels)
    extras = None
    if rank == 0 and world == 1 and not c4 and method == "inverse_split2" and not args.no_extras:
        extras = {}
        try:
            dev._Winv.pop("split2", None)
            dev.inverse_factor(True)               # fp32 copy of W (recomputes W on the fp64 MFMA: untimed)
            dev._Winv.pop("f64", None)

            def time_form(m):
                dev.predict_var_dev(q32, kss, 0.0, "float32", m)
                torch.cuda.synchronize()
                dev.timing(True)
                ts = []
                for _ in range(3):
                    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    a.record()
                    dev.predict_mean_dev(q32, y_mean, y_std, "float32")
                    v = dev.predict_var_dev(q32, kss, 0.0, "float32", m)
                    b.record()
                    torch.cuda.synchronize()
                    ts.append(a.elapsed_time(b) * 1e-3)
                km = dev.kernel_times(_lib.GPK_TIMED_K5)
                dev.timing(False)
                t = sorted(ts)[1]
                e = float(((torch.sqrt(v) - torch.sqrt(v64)).abs() / torch.sqrt(v64)).max()) if v64 is not None else None
                return t, float(np.mean(km)), e

            t32, k32, e32 = time_form("inverse")
            extras["fp32_mfma_path"] = {
                "what": "mean + variance with the variance GEMM on v_mfma_f32_32x32x2_f32 (exact fp32 products; round 1's "
                        "headline path; --var-method inverse)",
                "ms_per_step": t32 * 1e3, "predictions_per_s": M / t32, "kernel_ms": k32,
                "fp32_mfma_TFLOPs": float(N) * N * M / (k32 * 1e-3) / 1e12,
                "frac_of_fp32_mfma_peak": float(N) * N * M / (k32 * 1e-3) / 1e12 / MFMA_F32_PEAK_TF,
                "std_max_rel_err_vs_fp64": e32}
            dev.split_inverse_factor()
            dev._Winv.pop("f32", None)
            t3, k3, e3 = time_form("inverse_split")
            extras["bf16x3_split_path"] = {
                "what": "mean + variance with the variance GEMM on v_mfma_f32_32x32x16_bf16: fp32 operands as 3 exact bf16 "
                        "parts, 6 products per block (--var-method inverse_split)",
                "ms_per_step": t3 * 1e3, "predictions_per_s": M / t3, "kernel_ms": k3,
                "bf16_mfma_TFLOPs": 6.0 * float(N) * N * M / (k3 * 1e-3) / 1e12,
                "frac_of_bf16_mfma_peak": 6.0 * float(N) * N * M / (k3 * 1e-3) / 1e12 / MFMA_BF16_PEAK_TF,
                "std_max_rel_err_vs_fp64": e3}
            extras["std_max_rel_err_vs_fp64_of_the_timed_path"] = parity["std_max_rel_err_vs_fp64"] if parity else None
            dev._Winv.pop("split", None)
        except Exception as e:  # noqa: BLE001 - an extra must never take the headline down
            extras["error"] = repr(e)

    peak_all = [int(torch.cuda.max_memory_allocated(be.device))]
    if use_dist:
        gathered = [None] * world
        dist.all_gather_object(gathered, peak_all[0])
        peak_all = [int(v) for v in gathered]
    if rank == 0:
        total_pred = float(M) * world * args.steps
        line = {
            "metric": "GP predictions/sec (mean+var) at N_train=65536, D=9" if not c4 else
                      "GP predictions/sec (posterior means, 1M queries sharded) at N_train=65536, D=9",
            "value": total_pred / dt, "unit": "predictions/s",
            "n_gpus": world, "ranks_seen": ranks_seen, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "strong" if c4 else "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": (f"C3: N_train={N}, D={D}, P={P}, batched predict mean+var over {M} query points "
                                    f"per GPU per step (horizon 20 x 500 rollouts), fp32 predict on an fp64 factor")
                                   if not c4 else
                                   (f"C4: N_train={N}, D={D}, P={P}, {M * world} queries sharded over {world} GPU(s), "
                                    f"posterior means, fp32, all-gather of the means"),
                       "n_train": N, "features": D, "outputs": P, "queries_per_gpu_per_step": M,
                       "variance_path": None if c4 else method,
                       "arithmetic": "fp32 operands and accumulation" +
                                     ("; products on the bf16 MFMA pipe from an exact 3-way bf16 split of every fp32 "
                                      "operand (error class of the fp32 MFMA, checked under \"parity\")"
                                      if (method == "inverse_split" and not c4) else
                                      ("; products on the fp16 MFMA pipe: every fp32 operand as two round-to-nearest fp16 parts "
                                       "(a0 + a1 = a to 2^-23 at worst), block products a1 b0 + a0 b1 + a0 b0 (the dropped a1 b1 "
                                       "is below 2^-22 |a b|; measured error equal to the exact-fp32 MFMA launch's: checked under "
                                       "\"parity\" and, for all three fp32 forms, under \"extras\")"
                                       if (method == "inverse_split2" and not c4) else "")),
                       "parallelism": f"query-sharded x{world}, model replicated (" +
                                      ("rank 0 fits, RCCL broadcast of X, alpha and the split inverse factor" if by_broadcast
                                       else "every rank fits redundantly") + ")" +
                                      ((", RCCL all-gather of the means" if c4 else ", RCCL all-gather of [mean|var]")
                                       if use_dist else "")},
            "roofline": roof,
            "peak_hbm_bytes_per_rank": int(torch.cuda.max_memory_allocated(be.device)),
            "peak_hbm_bytes_all_ranks": peak_all,
            "parity": parity,
            "fit": fit, "allgather_ms": allgather_ms,
            "host_api": host_api,
            "extras": extras,
        }
        if not args.no_cpu_baseline and world == 1 and not c4:     # reported baseline: rank 0 at N=1 only
            cb = cpu_baseline(n_full=N)
            try:
                full = cpu_full_size_predict(dev, X, y_mean, y_std, dev.alpha_host(), ls, noise)
                if "mean" in full:
                    # the CPU result doubles as a parity check AT THE HEADLINE SIZE against scikit-learn itself
                    # (`_gpr.py:441-494` on the same factor): posterior mean and standard deviation of the fp64 kernels
                    # (bar 1e-8) and of the timed fp32 serving call (bars 1e-4 / 1e-3) on the same 200 queries
                    Xq_t = torch.as_tensor(full["Xq"], device=be.device, dtype=torch.float64)
                    mg = dev.predict_mean_dev(Xq_t, y_mean, y_std, "float64").cpu().numpy()
                    vg = dev.predict_var_dev(Xq_t, kss, 0.0, "float64", "solve")
                    sg = torch.sqrt(vg[:, None] * ystd2[None, :]).cpu().numpy()
                    o32 = dev.predict_packed_dev(Xq_t.float(), y_mean, y_std, kss, 0.0, "float32", method).double().cpu().numpy()
                    ref_m, ref_s = full["mean"], full["std"]
                    rel = lambda a, b: float(np.max(np.abs(a - b)) / np.max(np.abs(b)))          # noqa: E731
                    vs = {"queries": int(ref_m.shape[0]), "n_train": N,
                          "fp64_mean_max_rel_err": rel(mg, ref_m),
                          "fp64_std_max_rel_err": float(np.max(np.abs(sg - ref_s) / ref_s)), "fp64_tol": 1e-8,
                          "fp32_mean_max_rel_err": rel(o32[:, :P], ref_m),
                          "fp32_std_max_rel_err": float(np.max(np.abs(np.sqrt(o32[:, P:]) - ref_s) / ref_s)),
                          "fp32_mean_tol": 1e-4, "fp32_std_tol": 1e-3}
                    vs["ok"] = bool(vs["fp64_mean_max_rel_err"] < 1e-8 and vs["fp64_std_max_rel_err"] < 1e-8 and
                                    vs["fp32_mean_max_rel_err"] < 1e-4 and vs["fp32_std_max_rel_err"] < 1e-3)
                    line["parity"]["vs_sklearn_at_n_train"] = vs
                    assert vs["ok"], f"GPU predictions disagree with scikit-learn at N_train={N}: {vs}"
                    full["gpu_fp64_mean_max_rel_err_vs_cpu"] = vs["fp64_mean_max_rel_err"]
                    full["gpu_fp64_std_max_rel_err_vs_cpu"] = vs["fp64_std_max_rel_err"]
                    for k in ("mean", "std", "Xq"):
                        full.pop(k)
                cb["measured_at_n_train"] = full
                if "value" in full:          # the host could hold the factor: the measured rate IS the like-for-like figure
                    cb["extrapolated_value"] = cb["value"]
                    cb["value"], cb["basis"] = full["value"], "measured"
                    cb["sample"] = (f"scikit-learn GaussianProcessRegressor.predict(return_std=True) of {full['queries']} queries at "
                                    f"N_train={N} on a regressor carrying the factor computed on the GPU (downloaded, untimed): "
                                    f"{full['predict_seconds']:.1f} s on {cb['cores']} threads")
            except AssertionError:
                raise
            except Exception as e:  # noqa: BLE001
                cb["measured_at_n_train"] = {"skipped": repr(e)}
            line["cpu_baseline"] = cb
        print(json.dumps(line), flush=True)
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
