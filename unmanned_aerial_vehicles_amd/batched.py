"""B independent single-output ARD GPs that share their training inputs (BASELINE config 5: the per-axis
ax/ay/az models of `src/px4/gp_trainer.py:139-179`, one `C(1,fixed)*RBF(ls in R^D)+WhiteKernel` GP per
output).

* training / hyper-parameter steps: the B factorisations are independent, so each model runs on its own
  libgpk handle and HIP stream from a worker thread (ctypes releases the GIL) — at the small and medium
  N of this use case a single model's launch chain cannot fill 256 CUs, the B chains overlap;
* prediction: ONE fused launch evaluates all B posterior means (`gpk_predict_mean_multi`): the feature
  differences of a (query, training point) pair are formed once and reused by every model.
"""
from __future__ import annotations

import ctypes as C
from concurrent.futures import ThreadPoolExecutor

import numpy as np

from . import _lib
from .device import Backend, get_backend
from .gpr import GaussianProcessRegressor
from .kernels import RBF, ConstantKernel, WhiteKernel


class BatchedARDGP:
    def __init__(self, length_scale=1.0, length_scale_bounds=(0.1, 10.0), noise_level=0.01,
                 noise_level_bounds=(1e-5, 1e1), alpha=1e-6, normalize_y=True, optimizer="fmin_l_bfgs_b",
                 n_restarts_optimizer=0, device=None, predict_dtype="float64", concurrent=True):
        self.length_scale, self.length_scale_bounds = length_scale, length_scale_bounds
        self.noise_level, self.noise_level_bounds = noise_level, noise_level_bounds
        self.alpha, self.normalize_y = alpha, normalize_y
        self.optimizer, self.n_restarts_optimizer = optimizer, n_restarts_optimizer
        self.device, self.predict_dtype, self.concurrent = device, predict_dtype, concurrent
        self.models = []
        self._workers = []
        self._fused = None
        self._fs = None
        self._serve = None

    def __getstate__(self):
        st = self.__dict__.copy()
        st["_workers"], st["_fused"], st["_fs"] = [], None, None   # handles, streams, device tensors: rebuilt lazily
        st["_serve"] = None
        return st

    # ------------------------------------------------------------------ workers
    def _backend(self, b):
        import torch
        main = get_backend(self.device)
        if not self.concurrent:
            return main, None
        while len(self._workers) <= b:
            self._workers.append((Backend(main.device_index), torch.cuda.Stream(device=main.device)))
        return self._workers[b]

    def _map(self, fn, n):
        """Run fn(b) for b in range(n); concurrently (one stream + handle per model) when enabled."""
        import torch
        if not self.concurrent or n == 1:
            return [fn(b) for b in range(n)]

        def run(b):
            be, stream = self._backend(b)
            with torch.cuda.stream(stream):
                out = fn(b)
                stream.synchronize()
            return out

        for b in range(n):
            self._backend(b)
        torch.cuda.current_stream().synchronize()
        with ThreadPoolExecutor(max_workers=n) as ex:
            return list(ex.map(run, range(n)))

    def _kernel(self, D):
        ls = np.broadcast_to(np.asarray(self.length_scale, dtype=np.float64), (D,)).copy()
        return (ConstantKernel(1.0, constant_value_bounds="fixed") * RBF(ls, self.length_scale_bounds)
                + WhiteKernel(self.noise_level, self.noise_level_bounds))

    # ------------------------------------------------------------------ fit / LML
    def fit(self, X, Y):
        """Fit B single-output ARD GPs on the shared inputs X (N, D); Y is (N, B).

        With the L-BFGS-B optimiser and B > 1 the hyper-parameters of all models are optimised together: the
        objective is the sum of the B log-marginal likelihoods (block-separable, so the joint optimum is the
        per-model optimum) and every evaluation is ONE fused launch chain for all models.  Restarts draw a new
        start for every model at once and each model keeps its best run."""
        import scipy.optimize
        X = np.asarray(X, dtype=np.float64)
        Y = np.asarray(Y, dtype=np.float64).reshape(len(X), -1)
        B, D = Y.shape[1], X.shape[1]
        if B > 8:
            raise ValueError("at most 8 models per batch")
        joint = self.optimizer == "fmin_l_bfgs_b" and B > 1

        def one(b):
            be = self._backend(b)[0]
            g = GaussianProcessRegressor(kernel=self._kernel(D), alpha=self.alpha, normalize_y=self.normalize_y,
                                         optimizer=None if joint else self.optimizer,
                                         n_restarts_optimizer=self.n_restarts_optimizer,
                                         device=be, predict_dtype=self.predict_dtype)
            return g.fit(X, Y[:, b])

        self.models = self._map(one, B)
        self._fused = None
        self._fs = None
        if joint:
            nth = self.models[0].kernel_.n_dims
            bounds = np.tile(self.models[0].kernel_.bounds, (B, 1))

            def obj(flat):
                lml, grad = self._lml_fused(flat.reshape(B, nth), True)
                lml = np.where(np.isfinite(lml), lml, -1e25)          # a not-PD model: large penalty, zero gradient
                return -float(np.sum(lml)), -grad.reshape(-1)

            rng = np.random.mtrand._rand
            best_theta = self.thetas.copy()
            best_lml = np.full(B, -np.inf)
            starts = [self.thetas.reshape(-1)]
            starts += [rng.uniform(bounds[:, 0], bounds[:, 1]) for _ in range(self.n_restarts_optimizer)]
            for x0 in starts:
                res = scipy.optimize.minimize(obj, x0, method="L-BFGS-B", jac=True, bounds=bounds)
                th = res.x.reshape(B, nth)
                lml = self._lml_fused(th, False)
                better = lml > best_lml
                best_theta[better], best_lml[better] = th[better], lml[better]
            self.release_fused_buffers()

            def finish(b):
                g = self.models[b]
                g.kernel_.theta = best_theta[b]
                g.log_marginal_likelihood_value_ = float(best_lml[b])
                g._refactor()
                g._alpha_host = g._L_host = None
                return g

            self.models = self._map(finish, B)
        return self

    def log_marginal_likelihood(self, thetas, eval_gradient=False, fused=True):
        """thetas: (B, D+1) log-parameters [log ls_0..log ls_{D-1}, log noise] per model.
        Returns lml (B,) and, with eval_gradient, grad (B, D+1) — the hyper-parameter step of config 5.
        fused=True: the B factorisations, inversions and alpha solves share ONE launch chain (every kernel gets
        a batch grid dimension, `gpk_batch_begin`); fused=False: one chain per model on its own stream."""
        thetas = np.asarray(thetas, dtype=np.float64)
        if fused:
            return self._lml_fused(thetas, eval_gradient)

        def one(b):
            return self.models[b].log_marginal_likelihood(thetas[b], eval_gradient)

        res = self._map(one, len(self.models))
        if not eval_gradient:
            return np.array(res)
        return np.array([r[0] for r in res]), np.stack([r[1] for r in res])

    def _fused_state(self):
        """Stacked HBM buffers for the batched launch chain: K/L, leaf inverses, W = L^-1, K^-1, scratch."""
        import torch
        if getattr(self, "_fs", None) is None:
            m0 = self.models[0]
            for m in self.models:
                m._ensure_device()
            be = get_backend(self.device)
            B, N, D = len(self.models), m0._dev.N, m0._dev.D
            Np = m0._dev.Np
            tsz = (Np // 2 + 128) ** 2
            f64 = torch.float64
            # per-model rows of Yn / alpha are registered as batch buffers: gpk_batch_buffer wants strides that are
            # multiples of 16 bytes, so the rows are N rounded up to even long (an odd N - e.g. 241 of 302 samples
            # after the 80/20 split - otherwise fails with GPK_BAD_ARG)
            Ne = N + (N & 1)
            Yn = torch.zeros((B, Ne), dtype=f64, device=be.device)
            for b, m in enumerate(self.models):
                Yn[b, :N].copy_(m._dev.Yn[:, 0])
            self._fs = {
                "be": be, "B": B, "N": N, "Ne": Ne, "D": D, "Np": Np, "tsz": tsz,
                "X": m0._dev.X.to(be.device),
                "Yn": Yn,                                                                              # (B, Ne)
                "K": be.empty((B, Np, Np), f64), "winv": be.empty((B, Np, 128), f64),
                "W": be.empty((B, Np, Np), f64), "Kinv": be.empty((B, Np, Np), f64),
                "T": be.empty((B, tsz), f64), "alpha": be.empty((B, Ne), f64),
            }
        return self._fs

    def _lml_fused(self, thetas, eval_gradient):
        fs = self._fused_state()
        be, B, N, D, Np = fs["be"], fs["B"], fs["N"], fs["D"], fs["Np"]
        lib, dp = be.lib, _lib._dp
        comps = [self.models[b].kernel_.clone_with_theta(thetas[b]).components() for b in range(B)]
        ls = [np.ascontiguousarray(c.ls_vector(D)) for c in comps]
        p = lambda t: C.c_void_p(t.data_ptr())  # noqa: E731
        info = (C.c_int * B)()
        with be.lock:
            be.bind_stream()
            for b in range(B):       # K1 per model (one launch each)
                be.check(lib.gpk_gram(be.h, _lib.GPK_F64, p(fs["X"]), N, D, ls[b].ctypes.data_as(dp), comps[b].sf2,
                                      (comps[b].noise or 0.0) + float(self.alpha), p(fs["K"][b]), Np))
            be.check(lib.gpk_batch_begin(be.h, B))
            try:
                for name, row_bytes in (("K", Np * Np * 8), ("winv", Np * 128 * 8), ("W", Np * Np * 8),
                                        ("Kinv", Np * Np * 8), ("T", fs["tsz"] * 8), ("Yn", fs["Ne"] * 8),
                                        ("alpha", fs["Ne"] * 8)):
                    be.check(lib.gpk_batch_buffer(be.h, p(fs[name]), row_bytes))
                rc = lib.gpk_potrf(be.h, p(fs["K"]), Np, Np, p(fs["winv"]), info)          # K2, B problems
                if rc not in (_lib.GPK_OK, _lib.GPK_NOT_PD):
                    be.check(rc)
                be.check(lib.gpk_trtri(be.h, p(fs["K"]), Np, Np, p(fs["winv"]), p(fs["W"]), Np, p(fs["T"])))
                be.check(lib.gpk_potrs_inv(be.h, p(fs["W"]), Np, Np, p(fs["Yn"]), N, 1, p(fs["alpha"])))   # K3
                if eval_gradient:
                    be.check(lib.gpk_wtw(be.h, p(fs["W"]), Np, Np, p(fs["Kinv"]), Np))
            finally:
                lib.gpk_batch_end(be.h)
            lml = np.full(B, -np.inf)
            grad = np.zeros((B, thetas.shape[1]))
            terms = np.zeros(2)
            g = np.zeros(D + 2)
            for b in range(B):
                if info[b] != 0:       # not positive definite: (-inf, 0) as sklearn/_gpr.py:588-589
                    continue
                be.check(lib.gpk_lml_terms(be.h, p(fs["K"][b]), N, Np, p(fs["Yn"][b]), p(fs["alpha"][b]), 1,
                                           terms.ctypes.data_as(dp)))
                lml[b] = -0.5 * terms[1] - terms[0] - 0.5 * N * np.log(2.0 * np.pi)
                if eval_gradient:
                    be.check(lib.gpk_lml_grad(be.h, p(fs["X"]), N, D, ls[b].ctypes.data_as(dp), comps[b].sf2,
                                              comps[b].noise or 0.0, p(fs["alpha"][b]), 1, p(fs["Kinv"][b]), Np,
                                              g.ctypes.data_as(dp)))
                    grad[b] = comps[b].map_gradient(g, D)
        return (lml, grad) if eval_gradient else lml

    def release_fused_buffers(self):
        self._fs = None

    @property
    def thetas(self):
        return np.stack([m.kernel_.theta for m in self.models])

    # ------------------------------------------------------------------ predict
    def _build_fused(self):
        import torch
        m0 = self.models[0]
        for m in self.models:
            m._ensure_device()
        be = get_backend(self.device)
        # fp32 serving is gated as in the estimator (DeviceGP, "fp32 serving gates"): if any model's fp32 mean would leave
        # the stated 1e-4, the whole fused call runs on the fp64 kernels
        f32 = self.predict_dtype == "float32" and all(m._dev.fp32_mean_ok() for m in self.models)
        tdt = torch.float32 if f32 else torch.float64
        comps = [m.kernel_.components() for m in self.models]
        D = m0.n_features_in_
        with torch.cuda.device(be.device):
            torch.cuda.synchronize()
            alpha = torch.stack([m._dev.alpha[:, 0].to(be.device) for m in self.models], dim=1).to(tdt).contiguous()
            X = m0._dev.X.to(be.device).to(tdt).contiguous()
        self._fused = {
            "X": X, "alpha": alpha, "N": X.shape[0], "D": D, "tdt": tdt, "code": _lib.GPK_F32 if f32 else _lib.GPK_F64,
            "ls": np.ascontiguousarray(np.stack([c.ls_vector(D) for c in comps])),
            "sf2": np.ascontiguousarray([c.sf2 for c in comps], dtype=np.float64),
            "ym": np.ascontiguousarray([m._y_train_mean[0] for m in self.models], dtype=np.float64),
            "ys": np.ascontiguousarray([m._y_train_std[0] for m in self.models], dtype=np.float64),
        }

    MFMA_MIN_QUERIES = 1024      # below this the single fused launch wins on latency

    def predict_mean_dev(self, Xq):
        """All B posterior means (one fused launch, or one matrix-core launch per model for large fp32
        batches); returns a (M, B) device tensor."""
        import torch
        if self._fused is None:
            self._build_fused()
        f = self._fused
        be = get_backend(self.device)
        if isinstance(Xq, torch.Tensor):
            q = Xq.to(device=be.device, dtype=f["tdt"]).contiguous()
        else:
            q = be.upload(np.ascontiguousarray(Xq, dtype=np.float64), f["tdt"])
        M, B = q.shape[0], len(self.models)
        out = be.empty((M, B), f["tdt"])
        if M == 0:
            return out
        if (f["code"] == _lib.GPK_F32 and M >= self.MFMA_MIN_QUERIES
                and all(m._dev.mean_kernel_choice() == "mfma" for m in self.models)):
            # large fp32 batches: one matrix-core launch per model (distances on the bf16 MFMA pipe) beats the
            # fused vector-ALU kernel (0.115 vs 0.158 ms at N = 4096, 10 000 queries, B = 3)
            cols = [m._dev.predict_mean_dev(q, m._y_train_mean, m._y_train_std, "float32", "mfma") for m in self.models]
            return torch.cat(cols, dim=1)
        dp = _lib._dp
        with be.lock:
            be.bind_stream()
            be.check(be.lib.gpk_predict_mean_multi(
                be.h, f["code"], C.c_void_p(f["X"].data_ptr()), C.c_void_p(f["alpha"].data_ptr()), f["N"], f["D"], B,
                f["ls"].ctypes.data_as(dp), f["sf2"].ctypes.data_as(dp), f["ym"].ctypes.data_as(dp),
                f["ys"].ctypes.data_as(dp), C.c_void_p(q.data_ptr()), M, C.c_void_p(out.data_ptr())))
        return out

    # ------------------------------------------------------------------ control-loop batches
    SERVE_MAX_M = 32

    def _serve_state(self, want_std):
        """Argument block of `gpk_predict_host_multi` (one call, two launches for all models), or None when the
        models do not qualify: fp64 serving, single-output models on one device with the same training-set size,
        D <= 16, N <= 16384, at most 8 of them.  Rebuilt when a model has been refitted."""
        import torch
        ms = self.models
        if self.predict_dtype == "float32" or not (1 <= len(ms) <= 8):
            return None
        for m in ms:
            m._ensure_device()
        devs = [m._dev for m in ms]
        key = tuple((id(d), id(d.ls), d.factored) for d in devs)
        sv = getattr(self, "_serve", None)
        if sv is None or sv["key"] != key:
            d0 = devs[0]
            ok = all(d.factored and d.P == 1 and d.N == d0.N and d.D == d0.D and d.be.device == d0.be.device for d in devs)
            if not ok or d0.D > 16 or d0.Np > 16384:
                self._serve = {"key": key, "ok": False}
                return None
            comps = [m.kernel_.components() for m in ms]
            B = len(ms)
            vp = C.c_void_p * B
            sv = self._serve = {
                "key": key, "ok": True, "B": B, "dev0": d0, "keep": devs,
                "X": vp(*[d.X.data_ptr() for d in devs]), "alpha": vp(*[d.alpha.data_ptr() for d in devs]),
                "ls": np.ascontiguousarray(np.stack([d.ls for d in devs])),
                "sf2": np.ascontiguousarray([d.sf2 for d in devs], dtype=np.float64),
                "ym": np.ascontiguousarray([m._y_train_mean[0] for m in ms], dtype=np.float64),
                "ys": np.ascontiguousarray([m._y_train_std[0] for m in ms], dtype=np.float64),
                "kss": np.ascontiguousarray([c.sf2 + (c.noise or 0.0) for c in comps], dtype=np.float64),
                "W": None,
            }
            with torch.cuda.device(d0.be.device):
                torch.cuda.synchronize()          # the models may have been fitted on their own streams
        if not sv["ok"]:
            return None
        if want_std and sv["W"] is None:
            if not all(d.host_path_ok(1, True) for d in sv["keep"]):
                return None
            Ws = [d.inverse_factor(False) for d in sv["keep"]]
            with torch.cuda.device(sv["dev0"].be.device):
                torch.cuda.synchronize()
            sv["Wt"] = Ws
            sv["W"] = (C.c_void_p * sv["B"])(*[w.data_ptr() for w in Ws])
        return sv

    def predict_host(self, Xq, return_std=False):
        """<= 32 rows, all models, one C call: (mean (M, B), std (M, B) or None); None if the models do not qualify."""
        sv = self._serve_state(return_std)
        if sv is None:
            return None
        Xq = np.ascontiguousarray(Xq, dtype=np.float64)
        M, B, d0 = Xq.shape[0], sv["B"], sv["dev0"]
        if Xq.ndim != 2 or Xq.shape[1] != d0.D:
            raise ValueError(f"queries must be (M, {d0.D})")
        mean = np.empty((B, M))
        var = np.empty((B, M)) if return_std else None
        be = d0.be
        with be.lock:
            be.bind_stream()
            be.check(be.lib.gpk_predict_host_multi(
                be.h, B, sv["X"], sv["alpha"], d0.N, d0.D, sv["ls"].ctypes.data, sv["sf2"].ctypes.data,
                sv["ym"].ctypes.data, sv["ys"].ctypes.data, sv["W"] if return_std else None, d0.Np, d0.Np,
                sv["kss"].ctypes.data, 0.0, Xq.ctypes.data, M, mean.ctypes.data,
                var.ctypes.data if return_std else None))
        if not return_std:
            return mean.T, None
        return mean.T, (np.sqrt(var) * sv["ys"][:, None]).T

    def predict(self, Xq, return_std=False):
        Xq = np.atleast_2d(np.asarray(Xq, dtype=np.float64))
        if 1 <= Xq.shape[0] <= self.SERVE_MAX_M:
            out = self.predict_host(Xq, return_std)
            if out is not None:
                return out if return_std else out[0]
        mean = self.predict_mean_dev(Xq).double().cpu().numpy()
        if not return_std:
            return mean
        std = np.stack(self._map(lambda b: self.models[b].predict(Xq, return_std=True)[1], len(self.models)), axis=1)
        return mean, std
