"""Device-side GP model: owns the HBM buffers (as torch tensors — torch is used for device
memory and streams only) and drives the HIP kernels through the libgpk C ABI.

Layout in HBM (row-major, Np = N rounded up to 128, identity in the padding):
    X      (N, D)   f64   training inputs                     [+ f32 copy for the fp32 predict path]
    Yn     (N, P)   f64   normalised targets
    K / L  (Np, Np) f64   Gram matrix, overwritten in place by its lower Cholesky factor
    winv   (Np,128) f64   inverses of the 128x128 diagonal blocks of L
    alpha  (N, P)   f64   K^-1 Yn
    Xf, alphaf, Lf, winvf f32 copies, built on the first fp32 predict
    Kinv, W (Np, Np) f64  only while hyper-parameter gradients are being evaluated
"""
from __future__ import annotations

import ctypes as C
import os
import threading

import numpy as np

from . import _lib
from ._lib import GPK_F32, GPK_F64, GPKError, NotPositiveDefinite

_backends = {}
_backends_lock = threading.Lock()


def _torch():
    import torch
    return torch


class Backend:
    """One libgpk handle per GPU, bound to torch's current stream on that device."""

    def __init__(self, device_index=0):
        torch = _torch()
        self.lib = _lib.load()
        if not torch.cuda.is_available():
            raise RuntimeError("no HIP device visible: the GP kernels need an MI355X (there is no CPU fallback)")
        self.device_index = int(device_index)
        self.device = torch.device("cuda", self.device_index)
        torch.cuda.set_device(self.device)
        h = C.c_void_p()
        rc = self.lib.gpk_create(C.byref(h), self.device_index)
        if rc != _lib.GPK_OK:
            raise GPKError(rc, "gpk_create failed")
        self.h = h
        self.lock = threading.RLock()
        if os.environ.get("GPK_DEBUG_FILL"):      # debugging aid (the test suite sets it): poison the handle's scratch too
            self.check(self.lib.gpk_set_option(self.h, b"debug_fill", 1))
        self.bind_stream()
        # running count of the rows the fp32 variance gate sent back (the packed finalise kernel adds to it and nobody
        # resets it: a serving call reads it before and after under the lock, so no fill kernel runs per call)
        self.low_count = torch.zeros((1,), dtype=torch.int32, device=self.device)
        self.low_seen = 0

    def bind_stream(self):
        torch = _torch()
        s = torch.cuda.current_stream(self.device).cuda_stream
        self.check(self.lib.gpk_set_stream(self.h, C.c_void_p(s)))

    def check(self, rc):
        if rc == _lib.GPK_OK:
            return
        msg = self.lib.gpk_last_error(self.h).decode()
        if rc == _lib.GPK_NOT_PD:
            raise NotPositiveDefinite(msg)
        raise GPKError(rc, msg)

    def sync(self):
        self.check(self.lib.gpk_synchronize(self.h))

    def set_options(self, **opts):
        """Tuning knobs of the handle (include/gpk.h, gpk_set_option): `be.set_options(ptile=0, trtri_levels=0)`; a str value
        goes to gpk_set_option_str.  The library itself reads nothing from the environment."""
        for k, v in opts.items():
            if isinstance(v, str):
                self.check(self.lib.gpk_set_option_str(self.h, k.encode(), v.encode()))
            else:
                self.check(self.lib.gpk_set_option(self.h, k.encode(), int(v)))
        return self

    def empty(self, shape, dtype):
        t = _torch().empty(shape, dtype=dtype, device=self.device)
        if os.environ.get("GPK_DEBUG_FILL"):      # debugging aid: poison fresh buffers (reads of unwritten memory show up as NaN)
            t.fill_(float(os.environ["GPK_DEBUG_FILL"]) if t.is_floating_point() else 255)   # (bytes 0xFF: NaN as bf16 pairs)
        return t

    def upload(self, a, dtype=None):
        torch = _torch()
        t = torch.from_numpy(np.ascontiguousarray(a))
        if dtype is not None:
            t = t.to(dtype)
        return t.to(self.device)


def get_backend(device_index=None) -> Backend:
    """Shared per-GPU backend; a `Backend` instance passes through (private handle + stream, used by
    the concurrent per-model workers of `BatchedARDGP`)."""
    torch = _torch()
    if isinstance(device_index, Backend):
        return device_index
    if device_index is None:
        device_index = torch.cuda.current_device() if torch.cuda.is_available() else 0
    with _backends_lock:
        b = _backends.get(device_index)
        if b is None:
            b = _backends[device_index] = Backend(device_index)
        return b


_workers = {}


def worker_backends(device_index, n):
    """n private (Backend, torch stream) pairs on a GPU, created once and reused: independent pieces of work - the restarts of
    the estimator's optimiser, the per-axis models of BatchedARDGP - run on them from host threads (ctypes releases the GIL),
    so that one piece's latency-bound launches (a factorisation at the reference's sizes keeps 5-20 % of the matrix pipe busy)
    overlap the other's."""
    torch = _torch()
    main = get_backend(device_index)
    with _backends_lock:
        pool = _workers.setdefault(main.device_index, [])
        while len(pool) < n:
            pool.append((Backend(main.device_index), torch.cuda.Stream(device=main.device)))
        return pool[:n]


def padded(n):
    return (int(n) + 127) // 128 * 128


def _p(t):
    return C.c_void_p(t.data_ptr())


class DeviceGP:
    """Training set + factorisation resident on one GPU."""

    # largest scratch (bytes) used for one variance panel  V = L^-1 K*^T
    VAR_PANEL_BYTES = 6 << 30
    VAR_PANEL_MAX = 16384
    # up to this padded size the inverse factor W = L^-1 is formed right away (cheap), so that alpha and
    # every later variance / gradient call are single launches
    INVERSE_EAGER_NP = 32768
    # up to this padded size factor + inverse factor are one launch given an (Np, Np) scratch (the library's ptile_inv_max_np)
    FUSED_INVERSE_NP = 4608
    MAX_FEATURES = 16

    def __init__(self, X, Yn, backend: Backend | None = None):
        torch = _torch()
        self.be = backend or get_backend()
        X = np.ascontiguousarray(X, dtype=np.float64)
        Yn = np.ascontiguousarray(Yn, dtype=np.float64)
        if X.ndim != 2 or Yn.ndim != 2 or X.shape[0] != Yn.shape[0]:
            raise ValueError("X must be (N, D) and Y (N, P)")
        self.N, self.D = X.shape
        self.P = Yn.shape[1]
        if not (1 <= self.D <= self.MAX_FEATURES):
            # gpk_gram alone takes up to GPK_MAX_D = 64 features; the fused mean, the cross-Gram panel, the small-batch
            # serving kernels and the LML gradient are compiled for D <= 16 (the reference's largest model has 16
            # inputs, gaussian_process.py:66): refuse at construction rather than fit a model that cannot predict
            raise ValueError(f"D must be in [1, {self.MAX_FEATURES}] on the MI355X path (got {self.D})")
        if not (1 <= self.P <= _lib.GPK_MAX_P):
            raise ValueError(f"P must be in [1, {_lib.GPK_MAX_P}]")
        self.Np = padded(self.N)
        self._Xh = X                      # host copy: spread statistics that gate the MFMA mean kernel
        self._xc = X.mean(axis=0)
        self._r2 = {}
        self.X = self.be.upload(X)
        self.Yn = self.be.upload(Yn)
        self.K = None       # (Np, Np) f64: Gram, then L
        self.winv = None
        self.alpha = self.be.empty((self.N, self.P), torch.float64)
        self.factored = False
        self.ls = None
        self.sf2 = None
        self._f32 = None    # dict of fp32 copies: X, alpha [, L, winv]
        self._Winv = {}     # explicit inverse factor L^-1: {'f64': tensor} and/or {'f32': tensor}
        self._host_args = None   # predict_host: cached argument addresses
        self._amp = self._alpha_sq = None         # fp32 mean gate: cached amplification estimate
        self._Kinv = None
        self.replica = False     # True: a serving replica built by from_serving_state (no factor, no fp64 inverse factor)

    # ---- fit-side -------------------------------------------------------------------------
    def _ensure_K(self):
        torch = _torch()
        if self.K is None:
            self.K = self.be.empty((self.Np, self.Np), torch.float64)
            self.winv = self.be.empty((self.Np, 128), torch.float64)

    def gram(self, ls, sf2, diag_add):
        """K1: build the padded Gram matrix in HBM."""
        self._ensure_K()
        ls = np.ascontiguousarray(np.broadcast_to(np.asarray(ls, dtype=np.float64), (self.D,)))
        be = self.be
        with be.lock:
            be.bind_stream()
            be.check(be.lib.gpk_gram(be.h, GPK_F64, _p(self.X), self.N, self.D, ls.ctypes.data_as(_lib._dp),
                                     float(sf2), float(diag_add), _p(self.K), self.Np))
        self.ls, self.sf2 = ls, float(sf2)
        self.factored = False
        self._f32 = None
        self._Winv = {}

    def factorize(self, ls, sf2, diag_add):
        """K1 + K2: Gram build and in-place blocked Cholesky.  Raises NotPositiveDefinite."""
        self.gram(ls, sf2, diag_add)
        be = self.be
        info = C.c_int(0)
        with be.lock:
            be.check(be.lib.gpk_potrf(be.h, _p(self.K), self.Np, self.Np, _p(self.winv), C.byref(info)))
        self.factored = True

    def load_factor(self, L, ls, sf2):
        """Import an existing lower factor (host, N x N), e.g. from a scikit-learn pickle."""
        torch = _torch()
        self._ensure_K()
        L = np.asarray(L, dtype=np.float64)
        Kh = torch.eye(self.Np, dtype=torch.float64)
        Kh[: self.N, : self.N] = torch.from_numpy(np.tril(L))
        self.K.copy_(Kh)
        be = self.be
        with be.lock:
            be.bind_stream()
            be.check(be.lib.gpk_leaf_inverses(be.h, _p(self.K), self.Np, self.Np, _p(self.winv)))
        self.ls = np.ascontiguousarray(np.broadcast_to(np.asarray(ls, dtype=np.float64), (self.D,)))
        self.sf2 = float(sf2)
        self.factored = True
        self._f32 = None
        self._Winv = {}

    def solve_alpha(self, method="auto"):
        """K3: alpha = L^-T L^-1 Yn.  "chain": recursive blocked solves with L; "inverse": W^T (W Yn)
        with the explicit inverse factor (two launches); "auto": inverse if W is already at hand or
        cheap to form (Np <= INVERSE_EAGER_NP), else chain."""
        assert self.factored
        be = self.be
        if method == "auto":
            method = "inverse" if ("f64" in self._Winv or self.Np <= self.INVERSE_EAGER_NP) else "chain"
        if method == "inverse":
            W = self.inverse_factor(False)
            with be.lock:
                be.bind_stream()
                be.check(be.lib.gpk_potrs_inv(be.h, _p(W), self.Np, self.Np, _p(self.Yn), self.N, self.P,
                                              _p(self.alpha)))
        else:
            with be.lock:
                be.bind_stream()
                be.check(be.lib.gpk_potrs(be.h, _p(self.K), self.Np, self.Np, _p(self.winv), _p(self.Yn), self.N,
                                          self.P, _p(self.alpha)))
        self._amp = self._alpha_sq = None
        if self._f32:
            self._f32.pop("alpha", None)       # the fp32 copy of alpha is stale; X / L copies stay valid

    def set_alpha(self, alpha):
        self.alpha.copy_(self.be.upload(np.asarray(alpha, dtype=np.float64).reshape(self.N, self.P)))
        self._amp = self._alpha_sq = None
        if self._f32:
            self._f32.pop("alpha", None)

    def lml_terms(self):
        """K6a: (sum log diag L, [y_p . alpha_p])."""
        be = self.be
        out = np.zeros(1 + self.P)
        with be.lock:
            be.bind_stream()
            be.check(be.lib.gpk_lml_terms(be.h, _p(self.K), self.N, self.Np, _p(self.Yn), _p(self.alpha), self.P,
                                          out.ctypes.data_as(_lib._dp)))
        return out[0], out[1:]

    def lml_grad(self, noise):
        """K6b: [dLML/dlog ls_d ..., dLML/dlog noise, dLML/dlog sf2] at the current factor/alpha."""
        torch = _torch()
        assert self.factored
        if self.D > 16:
            raise ValueError("analytic LML gradients support D <= 16")
        if self._Kinv is None:
            self._Kinv = self.be.empty((self.Np, self.Np), torch.float64)
        W = self.inverse_factor(False)
        be = self.be
        g = np.zeros(self.D + 2)
        with be.lock:
            be.bind_stream()
            be.check(be.lib.gpk_wtw(be.h, _p(W), self.Np, self.Np, _p(self._Kinv), self.Np))
            be.check(be.lib.gpk_lml_grad(be.h, _p(self.X), self.N, self.D, self.ls.ctypes.data_as(_lib._dp),
                                         self.sf2, float(noise), _p(self.alpha), self.P, _p(self._Kinv), self.Np,
                                         g.ctypes.data_as(_lib._dp)))
        return g

    def lml_eval(self, ls, sf2, diag_add, noise, eval_gradient=True):
        """One optimiser evaluation as one chain with one synchronisation (gpk_lml_eval): K1, K2, W = L^-1, alpha, K6a and -
        eval_gradient - K^-1 and K6b; the model is left factored at these hyper-parameters.  Returns (log-det / 2,
        [y_p . alpha_p], gradient as lml_grad or None).  Raises NotPositiveDefinite.  For Np <= INVERSE_EAGER_NP (where
        alpha goes through the inverse factor anyway)."""
        torch = _torch()
        if eval_gradient and self.D > 16:
            raise ValueError("analytic LML gradients support D <= 16")
        self._ensure_K()
        ls = np.ascontiguousarray(np.broadcast_to(np.asarray(ls, dtype=np.float64), (self.D,)))
        be = self.be
        self._f32 = None
        self._Winv = {}
        self.factored = False
        W = be.empty((self.Np, self.Np), torch.float64)
        work = be.empty(((self.Np // 2 + 128) ** 2,), torch.float64)
        # (K^-1 buffer: the gradient's, and - up to FUSED_INVERSE_NP rows - the scratch that lets factor and inverse factor run as
        # ONE persistent launch also in a value-only evaluation: the final fit at the selected hyper-parameters)
        want_kinv = eval_gradient or self.Np <= self.FUSED_INVERSE_NP
        if want_kinv and self._Kinv is None:
            self._Kinv = be.empty((self.Np, self.Np), torch.float64)
        terms = np.zeros(1 + self.P)
        g = np.zeros(self.D + 2) if eval_gradient else None
        info = C.c_int(0)
        with be.lock:
            be.bind_stream()
            self.ls, self.sf2 = ls, float(sf2)
            be.check(be.lib.gpk_lml_eval(
                be.h, _p(self.X), self.N, self.D, ls.ctypes.data_as(_lib._dp), float(sf2), float(diag_add), float(noise),
                _p(self.Yn), self.P, _p(self.K), self.Np, _p(self.winv), _p(W), _p(work), _p(self.alpha),
                _p(self._Kinv) if want_kinv else None, terms.ctypes.data_as(_lib._dp),
                g.ctypes.data_as(_lib._dp) if eval_gradient else None, C.byref(info)))
        self.factored = True
        self._Winv["f64"] = W
        self._amp = self._alpha_sq = None
        return terms[0], terms[1:], g

    def release_grad_buffers(self):
        self._Kinv = None

    # ---- host views ------------------------------------------------------------------------
    def L_host(self):
        return np.tril(self.K[: self.N, : self.N].cpu().numpy())

    def alpha_host(self):
        return self.alpha.cpu().numpy()

    # ---- predict-side ----------------------------------------------------------------------
    def _f32_data(self):
        """fp32 copies of X and alpha (enough for the fused mean)."""
        torch = _torch()
        if self._f32 is None:
            self._f32 = {}
        if "X" not in self._f32:
            self._f32["X"] = self.X.to(torch.float32)
        if "alpha" not in self._f32:
            self._f32["alpha"] = self.alpha.to(torch.float32)
        return self._f32

    def _f32_factor(self):
        """fp32 copies of the factor and its leaf inverses (variance path)."""
        torch = _torch()
        c = self._f32_data()
        if "L" not in c:
            assert self.factored
            be = self.be
            Lf = be.empty((self.Np, self.Np), torch.float32)
            wf = be.empty((self.Np, 128), torch.float32)
            with be.lock:
                be.bind_stream()
                be.check(be.lib.gpk_factor_to_f32(be.h, _p(self.K), self.Np, self.Np, _p(self.winv), _p(Lf),
                                                  self.Np, _p(wf)))
            c["L"], c["winv"] = Lf, wf
        return c

    def _as_queries(self, Xq, tdtype):
        torch = _torch()
        if isinstance(Xq, torch.Tensor):
            q = Xq.to(device=self.be.device, dtype=tdtype).contiguous()
        else:
            q = self.be.upload(np.ascontiguousarray(Xq, dtype=np.float64), tdtype)
        if q.ndim != 2 or q.shape[1] != self.D:
            raise ValueError(f"queries must be (M, {self.D})")
        return q

    def timing(self, enable=True):
        """Event brackets around the dominant launches (gpk_timing); `kernel_times(tag)` reads them back."""
        be = self.be
        with be.lock:
            be.bind_stream()
            be.check(be.lib.gpk_timing(be.h, 1 if enable else 0))

    def kernel_times(self, tag, max_n=64):
        be = self.be
        ms = np.zeros(max_n)
        n = C.c_int(0)
        with be.lock:
            be.check(be.lib.gpk_kernel_times(be.h, int(tag), ms.ctypes.data_as(_lib._dp), max_n, C.byref(n)))
        return ms[: n.value].copy()

    # the fp32 MFMA mean kernel expands |a - b|^2 = |a|^2 + |b|^2 - 2 a.b around the training mean; it is used
    # while the largest scaled squared norm (in exp2 units) stays below this bound (error in the exponent
    # ~ bound * 2^-22), the exact-difference kernel otherwise
    MFMA_MEAN_R2_MAX = 64.0

    def mean_kernel_choice(self):
        """"mfma" if the fp32 matrix-core mean kernel is admissible for the current length-scales."""
        key = self.ls.tobytes()
        if key not in self._r2:
            u = (self._Xh - self._xc) / self.ls
            self._r2 = {key: 0.5 * np.log2(np.e) * float(np.max(np.einsum("ij,ij->i", u, u)))}
        return "mfma" if (self.P <= 8 and self._r2[key] <= self.MFMA_MEAN_R2_MAX) else "valu"

    def predict_mean_dev(self, Xq, y_mean, y_std, dtype="float64", kernel="auto", _alpha=None, _kernel=None):
        """K4 on device tensors; returns a (M, P) tensor of `dtype`.  kernel: "valu" (exact differences on
        the vector ALU), "mfma" (fp32 only: distances on the matrix cores) or "auto" (mfma for fp32 when
        `mean_kernel_choice` admits it)."""
        torch = _torch()
        f32 = dtype in ("float32", np.float32, torch.float32)
        if kernel == "auto":
            kernel = self.mean_kernel_choice() if f32 else "valu"
        if kernel not in ("valu", "mfma") or (kernel == "mfma" and not f32):
            raise ValueError("kernel must be 'auto', 'valu' or 'mfma' (mfma: float32 only)")
        tdt = torch.float32 if f32 else torch.float64
        q = self._as_queries(Xq, tdt)
        M = q.shape[0]
        out = self.be.empty((M, self.P), tdt)
        if M == 0:
            return out
        if f32:
            c = self._f32_data()
            Xd, ad = c["X"], c["alpha"]
        else:
            Xd, ad = self.X, (self.alpha if _alpha is None else _alpha)     # (_alpha: fp64 weights other than alpha)
        ym = np.ascontiguousarray(np.broadcast_to(np.asarray(y_mean, dtype=np.float64), (self.P,)))
        ys = np.ascontiguousarray(np.broadcast_to(np.asarray(y_std, dtype=np.float64), (self.P,)))
        be = self.be
        ls_, sf2_ = (self.ls, self.sf2) if _kernel is None else (np.ascontiguousarray(_kernel[0]), float(_kernel[1]))
        if kernel == "mfma":
            with be.lock:
                be.bind_stream()
                be.check(be.lib.gpk_predict_mean_mfma(be.h, _p(Xd), _p(ad), self.N, self.D, self.P,
                                                      self.ls.ctypes.data_as(_lib._dp), self.sf2,
                                                      self._xc.ctypes.data_as(_lib._dp), ym.ctypes.data_as(_lib._dp),
                                                      ys.ctypes.data_as(_lib._dp), _p(q), M, _p(out)))
            return out
        with be.lock:
            be.bind_stream()
            be.check(be.lib.gpk_predict_mean(be.h, GPK_F32 if f32 else GPK_F64, _p(Xd), _p(ad), self.N, self.D,
                                             self.P, ls_.ctypes.data_as(_lib._dp), sf2_,
                                             ym.ctypes.data_as(_lib._dp), ys.ctypes.data_as(_lib._dp), _p(q), M,
                                             _p(out)))
        return out

    # queries per call up to which predict() goes through the one-call host path (gpk_predict_host)
    HOST_PATH_MAX_M = 256

    def host_path_ok(self, M, want_var):
        """The fp64 one-call serving path applies to small batches; its variance needs the fp64 inverse factor,
        which is at hand (or cheap to form) up to INVERSE_EAGER_NP."""
        if not (1 <= M <= self.HOST_PATH_MAX_M and self.D <= 16):
            return False
        return (not want_var) or ("f64" in self._Winv) or self.Np <= self.INVERSE_EAGER_NP

    def predict_host(self, Xq, y_mean, y_std, kss=None, floor=0.0):
        """One C call per batch: host queries (M, D) float64 -> (mean (M, P), var (M,) or None) as NumPy arrays
        (mean un-normalised, var in normalised-target units); kss=None skips the variance."""
        assert self.factored or kss is None
        Xq = np.ascontiguousarray(Xq, dtype=np.float64)
        M = Xq.shape[0]
        if Xq.ndim != 2 or Xq.shape[1] != self.D:
            raise ValueError(f"queries must be (M, {self.D})")
        mean = np.empty((M, self.P))
        var = np.empty((M,)) if kss is not None else None
        W = self.inverse_factor(False) if kss is not None else None
        be = self.be
        # the addresses that do not change between calls are looked up once per (factorisation, target scaling):
        # this call runs at the control rate and every ctypes conversion costs about a microsecond
        c = self._host_args
        if (c is None or c[0] is not self.X or c[1] is not self.alpha or c[2] is not self.ls or c[3] is not y_mean
                or c[4] is not y_std):
            ym = np.ascontiguousarray(np.broadcast_to(np.asarray(y_mean, dtype=np.float64), (self.P,)))
            ys = np.ascontiguousarray(np.broadcast_to(np.asarray(y_std, dtype=np.float64), (self.P,)))
            c = self._host_args = (self.X, self.alpha, self.ls, y_mean, y_std, ym, ys, self.X.data_ptr(),
                                   self.alpha.data_ptr(), self.ls.ctypes.data, ym.ctypes.data, ys.ctypes.data)
        with be.lock:
            be.bind_stream()
            be.check(be.lib.gpk_predict_host(be.h, c[7], c[8], self.N, self.D, self.P, c[9], self.sf2, c[10], c[11],
                                             W.data_ptr() if W is not None else None, self.Np, self.Np,
                                             float(kss) if kss is not None else 0.0, float(floor),
                                             Xq.ctypes.data, M, mean.ctypes.data,
                                             var.ctypes.data if var is not None else None))
        return mean, var

    def inverse_factor(self, f32):
        """W = L^-1 (lower, by tiles) on the fp64 MFMA; kept as fp64 or as an fp32 copy.  One-off
        N^3/3 flops per factorisation; makes every later variance call a single GEMM launch."""
        torch = _torch()
        assert self.factored
        key = "f32" if f32 else "f64"
        if key not in self._Winv:
            be = self.be
            W = self._Winv.get("f64")
            if W is None:
                W = be.empty((self.Np, self.Np), torch.float64)
                work = be.empty(((self.Np // 2 + 128) ** 2,), torch.float64)
                # (the row-block maxima the fp16 x 2 split needs come out of the products' epilogue: no pass over W for them)
                absmax = be.empty((self.Np // 128,), torch.float32)
                with be.lock:
                    be.bind_stream()
                    be.check(be.lib.gpk_trtri_absmax(be.h, _p(self.K), self.Np, self.Np, _p(self.winv), _p(W), self.Np,
                                                     _p(work), _p(absmax)))
                self._Winv["absmax"] = absmax
                del work
            if f32:
                Wf = be.empty((self.Np, self.Np), torch.float32)
                with be.lock:
                    be.bind_stream()
                    be.check(be.lib.gpk_tril_to_f32(be.h, _p(W), self.Np, self.Np, _p(Wf), self.Np))
                self._Winv["f32"] = Wf
            else:
                self._Winv["f64"] = W
        return self._Winv[key]

    def split_inverse_factor(self):
        """The fp32 inverse factor as three exact bf16 parts per entry (gpk_split3 layout, 6 bytes per entry):
        the operand of the bf16-pipe variance launch (`method="inverse_split"`)."""
        torch = _torch()
        if "split" not in self._Winv:
            had_f32 = "f32" in self._Winv
            Wf = self.inverse_factor(True)
            be = self.be
            W3 = be.empty((self.Np * self.Np * 6,), torch.uint8)
            with be.lock:
                be.bind_stream()
                be.check(be.lib.gpk_split3(be.h, _p(Wf), self.Np, self.Np, self.Np, _p(W3)))
            self._Winv["split"] = W3
            if not had_f32:
                self._Winv.pop("f32", None)      # the plain fp32 copy only served as the source of the split
        return self._Winv["split"]

    def split2_inverse_factor(self):
        """The fp32 inverse factor as two fp16 parts per entry in fragment order (gpk_split2_rows: 4 bytes per entry, one
        power-of-two scale per 128-row block, computed on the device): the operand of `method="inverse_split2"`."""
        torch = _torch()
        if "split2" not in self._Winv:
            be = self.be
            scales = be.empty((self.Np // 128,), torch.float32)
            W2 = be.empty((self.Np * self.Np * 4,), torch.uint8)
            if "f32" in self._Winv:              # an fp32 copy already exists (another variance form made it): split that
                with be.lock:
                    be.bind_stream()
                    be.check(be.lib.gpk_split2_rows(be.h, _p(self._Winv["f32"]), self.Np, self.Np, _p(scales), _p(W2)))
            else:                                # straight from the fp64 inverse factor: no fp32 copy, bit-identical parts
                W = self.inverse_factor(False)
                absmax = self._Winv.pop("absmax", None)
                with be.lock:
                    be.bind_stream()
                    if absmax is not None:       # gpk_trtri_absmax left the block maxima: one pass over W instead of two
                        scales = absmax
                        be.check(be.lib.gpk_split2_rows_f64_absmax(be.h, _p(W), self.Np, self.Np, _p(scales), _p(W2)))
                    else:
                        be.check(be.lib.gpk_split2_rows_f64(be.h, _p(W), self.Np, self.Np, _p(scales), _p(W2)))
            self._Winv["split2"] = (W2, scales)
        return self._Winv["split2"]

    # ---- replication by broadcast (multi-GPU serving, SURVEY.md 8(e): "broadcast once from rank 0") --------------------
    SERVING_TENSORS = ("X", "alpha", "W2", "w_scales")

    def serving_state(self):
        """Everything a rank needs to SERVE this model in fp32 (posterior mean + variance through the fp16 x 2 variance
        launch): the training inputs, alpha, the split inverse factor with its row-block scales (4 bytes per entry of the
        lower tiles: 17 GB at N = 65 536) and the hyper-parameters - not L, not the fp64 inverse factor.  Returns
        (meta dict of plain Python values, dict of device tensors named in SERVING_TENSORS)."""
        assert self.factored and not self.replica
        W2, w_scales = self.split2_inverse_factor()
        meta = {"N": self.N, "D": self.D, "P": self.P, "Np": self.Np, "ls": np.asarray(self.ls, dtype=np.float64).tolist(),
                "sf2": float(self.sf2), "shapes": {"X": (self.N, self.D), "alpha": (self.N, self.P),
                                                   "W2": (self.Np * self.Np * 4,), "w_scales": (self.Np // 128,)},
                "dtypes": {"X": "float64", "alpha": "float64", "W2": "uint8", "w_scales": "float32"}}
        return meta, {"X": self.X, "alpha": self.alpha, "W2": W2, "w_scales": w_scales}

    @classmethod
    def from_serving_state(cls, meta, tensors, backend=None):
        """A serving replica from `serving_state()` as received on another rank: fp32 mean and variance (and the fp64
        mean) work as on the fitting rank; whatever needs the factor (fp64 variance, LML, refit) does not exist here."""
        torch = _torch()
        self = cls.__new__(cls)
        self.be = backend or get_backend()
        self.N, self.D, self.P, self.Np = int(meta["N"]), int(meta["D"]), int(meta["P"]), int(meta["Np"])
        self.X = tensors["X"]
        self._Xh = self.X.cpu().numpy()
        self._xc = self._Xh.mean(axis=0)
        self._r2 = {}
        self.Yn = None
        self.K = self.winv = None
        self.alpha = tensors["alpha"]
        self.factored = True
        self.ls = np.ascontiguousarray(meta["ls"], dtype=np.float64)
        self.sf2 = float(meta["sf2"])
        self._f32 = None
        self._Winv = {"split2": (tensors["W2"], tensors["w_scales"])}
        self._host_args = None
        self._amp = self._alpha_sq = None
        self._Kinv = None
        self.replica = True
        return self

    # ---- fp32 serving gates -------------------------------------------------------------------------------------
    # The fp32 predict path is stated as: mean within 1e-4, std within 1e-3 (relative to the largest value) of the
    # fp64 path.  An fp32 kernel value carries the rounding of its exponent - an ulp of d^2/2 ~ 10 is 1e-6 - so every
    # term k*_j alpha_j of the mean is off by a few 1e-7 of itself, with random sign: the mean error is
    # c * sqrt(sum_j (k*_j alpha_j)^2) with c = 2.1e-7 .. 3.8e-7 for the matrix-core kernel and 3.1e-7 .. 8.9e-7 for the
    # exact-difference kernel (worst query of a batch of <= 2000, measured over 41 random models, N = 431 .. 65 536,
    # D = 1 .. 16, noise 1e-3 .. 0.3: tools/exp_fp32_gate.py, profiles/r02_fp32_gate_calibration.log and, with round 3's
    # kernels, r03_fp32_gate_calibration.log; the fp32 rounding of the inputs themselves accounts for about half of it;
    # the constants below are upper bounds of that wherever the amplification is >= 10).  Fine for the models the reference trains
    # (noise 0.03 - 0.3), not for sf2 N / noise ~ 1e7, where alpha is huge and cancels.  `fp32_mean_amplification`
    # measures A2 = max_m sqrt(sum_j (k_mj alpha_j)^2) / max_m |mean_m| on a sample of training rows (where it is
    # largest) once per alpha; the estimator routes a model with c * A2 above 1e-4 to the fp64 kernels (gpr.py).
    # Round 4 re-calibration (profiles/r04_fp32_gate_calibration.log: the same 41 models, batches of 131 072 queries, and the
    # N = 65 536 model with 1 048 576): the worst error of a batch grows with its size, and at these sizes err / A2 - A2 taken
    # where the queries are - reaches 6.6e-7 (matrix-core kernel) and 1.0e-6 (exact-difference kernel): the round-2 constants
    # (4.0e-7 / 9.0e-7, batches <= 2000) were too small for large batches (one D = 1 model passed the gate with a measured
    # error of 1.3e-4).  The constants below bound every case measured, and the gate has two levels: the MODEL passes if
    # c * A2 on its training rows - where the amplification is largest - is within the bar (then any batch is served in
    # fp32); otherwise the BATCH passes if c * 1.3 * A2 on up to 1024 of its own rows is (a reference-like model whose
    # queries are not at its training points: the benchmark model reads A2 = 172 on training rows, 58 on its queries).
    FP32_MEAN_ERR_PER_AMP = {"mfma": 7.0e-7, "valu": 1.1e-6}
    FP32_MEAN_TOL = 1e-4
    FP32_BATCH_GATE_ROWS = 1024
    FP32_BATCH_GATE_MARGIN = 1.3
    # Variance: |W k*|^2 in fp32 is off by 2e-7 .. 4e-5 of kss, growing as the noise shrinks (31 random models, all three
    # fp32 forms alike: profiles/r02_fp32_variance_forms_accuracy.log), i.e. the relative error of the standard
    # deviation is that over 2 var / kss: queries whose variance is below this fraction of the prior's - possible only
    # next to training points of a model with noise << sf2, which is also where the error is largest - are recomputed
    # in fp64.
    FP32_VAR_RECHECK_FRACTION = 1e-2

    def _amplification_at(self, q64):
        """A2 = max_m sqrt(sum_j (k_mj alpha_j)^2) / max_m |mean_m| over the rows of q64: two fp64 K4 launches - the second
        with the squared kernel (length-scales / sqrt 2, sf2^2) and squared weights."""
        zeros, ones = np.zeros(self.P), np.ones(self.P)
        sq = getattr(self, "_alpha_sq", None)      # (reset with _amp wherever alpha is rewritten: the library writes it in place)
        if sq is None:
            sq = self._alpha_sq = self.alpha ** 2
        b = self.predict_mean_dev(q64, zeros, ones, "float64", "valu").abs().amax(dim=0)
        a2 = self.predict_mean_dev(q64, zeros, ones, "float64", "valu", _alpha=sq,
                                   _kernel=(self.ls / np.sqrt(2.0), self.sf2 ** 2)).amax(dim=0).sqrt()
        return float((a2 / b.clamp_min(1e-300)).max())

    def fp32_mean_amplification(self):
        """A2 (see above) on <= 1024 evenly spaced training rows, cached per alpha."""
        torch = _torch()
        c = getattr(self, "_amp", None)            # (solve_alpha / set_alpha reset it: the library writes alpha in place)
        if c is not None and c[0] is self.ls:
            return c[1]
        idx = torch.linspace(0, self.N - 1, min(self.N, 1024), device=self.be.device).round().long()
        amp = self._amplification_at(self.X[idx].contiguous())
        self._amp = (self.ls, amp)
        return amp

    def fp32_mean_ok(self, q=None):
        """The fp32 mean gate.  q = None: the model-level answer (any queries).  q = the batch about to be served (device
        tensor or array): the model-level answer, or failing that the batch-level one."""
        torch = _torch()
        c = self.FP32_MEAN_ERR_PER_AMP[self.mean_kernel_choice()]
        if c * self.fp32_mean_amplification() <= self.FP32_MEAN_TOL:
            return True
        if q is None or len(q) == 0:
            return False
        M = len(q)
        # (no cache of the batch-level answer: a new tensor that the allocator places at a freed batch's address would
        # inherit its answer; the check is two K4 launches on <= 1024 rows, ~0.3 ms at N = 65 536 - a caller that serves ONE
        # resident batch many times asks once and passes `mean_gate=` to the predictors)
        if isinstance(q, torch.Tensor):
            idx = torch.linspace(0, M - 1, min(M, self.FP32_BATCH_GATE_ROWS), device=q.device).round().long()
            qs = q[idx].to(device=self.be.device, dtype=torch.float64).contiguous()
        else:
            idx = np.round(np.linspace(0, M - 1, min(M, self.FP32_BATCH_GATE_ROWS))).astype(np.int64)
            qs = self.be.upload(np.ascontiguousarray(np.asarray(q, dtype=np.float64)[idx]))
        return c * self.FP32_BATCH_GATE_MARGIN * self._amplification_at(qs) <= self.FP32_MEAN_TOL

    def predict_var_dev(self, Xq, kss, floor=0.0, dtype="float64", method="auto"):
        """K5 on device tensors; returns a (M,) float64 tensor (normalised-target units).

        method "solve": V = L^-1 K*^T by the blocked triangular solve (the reference's
        solve_triangular; a chain of 2 Np/128 - 1 GEMM launches); "inverse": |W k*|^2 with the
        explicit inverse factor W = L^-1, formed once per factorisation, in ONE fused GEMM launch (fp64 MFMA, or
        the exact-fp32 MFMA).  fp32 only - the same launch with the products on the 16-bit matrix pipe (fp32
        accumulation): "inverse_split2": every fp32 operand as two round-to-nearest fp16 parts (a0 + a1 = a to 2^-23 at
        worst; W scaled per 128-row block), block products a1 b0 + a0 b1 + a0 b0 (the dropped a1 b1 is below 2^-22 |a b|:
        at most 2^-21 |a b| per product in the worst case, ~2^-24 rms; measured over 31 random models: the same |W k*|^2
        error as the exact-fp32 MFMA, profiles/r02_fp32_variance_forms_accuracy.log; 2.9x its speed), operands in
        fragment order loaded from L2 straight into registers; "inverse_split": three bf16 parts (exact), six products
        per block (1.5x the fp32 MFMA's speed).  "auto": "inverse" for fp64, "inverse_split2" for fp32."""
        torch = _torch()
        assert self.factored
        f32 = dtype in ("float32", np.float32, torch.float32)
        if method == "auto":
            method = "inverse_split2" if f32 else "inverse"
        if method not in ("solve", "inverse", "inverse_split", "inverse_split2"):
            raise ValueError("method must be 'auto', 'solve', 'inverse', 'inverse_split' or 'inverse_split2'")
        if method in ("inverse_split", "inverse_split2") and not f32:
            raise ValueError("inverse_split / inverse_split2 are fp32 serving forms (bf16x3 / fp16x2 operand splits)")
        tdt = torch.float32 if f32 else torch.float64
        es = 4 if f32 else 8
        code = GPK_F32 if f32 else GPK_F64
        q = self._as_queries(Xq, tdt)
        M = q.shape[0]
        out = self.be.empty((M,), torch.float64)
        if M == 0:
            return out
        Xd = self._f32_data()["X"] if f32 else self.X
        if method == "inverse_split":
            W3 = self.split_inverse_factor()
        elif method == "inverse_split2":
            W2, w_scales = self.split2_inverse_factor()
        elif method == "inverse":
            Wd = self.inverse_factor(f32)
        elif f32:
            c = self._f32_factor()
            Ld, wd = c["L"], c["winv"]
        else:
            Ld, wd = self.K, self.winv
        panel = max(128, min(self.VAR_PANEL_MAX, (self.VAR_PANEL_BYTES // (self.Np * es)) // 128 * 128))
        panel = min(panel, padded(M))
        work = self.be.empty((self.Np * panel,), tdt) if method != "inverse_split2" else None     # (K* only ever exists split)
        work3 = (self.be.empty((self.Np * panel * (6 if method == "inverse_split" else 4),), torch.uint8)
                 if method in ("inverse_split", "inverse_split2") else None)
        var = self.be.empty((panel,), torch.float64)
        be = self.be
        lsp = self.ls.ctypes.data_as(_lib._dp)
        with be.lock:
            be.bind_stream()
            for m0 in range(0, M, panel):
                m1 = min(M, m0 + panel)
                if method == "inverse_split":
                    be.check(be.lib.gpk_predict_var_inv_split(be.h, _p(Xd), self.N, self.D, lsp, self.sf2, _p(W3),
                                                              self.Np, _p(q[m0:m1]), m1 - m0, float(kss), float(floor),
                                                              _p(work), _p(work3), _p(var)))
                elif method == "inverse_split2":
                    be.check(be.lib.gpk_predict_var_inv_split2(be.h, _p(Xd), self.N, self.D, lsp, self.sf2, _p(W2),
                                                               _p(w_scales), self.Np, _p(q[m0:m1]), m1 - m0, float(kss),
                                                               float(floor), _p(work3), _p(var)))
                elif method == "inverse":
                    be.check(be.lib.gpk_predict_var_inv(be.h, code, _p(Xd), self.N, self.D, lsp, self.sf2, _p(Wd),
                                                        self.Np, self.Np, _p(q[m0:m1]), m1 - m0, float(kss),
                                                        float(floor), _p(work), _p(var)))
                else:
                    be.check(be.lib.gpk_predict_var(be.h, code, _p(Xd), self.N, self.D, lsp, self.sf2, _p(Ld),
                                                    self.Np, self.Np, _p(wd), _p(q[m0:m1]), m1 - m0, float(kss),
                                                    float(floor), _p(work), _p(var)))
                out[m0:m1].copy_(var[: m1 - m0])
        return out

    # ---- gated serving: the one place every fp32 surface (estimator, sharded predictor, package GP, per-axis models) goes
    # through -------------------------------------------------------------------------------------------------------------
    def _rows64(self, Xq, q, rows):
        """The caller's own rows `rows` of a query batch as a contiguous float64 device tensor (NOT the fp32-rounded copy
        the fp32 kernels were given: an fp64 re-check sees the queries the caller passed)."""
        torch = _torch()
        if isinstance(Xq, torch.Tensor):
            return Xq[rows.to(Xq.device)].to(device=self.be.device, dtype=torch.float64).contiguous()
        return self.be.upload(np.ascontiguousarray(np.asarray(Xq, dtype=np.float64)[rows.cpu().numpy()]))

    def _fp64_var_method(self):
        if self.replica:
            raise RuntimeError("a serving replica holds no factor: fp64 variances are computed on the rank that fitted the model")
        return "inverse" if ("f64" in self._Winv or self.Np <= self.INVERSE_EAGER_NP) else "solve"

    def _gate_mean(self, dtype, var_method, gated, q=None, mean_gate=None):
        """(predict dtype, variance method) after the mean gate: an fp32 request for a model - or, at the batch level, a
        batch `q` - whose fp32 mean would leave the stated 1e-4 is served by the fp64 kernels.  mean_gate (True / False):
        the answer of `fp32_mean_ok` as the caller already has it - the sharded predictor decides ONCE for the whole batch
        and for every rank, a serving loop over one resident batch asks once."""
        torch = _torch()
        f32 = dtype in ("float32", np.float32, torch.float32)
        if f32 and gated and not (self.fp32_mean_ok(q) if mean_gate is None else mean_gate):
            return "float64", ("auto" if var_method in ("inverse_split", "inverse_split2") else var_method)
        return ("float32" if f32 else "float64"), var_method

    def predict_gated_dev(self, Xq, y_mean, y_std, kss=None, floor=0.0, dtype="float64", var_method="auto", gated=True,
                          mean_gate=None):
        """K4 (+ K5 when `kss` is given) with the fp32 serving gates applied: (mean (M, P) tensor of the dtype it was
        computed in, var (M,) float64 tensor in normalised-target units, or None).  dtype "float32" is a REQUEST: the mean gate (`fp32_mean_ok`) may
        route the model to the fp64 kernels, and single queries whose fp32 variance is below FP32_VAR_RECHECK_FRACTION of
        the prior's are recomputed by the fp64 launch.  gated=False: the raw fp32 kernels (tests, A/B timings)."""
        torch = _torch()
        pd, vm = self._gate_mean(dtype, var_method, gated, Xq, mean_gate)
        q = self._as_queries(Xq, torch.float32 if pd == "float32" else torch.float64)
        mean = self.predict_mean_dev(q, y_mean, y_std, pd)
        if kss is None:
            return mean, None
        var = self.predict_var_dev(q, kss, floor, pd, vm)
        if gated and pd == "float32" and q.shape[0]:
            low = torch.nonzero(var < self.FP32_VAR_RECHECK_FRACTION * kss).ravel()
            if low.numel():
                var[low] = self.predict_var_dev(self._rows64(Xq, q, low), kss, floor, "float64", self._fp64_var_method())
        return mean, var

    def predict_packed_dev(self, Xq, y_mean, y_std, kss, floor=0.0, dtype="float32", var_method="auto", gated=True,
                           mean_gate=None):
        """One serving step, the whole result in one (M, 2P) float64 device tensor: row m = [mean_m | var_m y_std^2]
        (un-normalised: what the all-gather of a sharded batch moves).  The fp32 default is ONE C call per panel
        (gpk_predict_mean_var_split2: K4, K* in split form, the variance launch, a finalising kernel that un-normalises,
        packs and counts the rows the variance gate must recompute); no torch arithmetic runs unless that count is
        non-zero.  Other dtypes / methods: the separate launches and gpk_pack_mean_var."""
        torch = _torch()
        pd, vm = self._gate_mean(dtype, var_method, gated, Xq, mean_gate)
        f32 = pd == "float32"
        if vm == "auto":
            vm = "inverse_split2" if f32 else "inverse"
        q = self._as_queries(Xq, torch.float32 if f32 else torch.float64)
        M = q.shape[0]
        out = self.be.empty((M, 2 * self.P), torch.float64)
        if M == 0:
            return out
        ym = np.ascontiguousarray(np.broadcast_to(np.asarray(y_mean, dtype=np.float64), (self.P,)))
        ys = np.ascontiguousarray(np.broadcast_to(np.asarray(y_std, dtype=np.float64), (self.P,)))
        be = self.be
        if not (f32 and vm == "inverse_split2"):
            mean = self.predict_mean_dev(q, ym, ys, pd)
            var = self.predict_var_dev(q, kss, floor, pd, vm)
            if gated and f32:
                low = torch.nonzero(var < self.FP32_VAR_RECHECK_FRACTION * kss).ravel()
                if low.numel():
                    var[low] = self.predict_var_dev(self._rows64(Xq, q, low), kss, floor, "float64", self._fp64_var_method())
            with be.lock:
                be.bind_stream()
                be.check(be.lib.gpk_pack_mean_var(be.h, GPK_F32 if f32 else GPK_F64, _p(mean), _p(var), M, self.P,
                                                  ys.ctypes.data_as(_lib._dp), _p(out)))
            return out
        c = self._f32_data()
        W2, w_scales = self.split2_inverse_factor()
        center = self._xc.ctypes.data_as(_lib._dp) if self.mean_kernel_choice() == "mfma" else None
        panel = max(128, min(self.VAR_PANEL_MAX, (self.VAR_PANEL_BYTES // (self.Np * 4)) // 128 * 128))
        panel = min(panel, padded(M))
        work2 = be.empty((self.Np * panel * 4,), torch.uint8)
        mean_tmp = be.empty((panel * self.P,), torch.float32)
        thr = self.FP32_VAR_RECHECK_FRACTION * kss if gated else 0.0
        nlow = 0
        with be.lock:
            be.bind_stream()
            try:
                for m0 in range(0, M, panel):
                    m1 = min(M, m0 + panel)
                    be.check(be.lib.gpk_predict_mean_var_split2(
                        be.h, _p(c["X"]), _p(c["alpha"]), self.N, self.D, self.P, self.ls.ctypes.data_as(_lib._dp), self.sf2,
                        center, ym.ctypes.data_as(_lib._dp), ys.ctypes.data_as(_lib._dp), _p(W2), _p(w_scales), self.Np,
                        _p(q[m0:m1]), m1 - m0, float(kss), float(floor), _p(work2), _p(mean_tmp), float(thr),
                        _p(be.low_count) if gated else None, _p(out[m0:m1])))
            finally:
                if gated:                        # (a 4-byte read-back on every exit path: the running counter and its host
                    seen = int(be.low_count.item()) & 0xFFFFFFFF     #  mirror stay in step even when a panel failed)
                    nlow = (seen - be.low_seen) & 0xFFFFFFFF
                    be.low_seen = seen
        if nlow and not self.replica:           # (a replica leaves them to ShardedPredictor: the fitting rank recomputes them)
            ys2 = torch.as_tensor(ys ** 2, device=be.device)
            # the kernel counted v < thr before un-normalising; re-select on the packed value with a hair of slack so that a
            # boundary row is never counted and then missed (recomputing one row too many is harmless)
            low = torch.nonzero(out[:, self.P] <= thr * float(ys[0] ** 2) * (1.0 + 1e-12)).ravel()
            v64 = self.predict_var_dev(self._rows64(Xq, q, low), kss, floor, "float64", self._fp64_var_method())
            out[low, self.P:] = v64[:, None] * ys2[None, :]
        return out

    def gram_host(self, ls, sf2, diag_add):
        """Debug/inspection helper: the (N, N) Gram matrix as a host array (overwrites the factor)."""
        self.gram(ls, sf2, diag_add)
        return self.K[: self.N, : self.N].cpu().numpy()
