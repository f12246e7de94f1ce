"""Multi-GPU batched prediction: queries are independent, so they shard across ranks.

One process per GPU (`torch.distributed`, backend "nccl" = RCCL over xGMI on ROCm).  The model
(X, alpha, L) is replicated — every rank fits or loads it redundantly, which is deterministic and
needs no communication.  Rank r predicts the contiguous slice
`[r * ceil(M/W), min(M, (r+1) * ceil(M/W)))` of the query batch; ONE all-gather of the
`(ceil(M/W), P)` result shards leaves the full `(M, P)` prediction on every rank.  No other
collective is on the data path.  (The path has no reference counterpart: the reference is a single
CPU process)
"""
from __future__ import annotations

import numpy as np


def shard_bounds(M, world_size, rank):
    per = -(-int(M) // int(world_size))
    m0 = min(int(M), rank * per)
    return m0, min(int(M), m0 + per), per


def all_gather_rows(local, M, group=None):
    """local: (rows_r, P) tensor holding this rank's shard -> (M, P) tensor on every rank."""
    import torch
    import torch.distributed as dist

    world = dist.get_world_size(group)
    per = -(-int(M) // world)
    P = local.shape[1]
    send = local
    if local.shape[0] != per:
        send = torch.zeros((per, P), dtype=local.dtype, device=local.device)
        send[: local.shape[0]] = local
    out = torch.empty((world * per, P), dtype=local.dtype, device=local.device)
    dist.all_gather_into_tensor(out, send.contiguous(), group=group)
    return out[:M]


def sharded_predict(predict_local, Xq, group=None):
    """Shard `Xq` (M, D) over the process group, run `predict_local(shard) -> (rows, P) tensor` on each
    rank, all-gather.  `predict_local` is the device model's K4 (and K5) call on a GPU; tests pass a
    CPU function to exercise the partition and the collective under gloo."""
    import torch.distributed as dist

    if not (dist.is_available() and dist.is_initialized()):
        return predict_local(Xq)
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    M = Xq.shape[0]
    m0, m1, _ = shard_bounds(M, world, rank)
    local = predict_local(Xq[m0:m1])
    return all_gather_rows(local, M, group)


def gram_slab_bounds(N, world_size, rank):
    """Rows [row0, row0 + nrows) of the padded (Np x Np) Gram matrix owned by `rank`: whole 128-row tiles, dealt in
    contiguous runs of ceil(tiles / world_size)."""
    ntiles = (int(N) + 127) // 128
    per = -(-ntiles // int(world_size))
    t0 = min(ntiles, rank * per)
    t1 = min(ntiles, t0 + per)
    return 128 * t0, 128 * (t1 - t0)


def sharded_gram(X, ls, sf2, diag_add, world_size=None, rank=None, backend=None, dtype="float64", out=None):
    """This rank's row slab of K = sf2 exp(-d^2/2) + diag_add I (the Gram build shards by row blocks
    with no exchange).  X: (N, D) array or device tensor, identical on every rank.  Returns (slab tensor
    (nrows, Np), row0); an empty slab (ranks beyond the tile count) has zero rows.  `out`: a tensor to reuse."""
    import ctypes as C

    import torch
    import torch.distributed as dist

    from . import _lib
    from .device import get_backend
    if world_size is None:
        on = dist.is_available() and dist.is_initialized()
        world_size, rank = (dist.get_world_size(), dist.get_rank()) if on else (1, 0)
    be = backend or get_backend()
    f64 = dtype in ("float64", np.float64, torch.float64)
    tdt = torch.float64 if f64 else torch.float32
    Xd = X.to(device=be.device, dtype=tdt).contiguous() if isinstance(X, torch.Tensor) else be.upload(np.asarray(X, dtype=np.float64), tdt)
    N, D = Xd.shape
    Np = (N + 127) // 128 * 128
    row0, nrows = gram_slab_bounds(N, world_size, rank)
    slab = out if out is not None else be.empty((nrows, Np), tdt)
    lsv = np.ascontiguousarray(np.broadcast_to(np.asarray(ls, dtype=np.float64), (D,)))
    if world_size == 1:
        # one rank owns every row: the fused kernel (each symmetric tile computed once and written to both halves)
        with be.lock:
            be.bind_stream()
            be.check(be.lib.gpk_gram(be.h, _lib.GPK_F64 if f64 else _lib.GPK_F32, C.c_void_p(Xd.data_ptr()), N, D,
                                     lsv.ctypes.data_as(_lib._dp), float(sf2), float(diag_add),
                                     C.c_void_p(slab.data_ptr()), Np))
    elif nrows:
        with be.lock:
            be.bind_stream()
            be.check(be.lib.gpk_gram_rows(be.h, _lib.GPK_F64 if f64 else _lib.GPK_F32, C.c_void_p(Xd.data_ptr()), N, D,
                                          lsv.ctypes.data_as(_lib._dp), float(sf2), float(diag_add), row0, nrows,
                                          C.c_void_p(slab.data_ptr()), Np))
    return slab, row0


BROADCAST_CHUNK_BYTES = 1 << 30      # large tensors go out in pieces of this size (one collective each)


def broadcast_state(meta, tensors, names, src=0, group=None, device=None, make_empty=None):
    """Replicate a model's serving state from rank `src` to every rank of the group: ONE object broadcast of the metadata
    (shapes, dtypes, hyper-parameters: plain Python) followed by one tensor broadcast per 1 GiB piece of each tensor in
    `names` (RCCL over xGMI on GPUs: 17 GB of split inverse factor at N = 65 536 takes ~0.15 s; gloo on CPU in the tests).
    Ranks other than `src` pass meta = tensors = None and receive freshly allocated tensors (`make_empty(shape, dtype)`,
    default torch.empty on `device`).  Returns (meta, tensors) on every rank."""
    import torch
    import torch.distributed as dist

    rank = dist.get_rank(group)
    box = [meta if rank == src else None]
    dist.broadcast_object_list(box, src=src, group=group)
    meta = box[0]
    out = {}
    for name in names:
        shape, dt = tuple(meta["shapes"][name]), getattr(torch, meta["dtypes"][name])
        if rank == src:
            t = tensors[name]
            assert tuple(t.shape) == shape and t.dtype == dt and t.is_contiguous(), name
        else:
            t = make_empty(shape, dt) if make_empty is not None else torch.empty(shape, dtype=dt, device=device)
        flat = t.view(-1)
        step = max(1, BROADCAST_CHUNK_BYTES // t.element_size())
        for i0 in range(0, flat.numel(), step):
            dist.broadcast(flat[i0:i0 + step], src=src, group=group)
        out[name] = t
    return meta, out


def patch_low_rows(out, P, threshold, recompute, src=0, group=None):
    """The fp32 variance gate across ranks: rows of the gathered (M, 2P) [mean | var] tensor whose variance is below
    `threshold` (identical on every rank after the all-gather, so every rank finds the same rows) are recomputed in fp64 by
    rank `src` - the one that holds the factor; `recompute(rows) -> (n, P) variances` is called there only - and ONE
    broadcast hands them out.  No collective at all while no row is low (the common case)."""
    import torch
    import torch.distributed as dist

    low = torch.nonzero(out[:, P] < threshold).ravel()
    if low.numel() == 0:
        return out, 0
    fix = torch.empty((low.numel(), P), dtype=out.dtype, device=out.device)
    if dist.get_rank(group) == src:
        fix.copy_(recompute(low))
    dist.broadcast(fix, src=src, group=group)
    out[low, P:] = fix
    return out, int(low.numel())


class _ReplicaModel:
    """What a rank that received a model by broadcast holds in place of the fitted estimator."""

    def __init__(self, dev, y_mean, y_std, sf2, noise):
        self._dev, self._y_train_mean, self._y_train_std = dev, y_mean, y_std
        self.sf2, self.noise = sf2, noise

    def _ensure_device(self):
        pass


class ShardedPredictor:
    """Query-sharded posterior mean (+ variance) for a fitted `GaussianProcessRegressor`.  dtype "float32" (the default)
    is served through the gated predictors of `DeviceGP` (`predict_gated_dev` / `predict_packed_dev`): a model whose fp32
    mean would leave the stated 1e-4 runs on the fp64 kernels on every rank (`mean_gate`: decided once on the whole batch
    and combined by an all-reduce, so all ranks decide alike), and low fp32 variances are recomputed in fp64 inside the
    owning rank (after `replicate()`: by the fitting rank)."""

    def __init__(self, gpr, group=None, dtype="float32", gated=True):
        self.gpr, self.group, self.dtype, self.gated = gpr, group, dtype, gated
        self.src = None                      # set by replicate(): the rank that fitted (and holds the factor)

    def replicate(self, src=0):
        """Replicate the model by broadcast instead of by redundant refit: rank `src` holds the fitted
        estimator, every other rank passes gpr=None and receives X, alpha and the split inverse factor + scales (what fp32
        serving needs; L and the fp64 inverse factor stay on `src` only).  fp32 variances below the re-check threshold
        are recomputed on `src` (`patch_low_rows`).  Returns self."""
        import torch.distributed as dist
        from .device import DeviceGP, get_backend
        rank = dist.get_rank(self.group)
        be = get_backend()
        meta = tensors = None
        if rank == src:
            g = self.gpr
            g._ensure_device()
            comp = g.kernel_.components()
            meta, tensors = g._dev.serving_state()
            meta = dict(meta, y_mean=np.asarray(g._y_train_mean).tolist(), y_std=np.asarray(g._y_train_std).tolist(),
                        kernel_sf2=float(comp.sf2), kernel_noise=float(comp.noise or 0.0))
        meta, tensors = broadcast_state(meta, tensors, DeviceGP.SERVING_TENSORS, src, self.group,
                                        make_empty=lambda shape, dt: be.empty(shape, dt))
        if rank != src:
            dev = DeviceGP.from_serving_state(meta, tensors, be)
            self.gpr = _ReplicaModel(dev, np.asarray(meta["y_mean"]), np.asarray(meta["y_std"]), meta["kernel_sf2"],
                                     meta["kernel_noise"])
        self.src = src
        return self

    def _kss(self):
        g = self.gpr
        if isinstance(g, _ReplicaModel):
            return g.sf2 + g.noise
        comp = g.kernel_.components()
        return comp.sf2 + (comp.noise or 0.0)

    def mean_gate(self, Xq):
        """The fp32 mean gate for the WHOLE batch, one answer for every rank.  The batch-level check of `fp32_mean_ok` looks at
        up to 1024 evenly spaced rows of what it is given: given a rank's own shard the ranks can disagree, and a rank that
        alone switches to the fp64 kernels - a replica cannot even serve fp64 variances - leaves the others waiting in the
        all-gather.  So every rank evaluates the gate on the full `Xq` (identical everywhere: same rows, same replicated
        model) and the answers are combined by ONE all-reduce (MIN) for good measure.  None: nothing to decide."""
        import torch
        import torch.distributed as dist
        if not (self.gated and self.dtype == "float32"):
            return None
        g = self.gpr
        g._ensure_device()
        ok = bool(g._dev.fp32_mean_ok(Xq))
        if dist.is_available() and dist.is_initialized() and dist.get_world_size(self.group) > 1:
            # (gloo reduces host tensors, RCCL device tensors)
            on_gpu = dist.get_backend(self.group) == "nccl"
            flag = torch.tensor([1 if ok else 0], dtype=torch.int32, device=g._dev.be.device if on_gpu else "cpu")
            dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=self.group)
            ok = bool(int(flag.item()))
        return ok

    def predict_mean(self, Xq):
        """Xq: (M, D) tensor or array, identical on every rank -> (M, P) device tensor."""
        g = self.gpr
        g._ensure_device()
        gate = self.mean_gate(Xq)
        return sharded_predict(
            lambda q: g._dev.predict_gated_dev(q, g._y_train_mean, g._y_train_std, None, 0.0, self.dtype, "auto",
                                               self.gated, gate)[0], Xq, self.group)

    def predict_mean_var(self, Xq):
        import torch
        import torch.distributed as dist
        g = self.gpr
        g._ensure_device()
        kss = self._kss()
        P = g._dev.P
        gate = self.mean_gate(Xq)
        ys = np.broadcast_to(np.asarray(g._y_train_std, dtype=np.float64), (P,))

        def var64(rows):                      # fp64 variances of the caller's rows `rows` (index tensor, or None: all), un-normalised
            if rows is None:
                q = Xq if isinstance(Xq, torch.Tensor) else g._dev.be.upload(np.ascontiguousarray(Xq, dtype=np.float64))
            else:
                q = Xq[rows] if isinstance(Xq, torch.Tensor) else g._dev.be.upload(np.ascontiguousarray(Xq, dtype=np.float64)[rows.cpu().numpy()])
            v = g._dev.predict_var_dev(q.to(g._dev.be.device).double().contiguous(), kss, 0.0, "float64", g._dev._fp64_var_method())
            return v[:, None] * torch.as_tensor(ys ** 2, device=v.device)[None, :]

        if self.src is not None and gate is False:
            # The batch failed the gate and the model was replicated by broadcast: the replicas hold no factor, so they serve
            # the fp64 MEANS of their shards (X and alpha are fp64 everywhere) and the fitting rank computes every variance
            # and hands them out with one broadcast - the same two collectives on every rank, nobody is left waiting.
            mean = sharded_predict(
                lambda q: g._dev.predict_gated_dev(q, g._y_train_mean, g._y_train_std, None, 0.0, "float64", "auto", False)[0],
                Xq, self.group)
            var = torch.empty((mean.shape[0], P), dtype=torch.float64, device=mean.device)
            if dist.get_rank(self.group) == self.src:
                var.copy_(var64(None))
            dist.broadcast(var, src=self.src, group=self.group)
            return mean, var
        out = sharded_predict(
            lambda q: g._dev.predict_packed_dev(q, g._y_train_mean, g._y_train_std, kss, 0.0, self.dtype, "auto", self.gated, gate),
            Xq, self.group)
        if self.src is not None and self.gated and self.dtype == "float32":
            # replicas cannot recompute low fp32 variances (no factor): the fitting rank does it for everybody
            thr = g._dev.FP32_VAR_RECHECK_FRACTION * kss * float(ys[0] ** 2)
            out, _ = patch_low_rows(out, P, thr, var64, self.src, self.group)
        return out[:, :P], out[:, P:]
