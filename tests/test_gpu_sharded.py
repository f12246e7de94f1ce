"""RCCL tests of the query-sharded predict path on real GPUs: `ShardedPredictor` (libgpk launches followed by
`all_gather_into_tensor` on the same torch stream) under the `nccl` backend - one rank on the one-GPU box, two ranks
where two GPUs are visible - against the unsharded prediction of the same model.  The partition / padding logic
itself is covered on CPU by tests/test_sharded_gloo.py."""
import os
import socket
import sys

import numpy as np
import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, M, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    os.environ.setdefault("GPK_DEBUG_FILL", "nan")
    import torch
    import torch.distributed as dist
    torch.cuda.set_device(rank)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", rank))
    try:
        from unmanned_aerial_vehicles_amd import RBF, GaussianProcessRegressor, ShardedPredictor, WhiteKernel
        rng = np.random.default_rng(0)
        X = rng.standard_normal((1500, 9))
        Y = np.sin(X @ rng.standard_normal((9, 3))) + 0.1 * rng.standard_normal((1500, 3))
        Xq = np.random.default_rng(1).standard_normal((M, 9))
        g = GaussianProcessRegressor(kernel=RBF(2.0) + WhiteKernel(0.1), alpha=1e-4, normalize_y=True, optimizer=None,
                                     device=rank).fit(X, Y)
        ref_mean, ref_std = g.predict(Xq, return_std=True)                # unsharded, fp64
        ok = True
        for dtype, tm, ts in (("float64", 1e-11, 1e-9), ("float32", 1e-4, 1e-3)):
            sp = ShardedPredictor(g, dtype=dtype)
            for _ in range(3):                                            # back-to-back steps reuse the handle's scratch
                mean, var = sp.predict_mean_var(Xq)
            mean2 = sp.predict_mean(Xq)
            torch.cuda.synchronize()
            mean, var, mean2 = mean.cpu().numpy(), var.cpu().numpy(), mean2.double().cpu().numpy()
            e_m = np.max(np.abs(mean - ref_mean)) / np.max(np.abs(ref_mean))
            e_s = np.max(np.abs(np.sqrt(var) - ref_std) / ref_std)
            e_m2 = np.max(np.abs(mean2 - ref_mean)) / np.max(np.abs(ref_mean))
            ok = ok and mean.shape == (M, 3) and e_m < tm and e_s < ts and e_m2 < tm
        # a low-noise model (noise = 1e-3) whose queries include training points: the fp32 request goes through the serving
        # gates of DeviceGP on every rank (fp64 kernels for the mean if its fp32 error would leave 1e-4; fp64 recompute of
        # the variances that are a small fraction of the prior's) and must meet the fp32 bars; the raw fp32 kernels
        # (gated=False) must NOT be what served it
        g2 = GaussianProcessRegressor(kernel=RBF(2.0) + WhiteKernel(1e-3), alpha=1e-8, normalize_y=True, optimizer=None,
                                      device=rank).fit(X, Y)
        Xq2 = np.vstack([X[:300], Xq[: max(M - 300, 1)]])
        r_mean, r_std = g2.predict(Xq2, return_std=True)
        mean, var = ShardedPredictor(g2, dtype="float32").predict_mean_var(Xq2)
        mean, var = mean.cpu().numpy(), var.cpu().numpy()
        e_m = np.max(np.abs(mean - r_mean)) / np.max(np.abs(r_mean))
        e_s = np.max(np.abs(np.sqrt(var) - r_std) / r_std)
        _, var_raw = ShardedPredictor(g2, dtype="float32", gated=False).predict_mean_var(Xq2)
        e_raw = np.max(np.abs(np.sqrt(var_raw.cpu().numpy()) - r_std) / r_std)
        ok = ok and e_m < 1e-4 and e_s < 1e-3 and e_raw > e_s
        # replication by broadcast: only rank 0's estimator is used, the other ranks receive X, alpha and
        # the split inverse factor and never hold L; the low variances of the low-noise model are recomputed on rank 0
        for gm, Xt, rm, rs in ((g, Xq, ref_mean, ref_std), (g2, Xq2, r_mean, r_std)):
            sp = ShardedPredictor(gm if rank == 0 else None, dtype="float32").replicate(0)
            if rank != 0:
                ok = ok and sp.gpr._dev.replica and sp.gpr._dev.K is None and "f64" not in sp.gpr._dev._Winv
            mean, var = sp.predict_mean_var(Xt)
            mean, var = mean.cpu().numpy(), var.cpu().numpy()
            ok = ok and np.max(np.abs(mean - rm)) / np.max(np.abs(rm)) < 1e-4 and np.max(np.abs(np.sqrt(var) - rs) / rs) < 1e-3
        # a replica built in this process serves bit-identically to the model it was taken from, and owns no factor
        from unmanned_aerial_vehicles_amd.device import DeviceGP
        meta, tens = g._dev.serving_state()
        rep = DeviceGP.from_serving_state(meta, tens, g._dev.be)
        a = g._dev.predict_packed_dev(Xq, g._y_train_mean, g._y_train_std, 1.1, 0.0, "float32")
        b = rep.predict_packed_dev(Xq, g._y_train_mean, g._y_train_std, 1.1, 0.0, "float32")
        ok = ok and torch.equal(a, b) and rep.fp32_mean_ok() == g._dev.fp32_mean_ok()
        try:
            rep.predict_var_dev(Xq[:4], 1.1, 0.0, "float64", rep._fp64_var_method())
            ok = False
        except RuntimeError:
            pass
        one = torch.ones(1, device=torch.device("cuda", rank))
        dist.all_reduce(one)
        q.put((rank, bool(ok), int(one.item())))
    finally:
        dist.destroy_process_group()


def _run(world, M):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, M, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=600) for _ in range(world)]
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    assert sorted(r[0] for r in res) == list(range(world))
    assert all(r[1] for r in res), "sharded predictions differ from the unsharded model"
    assert all(r[2] == world for r in res)


def test_sharded_predictor_rccl_one_rank():
    _run(1, 777)


def test_sharded_predictor_rccl_two_ranks():
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs")
    _run(2, 1001)
