"""World-size-2 (and 3) gloo test of the query-sharded predict path on CPU: the partition, the padded
all-gather and the reassembly are exercised with the oracle standing in for the per-rank kernel call."""
import os
import socket
import sys

import numpy as np
import pytest

from conftest import ROOT


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
    import torch
    import torch.distributed as dist
    from oracle import gp_oracle as O
    from unmanned_aerial_vehicles_amd.sharded import shard_bounds, sharded_predict
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        X, Y, Xq = O.synthetic_problem(200, M)
        st = O.fit_fixed(X, Y, 2.0, 1.0, 0.1, 1e-4)
        calls = []

        def local(qs):
            calls.append(len(qs))
            return torch.from_numpy(O.predict(st, qs) if len(qs) else np.zeros((0, 3)))

        out = sharded_predict(local, Xq, None).numpy()
        ref = O.predict(st, Xq)
        m0, m1, _ = shard_bounds(M, world, rank)
        q.put((rank, bool(np.allclose(out, ref, rtol=1e-13, atol=1e-15)), calls == [m1 - m0]))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,M", [(2, 101), (2, 64), (3, 10), (2, 1)])
def test_sharded_predict_gloo(world, M):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, M, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in range(world)]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert sorted(r[0] for r in res) == list(range(world))
    assert all(r[1] for r in res), "gathered predictions differ from the unsharded oracle"
    assert all(r[2] for r in res), "each rank must compute exactly its own shard"


def _replicate_worker(rank, world, port, q):
    """broadcast_state + patch_low_rows under gloo with CPU tensors standing in for the device buffers."""
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch
    import torch.distributed as dist
    from unmanned_aerial_vehicles_amd import sharded
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        sharded.BROADCAST_CHUNK_BYTES = 4096                       # several pieces per tensor
        names = ("X", "alpha", "W2", "w_scales")
        g = torch.Generator().manual_seed(5)
        full = {"X": torch.randn(300, 9, generator=g, dtype=torch.float64), "alpha": torch.randn(300, 3, generator=g, dtype=torch.float64),
                "W2": torch.randint(0, 256, (384 * 384 * 4,), generator=g, dtype=torch.uint8),
                "w_scales": torch.rand(3, generator=g, dtype=torch.float32)}
        src = 1 if world > 1 else 0
        meta = tensors = None
        if rank == src:
            meta = {"N": 300, "D": 9, "P": 3, "Np": 384, "ls": [2.0] * 9, "sf2": 1.0,
                    "shapes": {k: tuple(v.shape) for k, v in full.items()},
                    "dtypes": {k: str(v.dtype).replace("torch.", "") for k, v in full.items()}}
            tensors = full
        meta, got = sharded.broadcast_state(meta, tensors, names, src, None, device="cpu")
        same = all(torch.equal(got[k], full[k]) for k in names) and meta["Np"] == 384 and meta["ls"] == [2.0] * 9
        # the cross-rank variance gate: rows below the threshold are recomputed on `src` only and handed out
        out = torch.arange(40, dtype=torch.float64).reshape(10, 4).clone()      # P = 2: columns 2.. are variances
        out[3, 2:] = 0.001
        out[7, 2:] = 0.002
        calls = []

        def recompute(rows):
            calls.append(rows.tolist())
            return torch.full((rows.numel(), 2), 5.0, dtype=torch.float64) + rows[:, None].double()

        patched, n = sharded.patch_low_rows(out.clone(), 2, 0.01, recompute, src, None)
        ok_patch = (n == 2 and patched[3, 2:].tolist() == [8.0, 8.0] and patched[7, 2:].tolist() == [12.0, 12.0]
                    and torch.equal(patched[:, :2], out[:, :2]) and calls == ([[3, 7]] if rank == src else []))
        clean, n0 = sharded.patch_low_rows(out.clone(), 2, 1e-6, recompute, src, None)       # nothing low: no collective
        q.put((rank, bool(same), bool(ok_patch and n0 == 0 and torch.equal(clean, out))))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_replicate_by_broadcast_gloo(world):
    """SURVEY.md 8(e) "broadcast once from rank 0": the replication protocol of ShardedPredictor.replicate."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_replicate_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in range(world)]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert all(r[1] for r in res), "a rank received a different serving state"
    assert all(r[2] for r in res), "the cross-rank variance gate disagrees"
