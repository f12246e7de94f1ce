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
