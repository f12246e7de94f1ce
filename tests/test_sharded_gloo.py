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


class _FakeBackend:
    device = "cpu"

    @staticmethod
    def upload(a):
        import torch
        return torch.from_numpy(np.ascontiguousarray(a))


class _FakeDev:
    """Stands in for DeviceGP on a CPU rank: the serving entry points ShardedPredictor calls, on torch CPU tensors.
    mean = 2 q[:, :P] (+ 1e-3 when served "in fp32"), variance = 1 (fp32) / 0.5 (fp64); the fitting rank holds the factor."""
    FP32_VAR_RECHECK_FRACTION = 1e-2

    def __init__(self, rank, src, verdicts):
        self.P, self.be, self.rank, self.replica = 2, _FakeBackend(), rank, rank != src
        self.verdicts = verdicts            # rank -> what this rank's own fp32_mean_ok says
        self.log = []

    def fp32_mean_ok(self, q=None):
        self.log.append(("gate", len(q)))
        return self.verdicts[self.rank]

    def _fp64_var_method(self):
        if self.replica:
            raise RuntimeError("a serving replica holds no factor")
        return "inverse"

    def predict_var_dev(self, q, kss, floor, dtype, method):
        import torch
        assert not self.replica and dtype == "float64"
        self.log.append(("var64", len(q)))
        return torch.full((len(q),), 0.5, dtype=torch.float64)

    def predict_gated_dev(self, q, ym, ys, kss, floor, dtype, method, gated, mean_gate=None):
        import torch
        q = torch.as_tensor(q)
        f32 = dtype == "float32" and not (gated and mean_gate is False)
        self.log.append(("mean", "f32" if f32 else "f64", len(q)))
        return (2.0 * q[:, : self.P] + (1e-3 if f32 else 0.0)).to(torch.float32 if f32 else torch.float64), None

    def predict_packed_dev(self, q, ym, ys, kss, floor, dtype, method, gated, mean_gate=None):
        import torch
        assert mean_gate is not None, "the sharded predictor must hand its collective decision down"
        if mean_gate is False and self.replica:
            raise RuntimeError("a replica was sent down the fp64 variance route")
        mean = self.predict_gated_dev(q, ym, ys, None, floor, dtype, method, gated, mean_gate)[0].double()
        var = torch.full((len(mean), self.P), 1.0 if mean_gate else 0.5, dtype=torch.float64)
        return torch.cat([mean, var], dim=1)


class _FakeGPR:
    def __init__(self, dev):
        self._dev, self._y_train_mean, self._y_train_std = dev, np.zeros(2), np.ones(2)

    def _ensure_device(self):
        pass


def _gate_worker(rank, world, port, verdicts, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch
    import torch.distributed as dist
    from unmanned_aerial_vehicles_amd.sharded import ShardedPredictor
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        Xq = torch.arange(11 * 3, dtype=torch.float64).reshape(11, 3)
        dev = _FakeDev(rank, 0, verdicts)
        sp = ShardedPredictor(_FakeGPR(dev), dtype="float32")
        sp._kss = lambda: 1.0
        sp.src = 0                                       # as after replicate(): rank 0 fitted, the others are replicas
        mean, var = sp.predict_mean_var(Xq)
        expect_f32 = all(verdicts)
        ok_mean = bool(torch.allclose(mean, 2.0 * Xq[:, :2] + (1e-3 if expect_f32 else 0.0), rtol=0, atol=1e-6))
        ok_var = bool(torch.equal(var, torch.full((11, 2), 1.0 if expect_f32 else 0.5, dtype=torch.float64)))
        gate_rows = [e[1] for e in dev.log if e[0] == "gate"]
        served = {e[1] for e in dev.log if e[0] == "mean"}
        # every rank looked at the WHOLE batch, and served its shard in the dtype all ranks agreed on
        q.put((rank, ok_mean and ok_var, gate_rows == [11], served == ({"f32"} if expect_f32 else {"f64"}),
               [e for e in dev.log if e[0] == "var64"]))
        m2 = sp.predict_mean(Xq)
        assert m2.shape == (11, 2)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("verdicts", [(True, True), (True, False), (False, True)])
def test_fp32_gate_is_decided_collectively(verdicts):
    """one rank whose view fails the batch-level fp32 mean gate must not leave the others waiting in the
    all-gather.  The gate is evaluated on the whole batch and combined by an all-reduce (MIN); when it fails after
    replicate(), replicas serve fp64 means only and the fitting rank computes every variance."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_gate_worker, args=(r, 2, port, verdicts, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in range(2)]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    for rank, ok_values, ok_gate, ok_dtype, var_calls in res:
        assert ok_values, f"rank {rank}: wrong predictions"
        assert ok_gate, f"rank {rank}: the gate must look at the whole batch once"
        assert ok_dtype, f"rank {rank}: served in a dtype the ranks did not agree on"
        if all(verdicts):
            assert var_calls == []
        else:
            assert var_calls == ([("var64", 11)] if rank == 0 else []), "only the fitting rank computes fp64 variances"
