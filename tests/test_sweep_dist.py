"""CPU tests of the multi-GPU sweep path: unit sharding, packing and the final gather - the same code that runs one
process per GPU over RCCL (comm.RcclComm through the C ABI) - here with world_size-2 gloo processes standing in for the
collectives (the compute function is a stand-in; the HIP path on one GPU is covered by test_gpu_sweep.py)."""
import os
import socket
import sys

import numpy as np
import pytest
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


class GlooComm:
    """The four members the sweep code needs from a communicator, over torch.distributed / gloo (test-only: the
    package itself is torch-free)."""

    def __init__(self, dist):
        self.dist, self.rank, self.world = dist, dist.get_rank(), dist.get_world_size()

    def all_gather_bytes(self, payload):
        import torch
        t = torch.frombuffer(bytearray(payload), dtype=torch.uint8)
        out = [torch.empty_like(t) for _ in range(self.world)]
        self.dist.all_gather(out, t)
        return [bytes(o.numpy().tobytes()) for o in out]

    def all_reduce_sum(self, a):
        import torch
        t = torch.from_numpy(np.ascontiguousarray(a, dtype=np.float64).copy())
        self.dist.all_reduce(t)
        return t.numpy()

    def barrier(self):
        self.dist.barrier()


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    from koopman_realizations_amd import sweep, comm as kcomm
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        comm = GlooComm(dist)
        lassos = [0.1 * (i + 1) for i in range(7)]            # ragged: 7 units over 2 ranks
        calls = []

        def fit_one(l):
            calls.append(l)
            return np.full((3, 3), l) + np.eye(3)
        Ks = sweep.lasso_sweep(fit_one, lassos, comm, shape=(3, 3))
        Ko = sweep.lasso_sweep(fit_one, lassos, comm)           # object gather path (ragged pickles)
        Km = sweep.lasso_sweep(None, lassos, comm, shape=(3, 3), fit_many=lambda ls: [fit_one(l) for l in ls])
        assert all((a == b).all() for a, b in zip(Ks, Km))
        # device-resident K stack (sweep.lasso_sweep_device): the rank's shard stays in the "device" result buffer (a stand-in
        # context here), ONE gather of per-rank stacks with a padding slot on the short rank, column-major blocks
        class FakeCtx:
            def __init__(self):
                self.stack = None

            def fit_results(self, first, count, W):
                return self.stack[first:first + count]
        fctx = FakeCtx()

        def fit_device(ls):
            # each block column-major, as the device result buffer holds it
            fctx.stack = np.ascontiguousarray(np.transpose(np.stack([np.full((3, 3), l) + np.arange(9.0).reshape(3, 3) for l in ls]), (0, 2, 1)))
        Kd = sweep.lasso_sweep_device(fctx, fit_device, lassos, 3, comm)
        assert len(Kd) == 7 and all(np.array_equal(k, np.full((3, 3), l) + np.arange(9.0).reshape(3, 3)) for k, l in zip(Kd, lassos))
        systems = list(range(5))

        def eval_fn(sysid):
            return {mt: (np.arange(1, d + 1) * (sysid + 1.0), np.arange(d)) for mt, d in sweep.MAX_DEGREE.items()}
        tab = sweep.rand_models_sweep(systems, comm, eval_fn=eval_fn)
        # snapshot-sharded single fit: Grams of the local rows, one all-reduce, local solve
        rng = np.random.default_rng(5)
        P = rng.standard_normal((101, 6)); Y = rng.standard_normal((101, 6))
        lo, hi = sweep.shard_rows(101, rank, world)
        Ksh = sweep.fit_sharded(lambda: (P[lo:hi].T @ P[lo:hi], P[lo:hi].T @ Y[lo:hi]), np.linalg.solve, comm)
        assert np.abs(Ksh - np.linalg.lstsq(P, Y, rcond=None)[0]).max() < 1e-12
        assert kcomm.max_over_ranks(comm, 1.0 + rank) == float(world)
        objs = kcomm.all_gather_object(comm, {"rank": rank, "pad": "x" * (10 + 1000 * rank)})
        assert [o["rank"] for o in objs] == list(range(world))
        q.put((rank, len(calls) // 3, [k.tolist() for k in Ks], [k.tolist() for k in Ko], {k: v.tolist() for k, v in tab.items()}))
    finally:
        dist.destroy_process_group()


def test_unique_id_file_rendezvous(tmp_path, monkeypatch):
    """Rank 0 publishes the 128-byte id atomically, the other ranks read it (the id itself comes from RCCL on a GPU box)."""
    from koopman_realizations_amd import comm as kcomm
    monkeypatch.setattr(kcomm, "unique_id", lambda: bytes(range(128)))
    path = str(tmp_path / "id")
    with pytest.raises(TimeoutError):
        kcomm.exchange_unique_id(1, path, timeout=0.1)
    assert kcomm.exchange_unique_id(0, path) == bytes(range(128))
    assert kcomm.exchange_unique_id(1, path, timeout=1.0) == bytes(range(128))
    monkeypatch.setenv("KP_COMM_FILE", path)
    assert kcomm.rendezvous_file_from_env() == path
    loc = kcomm.LocalComm()
    assert kcomm.all_gather_object(loc, {"a": 1}) == [{"a": 1}] and kcomm.all_gather_array(loc, np.arange(3.0)).shape == (1, 3)


def test_communicator_creation_runs_under_a_watchdog(monkeypatch):
    """A bootstrap that never completes raises RcclInitTimeout (init_from_env then votes for the file backend); errors of
    a call that does return pass through; single-node launches default the bootstrap to the loopback interface."""
    import time as _t
    import types
    from koopman_realizations_amd import comm as kcomm, _ffi as F

    class Ctx:
        handle = None

    slow = types.SimpleNamespace(kp_comm_create=lambda *a: _t.sleep(5.0) or 0)
    monkeypatch.setattr(F, "lib", lambda: slow)
    monkeypatch.delenv("NCCL_SOCKET_IFNAME", raising=False)
    monkeypatch.delenv("MASTER_ADDR", raising=False)
    t0 = _t.time()
    with pytest.raises(kcomm.RcclInitTimeout):
        kcomm.RcclComm(Ctx(), 0, 2, bytes(128), timeout=0.2)
    assert _t.time() - t0 < 2.0
    assert os.environ["NCCL_SOCKET_IFNAME"] == "lo"
    ok = types.SimpleNamespace(kp_comm_create=lambda *a: 0)
    monkeypatch.setattr(F, "lib", lambda: ok)
    monkeypatch.setattr(F, "check", lambda rc, h=None: None)
    assert kcomm.RcclComm(Ctx(), 1, 2, bytes(128), timeout=5.0).rank == 1
    monkeypatch.delenv("NCCL_SOCKET_IFNAME", raising=False)
    monkeypatch.setenv("MASTER_ADDR", "10.0.0.7")          # another node named: RCCL keeps its own interface search
    kcomm.single_node_defaults()
    assert "NCCL_SOCKET_IFNAME" not in os.environ


def test_package_is_torch_free():
    import re
    pk = os.path.join(ROOT, "koopman-realizations_amd")
    for fn in os.listdir(pk):
        if fn.endswith(".py"):
            src = open(os.path.join(pk, fn)).read()
            assert not re.search(r"^\s*(import torch|from torch)", src, re.M), fn


def test_sharding_is_a_partition():
    from koopman_realizations_amd import sweep
    for n, w in [(101, 2), (100000, 8), (3, 8)]:
        edges = [sweep.shard_rows(n, r, w) for r in range(w)]
        assert edges[0][0] == 0 and edges[-1][1] == n and all(a[1] == b[0] for a, b in zip(edges[:-1], edges[1:]))
    for n, w in [(0, 2), (1, 8), (7, 2), (64, 8), (1024, 8), (5, 8)]:
        ids = sorted(i for r in range(w) for i in sweep.shard_units(n, r, w))
        assert ids == list(range(n))
        sizes = [len(sweep.shard_units(n, r, w)) for r in range(w)]
        assert max(sizes) - min(sizes) <= 1


def test_two_rank_sweeps_gather_identically():
    from koopman_realizations_amd import sweep
    ctxm = mp.get_context("spawn")
    q = ctxm.Queue()
    port = _free_port()
    procs = [ctxm.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(2)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    res.sort()
    assert [r[1] for r in res] == [4, 3]                       # 7 units: 4 + 3
    lassos = [0.1 * (i + 1) for i in range(7)]
    want = [(np.full((3, 3), l) + np.eye(3)).tolist() for l in lassos]
    for r in res:
        assert r[2] == want and r[3] == want                    # every rank holds every K, in unit order
    assert res[0][4] == res[1][4]
    lin = np.array(res[0][4]["linear"])
    assert lin.shape == (13, 5) and np.allclose(lin[:, 3], np.arange(1, 14) * 4.0)
    mean, std = sweep.sweep_statistics(np.array([[1.0, 2.0, np.nan, 50.0], [1.0, 1.0, 1.0, 1.0]]))
    assert np.allclose(mean, [1.5, 1.0])
    # prctile's rule: [1 2 3 4] -> 0/25/50/75/100 % = 1, 1.5, 2.5, 3.5, 4 (a 99 system is dropped first)
    bars = sweep.sweep_percentiles(np.array([[1.0, 2.0, 3.0, 4.0, 99.0]]))
    assert np.allclose(bars, [[1.0, 1.5, 2.5, 3.5, 4.0]])
