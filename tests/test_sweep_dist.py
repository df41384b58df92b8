"""CPU tests of the multi-GPU sweep path: unit sharding and the final gather, with
world_size-2 gloo processes (the compute function is a stand-in; on the GPU box the same
code runs one process per GPU over RCCL)."""
import os
import socket
import sys

import numpy as np
import pytest
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    from koopman_realizations_amd import sweep
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        lassos = [0.1 * (i + 1) for i in range(7)]            # ragged: 7 units over 2 ranks
        calls = []

        def fit_one(l):
            calls.append(l)
            return np.full((3, 3), l) + np.eye(3)
        Ks = sweep.lasso_sweep(fit_one, lassos, rank, world, dist, shape=(3, 3))
        Ko = sweep.lasso_sweep(fit_one, lassos, rank, world, dist)           # object gather path
        systems = list(range(5))

        def eval_fn(sysid):
            return {mt: (np.arange(1, d + 1) * (sysid + 1.0), np.arange(d)) for mt, d in sweep.MAX_DEGREE.items()}
        tab = sweep.rand_models_sweep(systems, rank, world, dist, eval_fn=eval_fn)
        # snapshot-sharded single fit: Grams of the local rows, one all-reduce, local solve
        rng = np.random.default_rng(5)
        P = rng.standard_normal((101, 6)); Y = rng.standard_normal((101, 6))
        lo, hi = sweep.shard_rows(101, rank, world)
        Ksh = sweep.fit_sharded(lambda: (P[lo:hi].T @ P[lo:hi], P[lo:hi].T @ Y[lo:hi]), np.linalg.solve, dist)
        assert np.abs(Ksh - np.linalg.lstsq(P, Y, rcond=None)[0]).max() < 1e-12
        q.put((rank, len(calls) // 2, [k.tolist() for k in Ks], [k.tolist() for k in Ko], {k: v.tolist() for k, v in tab.items()}))
    finally:
        dist.destroy_process_group()


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
