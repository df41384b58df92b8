"""GPU tests of the one-caller multi-GPU entry points (kp_multi_*, koopman_realizations_amd.multi) through the C ABI.

One GPU per box here, so the device is listed two or three times: two / three contexts, two / three worker threads - the
fan-out, the ragged shards, the scatter of the results into the caller's stack (pageable and page-locked destinations) and the
exchange of the snapshot-sharded fit are the code that runs with distinct devices.  Oracle: the single-context path of the
same library (itself checked against oracle/ in test_gpu_fit.py, test_gpu_lasso.py, test_gpu_sweep.py)."""
import numpy as np
import pytest

import koopman_realizations_amd as kra
from koopman_realizations_amd.multi import Multi, MultiMpc
from conftest import synth_pairs

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def cfg3(ctx):
    """BASELINE configs[3] at its shape: bilinear poly-3 (W = 336), 1e5 pairs, lasso values t/N log-spaced in [1e-2, 1e2]."""
    p = synth_pairs(100000, seed=0)
    exps = kra.poly_exponent_table(6, 3)[6:]
    b = kra.Basis(ctx, "bilinear", 6, 3, [("poly", exps)])
    s = kra.Snapshots(ctx, p["alpha"], p["beta"], p["u"])
    return {"p": p, "dic": ("bilinear", 6, 3, [("poly", exps)], None), "b": b, "s": s}


@pytest.mark.parametrize("ids", [[0, 0], [0, 0, 0]])
def test_lasso_grid_dealt_over_workers_equals_the_single_context_grid(ctx, cfg3, ids):
    """kp_multi_fit: 13 values (ragged over 2 and 3 workers) of the configs[3] grid, every worker lifting the snapshots once
    and solving its values as one batch; value i lands at position i of the caller's stack - bit for bit what ONE kp_fit
    call over the same values of a worker's shard gives."""
    vals = np.geomspace(1e-2, 1e2, 13)
    p = cfg3["p"]
    n = len(ids)
    ref = {}
    for r in range(n):                                       # the shard of worker r as ONE kp_fit call of the plain context
        mine = list(range(r, 13, n))
        for i, K in zip(mine, kra.fit(ctx, cfg3["b"], cfg3["s"], vals[mine])):
            ref[i] = K
    mg = Multi(ids)
    try:
        Ks = mg.fit(cfg3["dic"], p["alpha"], p["beta"], p["u"], vals)                       # pageable destination: staged
        assert Ks.shape == (13, 336, 336)
        for i in range(13):
            assert np.array_equal(Ks[i].T, ref[i]), i
        out = mg.host_array("K", (13, 336, 336))                                            # page-locked for every device: direct DMA
        out[...] = np.nan
        Kp = mg.fit(cfg3["dic"], p["alpha"], p["beta"], p["u"], vals, out=out)
        assert Kp is out and np.array_equal(Kp, Ks)
        tm = mg.timers()
        assert tm.shape == (n, 4) and (tm[:, 1] > 0).all() and (tm[:, 3] >= tm[:, 1]).all()
        # a second dictionary on the same object: the workers replace their resident dictionary
        e2 = kra.poly_exponent_table(6, 2)[6:]
        b2 = kra.Basis(ctx, "bilinear", 6, 3, [("poly", e2)])
        K2 = mg.fit(("bilinear", 6, 3, [("poly", e2)], None), p["alpha"], p["beta"], p["u"], [np.inf])
        assert np.array_equal(K2[0].T, kra.fit(ctx, b2, cfg3["s"])[0])
        b2.close()
        # fewer values than workers: the idle workers take no job
        K1 = mg.fit(cfg3["dic"], p["alpha"], p["beta"], p["u"], [np.inf])
        assert np.array_equal(K1[0].T, kra.fit(ctx, cfg3["b"], cfg3["s"])[0])
    finally:
        mg.close()


@pytest.mark.parametrize("ids", [[0], [0, 0], [0, 0, 0]])
def test_one_fit_sharded_over_snapshots_sums_the_grams_on_worker_zero(ctx, cfg3, ids):
    """kp_multi_fit_sharded: rows dealt in contiguous ranges (100 000 over 3 is ragged), each worker's fused Gram kernel on its
    rows, [G | C] by peer copy to worker 0, summed there in worker order, one solve.  Against the whole-matrix fit: equal to
    rounding (another summation order of the partial Grams), and reproducible run to run bit for bit."""
    p = cfg3["p"]
    Kref = kra.fit(ctx, cfg3["b"], cfg3["s"])[0]
    mg = Multi(ids)
    try:
        K1 = mg.fit_sharded(cfg3["dic"], p["alpha"], p["beta"], p["u"])[0].T.copy()
        K2 = mg.fit_sharded(cfg3["dic"], p["alpha"], p["beta"], p["u"])[0].T.copy()
        assert np.array_equal(K1, K2)
        assert np.abs(K1 - Kref).max() <= 1e-11 * np.abs(Kref).max()
        if len(ids) == 1:
            assert np.array_equal(K1, Kref)
        # property that does not need the reference fit: the sum of the shards' Grams is the Gram of the whole (linearity)
        lo = 0
        Gsum = 0.0
        for r in range(len(ids)):
            hi = 100000 * (r + 1) // len(ids)
            sr = kra.Snapshots(ctx, p["alpha"][lo:hi], p["beta"][lo:hi], p["u"][lo:hi])
            Gsum = Gsum + kra.fit_gram(ctx, cfg3["b"], sr)[0]
            sr.close()
            lo = hi
        G = kra.fit_gram(ctx, cfg3["b"], cfg3["s"])[0]
        assert np.abs(Gsum - G).max() <= 1e-12 * np.abs(G).max()
    finally:
        mg.close()


def test_sweep_and_batched_mpc_dealt_in_contiguous_chunks(ctx):
    from koopman_realizations_amd.device import Traj, Mpc
    from test_mex_gateway import _stacks
    Y, U, Yv, Uv, k = _stacks(nb=29, seed=3)                       # rows x 1 x nb stacks -> (nb, rows, 1)
    Yn, Un, Yvn, Uvn = (np.ascontiguousarray(np.transpose(x, (2, 0, 1))) for x in (Y, U, Yv, Uv))
    tp = Traj(ctx, Yn, Un, k, Yvn, Uvn)
    mg = Multi([0, 0, 0])
    try:
        mt_ = mg.traj_upload(Yn, Un, k, Yvn, Uvn)
        for mt, D, las in (("linear", 13, np.inf), ("bilinear", 6, np.inf), ("nonlinear", 4, 4.0)):
            nv = 1 + (mt == "nonlinear")
            e = kra.poly_exponent_table(nv, D)[nv:]
            err, st = mt_.sweep_eval_nested((mt, 1, 1, [("poly", e)], None), D, las)
            b = kra.Basis(ctx, mt, 1, 1, [("poly", e)])
            err1, st1 = tp.sweep_eval_nested(b, D, las)
            assert np.array_equal(err, err1, equal_nan=True) and np.array_equal(st, st1)
            b.close()
        mt_.close()
        # MPC
        p = synth_pairs(4000, 3, 2, seed=21)
        e = kra.poly_exponent_table(3, 2)[3:]
        bp = kra.Basis(ctx, "bilinear", 3, 2, [("poly", e)])
        K = kra.fit(ctx, bp, kra.Snapshots(ctx, p["alpha"], p["beta"], p["u"]))[0]
        N = bp.N
        A = np.asfortranarray(K.T[:N, :N]); B = np.asfortranarray(K.T[:N, N:])
        proj = np.hstack([np.eye(2), np.zeros((2, N - 2))])
        args = ("bilinear", A, B, 8, proj, 10.0, 100.0, np.array([3e-3, 2e-3]), np.array([-0.9, -0.9]), np.array([0.9, 0.9]), 0.2, None)
        mm = MultiMpc(mg, *args)
        m1 = Mpc(ctx, *args)
        rng = np.random.default_rng(1)
        nb = 50
        Z = bp.lift(1, rng.uniform(-0.5, 0.5, (nb, 3)))
        UP = rng.uniform(-0.3, 0.3, (nb, 2)); YR = rng.uniform(-0.5, 0.5, (nb, 18))
        Um, sm = mm.step_batch(Z, UP, YR)
        U1, s1 = m1.step_batch(Z, UP, YR)
        assert np.array_equal(Um, U1) and np.array_equal(sm, s1) and (sm == 0).all()
        mm.close(); m1.close()
    finally:
        mg.close()
        tp.close()


def test_eight_way_shards_of_the_grid_and_the_sweep_equal_the_single_context(ctx, cfg3):
    """The shapes an 8-GPU node runs - BASELINE configs[3] (64 lasso values on the bilinear poly-3 fit) and configs[4] (a batch of
    random systems x {linear, bilinear, nonlinear}) - through the one-caller block with EIGHT workers (the box's one device listed
    eight times: eight contexts, eight worker threads, eight shards; 61 values and 203 systems are ragged over 8).  Every value
    / system lands at its position of the caller's stack bit for bit as the single-context path computes it."""
    from koopman_realizations_amd.device import Traj
    from test_mex_gateway import _stacks
    ids = [0] * 8
    p = cfg3["p"]
    mg = Multi(ids)
    try:
        for nv in (64, 61):
            vals = np.geomspace(1e-2, 1e2, nv)
            ref = {}
            for r in range(8):                                   # worker r's shard (round robin) as ONE kp_fit call of the plain context
                mine = list(range(r, nv, 8))
                for i, K in zip(mine, kra.fit(ctx, cfg3["b"], cfg3["s"], vals[mine])):
                    ref[i] = K
            Ks = mg.fit(cfg3["dic"], p["alpha"], p["beta"], p["u"], vals)
            assert Ks.shape == (nv, 336, 336)
            for i in range(nv):
                assert np.array_equal(Ks[i].T, ref[i]), (nv, i)
        tm = mg.timers()
        assert tm.shape == (8, 4) and (tm[:, 1] > 0).all()
        # configs[4]: 203 systems in contiguous chunks of 26 / 25 over the eight workers
        Y, U, Yv, Uv, k = _stacks(nb=203, seed=5)
        Yn, Un, Yvn, Uvn = (np.ascontiguousarray(np.transpose(x, (2, 0, 1))) for x in (Y, U, Yv, Uv))
        tp = Traj(ctx, Yn, Un, k, Yvn, Uvn)
        mt_ = mg.traj_upload(Yn, Un, k, Yvn, Uvn)
        for mt, D, las in (("linear", 13, np.inf), ("bilinear", 6, np.inf), ("nonlinear", 4, 4.0)):
            nvar = 1 + (mt == "nonlinear")
            e = kra.poly_exponent_table(nvar, D)[nvar:]
            err, st = mt_.sweep_eval_nested((mt, 1, 1, [("poly", e)], None), D, las)
            b = kra.Basis(ctx, mt, 1, 1, [("poly", e)])
            err1, st1 = tp.sweep_eval_nested(b, D, las)
            assert 203 in err.shape and np.array_equal(err, err1, equal_nan=True) and np.array_equal(st, st1), mt
            b.close()
        mt_.close()
        tp.close()
        # one fit sharded eight ways over its snapshots: rounding-level agreement with the whole-matrix fit, reproducible
        Kref = kra.fit(ctx, cfg3["b"], cfg3["s"])[0]
        K1 = mg.fit_sharded(cfg3["dic"], p["alpha"], p["beta"], p["u"])[0].T.copy()
        K2 = mg.fit_sharded(cfg3["dic"], p["alpha"], p["beta"], p["u"])[0].T.copy()
        assert np.array_equal(K1, K2) and np.abs(K1 - Kref).max() <= 1e-11 * np.abs(Kref).max()
    finally:
        mg.close()


def test_multi_errors_name_the_device_and_leave_the_object_usable(ctx, cfg3):
    p = cfg3["p"]
    with pytest.raises(kra.KoopmanHipError):
        Multi([0, 97])                                             # no such device
    mg = Multi([0, 0])
    try:
        z = np.zeros((64, 6)); u = np.random.default_rng(0).uniform(-1, 1, (64, 3))
        K = mg.fit(cfg3["dic"], z, z, u, [np.inf, np.inf])         # singular Gram: a basic solution, like the plain path (no error)
        assert np.isfinite(K).all()
        with pytest.raises(kra.KoopmanHipError) as e:
            mg.fit(("bilinear", 40, 3, [("poly", np.zeros((1, 40), np.uint8))], None), np.zeros((10, 40)), np.zeros((10, 40)), p["u"][:10], [np.inf])
        assert e.value.code != 0 and "device 0 (worker" in str(e.value)
        K2 = mg.fit(cfg3["dic"], p["alpha"], p["beta"], p["u"], [np.inf, np.inf])
        ref = kra.fit(ctx, cfg3["b"], cfg3["s"])[0]
        assert np.array_equal(K2[0].T, ref) and np.array_equal(K2[1].T, ref)
    finally:
        mg.close()


def test_lasso_values_that_need_the_homotopy_through_the_worker_threads(ctx, golden):
    """kp_multi_fit on the arm data's bilinear poly-2 dim_red dictionary (cond(G) 3.5e10): every worker's values end in the
    regularisation-path homotopy (kp_lasso_path.hip) inside its own thread and context - the same matrices as the plain context's
    kp_fit, value by value, to the accuracy of a path at that conditioning (the workers' batches stop at other thetas: 1e-5 max|K|;
    the budgets are met exactly either way)."""
    from test_gpu_lasso_path import _arm
    ks = kra.Ksysid(_arm(golden), ctx=ctx, model_type="bilinear", obs_type=["poly"], obs_degree=[2], snapshots=np.inf, lasso=[1.0], delays=0, dim_red=True)
    sp = ks.snapshotPairs
    s = ks._resident_snapshots(sp["alpha"], sp["beta"], sp["u"])
    Kls = kra.fit(ctx, ks.basis_dev, s)[0]
    N = ks.params["N"]
    las = np.array([0.6, 0.3, 0.05]) * np.abs(Kls).sum() / N
    ref = kra.fit(ctx, ks.basis_dev, s, las)
    assert ctx.timer(11) > 0.0
    exps = kra.poly_exponent_table(6, 2)[6:]
    mg = Multi([0, 0])
    try:
        Ks = mg.fit(("bilinear", 6, 3, [("poly", exps)], ks.basis["pcs"]), sp["alpha"], sp["beta"], sp["u"], las)
        for i in range(3):
            assert np.abs(Ks[i].T - ref[i]).max() <= 1e-5 * np.abs(ref[i]).max(), i
            assert abs(np.abs(Ks[i]).sum() - las[i] * N) <= 1e-11 * las[i] * N
    finally:
        mg.close()


def test_closing_the_multi_object_first_releases_its_children_and_keeps_old_host_views_valid():
    """kp_multi_destroy tears down the contexts its trajectory / controller objects point into (include/koopman_hip.h: destroy
    the objects first).  The Python owner enforces the order: Multi.close() releases the children it created, their own close()
    / __del__ afterwards is a no-op (round-4 advisor: a use-after-free at interpreter shutdown).  And a named page-locked block
    that has to grow leaves the arrays handed out before usable until close()."""
    from test_mex_gateway import _stacks
    Y, U, Yv, Uv, k = _stacks(nb=6, seed=5)
    Yn, Un, Yvn, Uvn = (np.ascontiguousarray(np.transpose(x, (2, 0, 1))) for x in (Y, U, Yv, Uv))
    mg = Multi([0, 0])
    tr = mg.traj_upload(Yn, Un, k, Yvn, Uvn)
    A = np.array([[0.9, 0.1], [0.0, 0.8]]); B = np.array([[0.0], [1.0]])
    mm = MultiMpc(mg, "linear", A, B, 4, np.array([[1.0, 0.0]]), 1.0, 1.0, np.array([0.1]))
    first = mg.host_array("K", (4, 8, 8))
    first[:] = 3.0
    second = mg.host_array("K", (64, 8, 8))                          # grows: a new block
    second[:] = 5.0
    assert (first == 3.0).all()                                      # the old view still points at live memory
    mg.close()                                                       # children first, then the retired blocks, then the contexts
    tr.close(); mm.close()                                           # no-ops: nothing is released twice
    tr.close(); mm.close()
    del tr, mm, mg


def test_distinct_devices_when_the_box_has_them(ctx, cfg3):
    """The one-caller path over DISTINCT devices (peer access, hipMemcpyPeerAsync between two GPUs, one worker thread per
    device): skipped on a one-GPU box, run by whoever first has several - the lasso grid (one Gram + peer broadcast of
    [G | C]) and the snapshot-sharded fit against the single-context results."""
    import ctypes as C
    from koopman_realizations_amd import _ffi as F
    n = C.c_int()
    F.check(F.lib().kp_device_count(C.byref(n)))
    if n.value < 2:
        pytest.skip("one visible device: the distinct-device branch needs two")
    ids = list(range(n.value))
    p, dic = cfg3["p"], cfg3["dic"]
    N = cfg3["b"].N
    lasso = list(np.logspace(-2, 2, 2 * n.value + 1))
    ref = kra.fit(ctx, cfg3["b"], cfg3["s"], lasso)
    mg = Multi(ids)
    try:
        K = mg.fit(dic, p["alpha"], p["beta"], p["u"], lasso)
        for i in range(len(lasso)):
            assert np.array_equal(K[i].T, ref[i]), i
        tm = mg.timers()
        assert tm[0, 0] > 0.0 and (tm[1:, 0] == 0.0).all()          # only device 0 uploads the snapshots
        Ks = mg.fit_sharded(dic, p["alpha"], p["beta"], p["u"])
        Kls = kra.fit(ctx, cfg3["b"], cfg3["s"])[0]
        assert np.abs(Ks[0].T - Kls).max() <= 1e-10 * np.abs(Kls).max()
    finally:
        mg.close()
    del N
