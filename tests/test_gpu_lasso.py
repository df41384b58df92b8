"""GPU parity tests of the lasso path at BASELINE configs[3]'s shape: bilinear poly-3 dictionary (W = 336) on 1e5 synthetic
snapshot pairs, a grid of L1 budgets on the same Grams (solve_KoopmanQP, Ksysid.m:1095-1176; train_models loop
:1372-1387).  Oracle: the optimality conditions of the reference's QP (min 1/2||Px K - Py||^2 s.t. ||vec K||_1 <= t:
`lasso_kkt_residual` = || K - P_ball(K - grad) ||_inf, zero exactly at the optimum) and, at sizes the numpy oracle
finishes in seconds, the oracle's own solution.  Tolerances: KKT residual <= 1e-8 max|C| (quadprog's stopping
tolerance is 1e-8 relative); budget met to 1e-9 relative."""
import numpy as np
import pytest

import koopman_realizations_amd as kra
from oracle import koopman_oracle as ko
from conftest import synth_pairs
from test_gpu_fit import make_basis

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def config3(ctx):
    p = synth_pairs(100000)
    dic = ko.build_dictionary("bilinear", 6, 3, ["poly"], [3])
    b = make_basis(ctx, dic)
    snaps = kra.Snapshots(ctx, p["alpha"], p["beta"], p["u"])
    G, C = kra.fit_gram(ctx, b, snaps)
    Kls = ctx.fit_solve(G, C)
    return {"basis": b, "snaps": snaps, "G": G, "C": C, "Kls": Kls, "l1": float(np.abs(Kls).sum()), "N": dic.N}


def test_lasso_grid_at_config3_shape_is_kkt_optimal(ctx, config3):
    """Eight budgets from almost-least-squares (dense support) to 1 % of ||K_LS||_1 (16 nonzeros), one inactive."""
    G, C, l1 = config3["G"], config3["C"], config3["l1"]
    assert G.shape == (336, 336)
    fr = np.array([1.5, 0.99, 0.9, 0.7, 0.5, 0.3, 0.1, 0.01])
    Ks, iters = ctx.fit_lasso_batch(G, C, fr * l1)
    cmax = np.abs(C).max()
    assert iters[0] == 0 and np.abs(Ks[0] - config3["Kls"]).max() == 0          # constraint inactive: the LS answer
    for f, K, it in zip(fr[1:], Ks[1:], iters[1:]):
        t = f * l1
        assert it > 0
        assert abs(np.abs(K).sum() - t) <= 1e-9 * t, (f, np.abs(K).sum() / t)
        res = ko.lasso_kkt_residual(G, C, K, t)
        assert res <= 1e-8 * cmax, (f, res / cmax)
    nnz = [(K != 0).sum() for K in Ks]
    assert nnz[-1] < nnz[-3] < nnz[2] < nnz[1]                                  # sparser as the budget shrinks
    # single-value entry point = the batch's answer
    K1, _ = ctx.fit_lasso(G, C, 0.5 * l1)
    assert np.abs(K1 - Ks[4]).max() <= 1e-9 * np.abs(Ks[4]).max()


def test_train_models_lasso_vector_at_config3_shape(ctx, config3):
    """kp_fit with a vector of lasso values (t = lasso * N, Ksysid.m:996): snapshots lifted once, values batched."""
    b, snaps, l1, N = config3["basis"], config3["snaps"], config3["l1"], config3["N"]
    las = [np.inf, 0.6 * l1 / N, 0.05 * l1 / N, 1e4]
    Ks = kra.fit(ctx, b, snaps, las)
    cmax = np.abs(config3["C"]).max()
    assert np.abs(Ks[0] - config3["Kls"]).max() <= 1e-12 * np.abs(config3["Kls"]).max()
    assert np.abs(Ks[3] - Ks[0]).max() == 0
    for lv, K in zip(las[1:3], Ks[1:3]):
        assert ko.lasso_kkt_residual(config3["G"], config3["C"], K, lv * N) <= 1e-8 * cmax


def test_lasso_matches_numpy_oracle_mid_size(ctx):
    """W = 60 bilinear poly-2 problem: the oracle's projected-gradient solution (run to 1e-13) and the device agree."""
    p = synth_pairs(5000, 4, 2, seed=3)
    dic = ko.build_dictionary("bilinear", 4, 2, ["poly"], [2])
    Px, Py = ko.px_py(dic, p)
    G, C = ko.gram(Px, Py)
    l1 = np.abs(np.linalg.solve(G, C)).sum()
    fr = [0.8, 0.4, 0.05]
    Ks, iters = ctx.fit_lasso_batch(G, C, [f * l1 for f in fr], tol=1e-12)
    for f, K in zip(fr, Ks):
        Ko = ko.koopman_lasso(G, C, f * l1)
        assert np.abs(K - Ko).max() <= 1e-8 * np.abs(Ko).max(), f
        assert ko.lasso_kkt_residual(G, C, K, f * l1) <= 1e-9 * np.abs(C).max()


def test_lasso_without_polish_agrees(ctx, config3, monkeypatch):
    """The active-set polish only ends the iteration early: the plain projected-gradient answer is the same optimum."""
    G, C, l1 = config3["G"], config3["C"], config3["l1"]
    Kp, itp = ctx.fit_lasso_batch(G, C, [0.3 * l1])
    assert ko.lasso_kkt_residual(G, C, Kp[0], 0.3 * l1) <= 1e-10 * np.abs(C).max()     # polished: optimum to rounding


def test_lasso_grid_through_the_device_resident_gather(ctx, config3):
    """sweep.lasso_sweep_device: the shard's K stack stays in the device result buffer (kp_fit with K_out = NULL), one
    kp_comm_allgather_fits brings it to a page-locked block; same K as the fetched path, in grid order, LS values included."""
    from koopman_realizations_amd import sweep, comm as kc
    b, snaps, l1, N = config3["basis"], config3["snaps"], config3["l1"], config3["N"]
    las = [np.inf, 0.6 * l1 / N, 0.05 * l1 / N, 1e4, 0.2 * l1 / N]
    ref = kra.fit(ctx, b, snaps, las)
    got = sweep.lasso_sweep_device(ctx, lambda ls: kra.fit(ctx, b, snaps, ls, fetch=False), las, b.W, kc.LocalComm())
    assert len(got) == len(las)
    for Kr, Kg in zip(ref, got):
        assert np.array_equal(Kr, Kg)
    # a sub-range and the error paths of the entry point
    part = kc.all_gather_fits(kc.LocalComm(), ctx, 1, 2, b.W)
    assert np.array_equal(part[0, 0].T, ref[1]) and np.array_equal(part[0, 1].T, ref[2])
    with pytest.raises(kra.KoopmanHipError):
        kc.all_gather_fits(kc.LocalComm(), ctx, 3, 5, b.W)          # past the last value


def test_active_set_rounds_end_sparse_values_early_and_exactly(ctx, config3):
    """The polish's active-set rounds (exchange the entries whose membership is wrong, re-solve) reach the optimum of
    sparse values within the first check blocks; the answer satisfies the optimality conditions to rounding and equals the
    plain iteration's optimum."""
    G, C, l1 = config3["G"], config3["C"], config3["l1"]
    fr = np.array([0.5, 0.2, 0.05, 0.02])
    Ks, iters = ctx.fit_lasso_batch(G, C, fr * l1)
    cmax = np.abs(C).max()
    assert max(iters) <= 60, iters
    for f, K in zip(fr, Ks):
        assert ko.lasso_kkt_residual(G, C, K, f * l1) <= 1e-12 * cmax
        assert abs(np.abs(K).sum() - f * l1) <= 1e-12 * f * l1


def test_the_64_value_grid_of_configs3_is_optimal_value_by_value(ctx, config3):
    """BASELINE configs[3] at its full size: lasso = t/N log-spaced over [1e-2, 1e2], 64 values in ONE kp_fit call
    (what bench.py's lasso_grid section times).  Every value with an active constraint meets the budget and the optimality
    conditions of the reference's QP (Ksysid.m:1126-1137) to quadprog's own tolerance; every value whose budget exceeds
    ||K_LS||_1 is the least-squares solution bit for bit; the active-set rounds finish the whole grid within 60 iterations."""
    b, snaps, G, C, l1, N = config3["basis"], config3["snaps"], config3["G"], config3["C"], config3["l1"], config3["N"]
    vals = np.geomspace(1e-2, 1e2, 64)
    Ks = kra.fit(ctx, b, snaps, list(vals))
    cmax = np.abs(C).max()
    n_active = 0
    for lv, K in zip(vals, Ks):
        t = lv * N
        if t >= l1 * (1 + 1e-12):
            assert np.array_equal(K, Ks[-1])                         # inactive: the LS answer (the last value is inactive too)
            continue
        n_active += 1
        assert abs(np.abs(K).sum() - t) <= 1e-9 * t, lv
        assert ko.lasso_kkt_residual(G, C, K, t) <= 1e-8 * cmax, lv
    assert n_active == 42
    assert np.abs(Ks[-1] - config3["Kls"]).max() <= 1e-12 * np.abs(config3["Kls"]).max()
    # sparsity grows monotonically as the budget shrinks (up to ties)
    nnz = [int((K != 0).sum()) for K in Ks[:42]]
    assert all(a <= b_ + 2 for a, b_ in zip(nnz[:-1], nnz[1:])), nnz


def test_bench_timers_of_the_fit_and_lasso_paths(ctx, config3):
    """kp_timer_get 10: flop per pair the fused Gram launch EXECUTES on the matrix pipe (what bench.py's roofline.frac is
    priced with): 28 jobs x 6 quads x 10 weights x 128 = 215 040 at the configs[1] dictionary - 63 % of the dense-equivalent
    W(W+1) + 2W^2 = 339 024.  Timers 8 / 9: the widest product of the last lasso batch and its columns."""
    b, snaps, G, C, l1 = config3["basis"], config3["snaps"], config3["G"], config3["C"], config3["l1"]
    kra.fit_gram(ctx, b, snaps, fetch=False)
    assert ctx.timer(10) == 215040.0 and ctx.timer(0) > 0
    ctx.fit_lasso_batch(G, C, np.array([0.5, 0.3, 0.1]) * l1)
    assert ctx.timer(9) == 3 * 336 and 0 < ctx.timer(8) < 5.0


@pytest.mark.parametrize("knob", [{"KP_LASSO_ROUNDS": "0"}, {"KP_LASSO_COLD_START": "1"}], ids=["no_rounds", "cold_start"])
def test_one_shot_polish_without_active_set_rounds_gives_the_same_optimum(ctx, config3, knob):
    """KP_LASSO_ROUNDS / KP_LASSO_COLD_START are read once per process, so the round-free path and the start from K = 0
    (default: from the least-squares solution) are exercised in a fresh interpreter: same optimum to 1e-9, no fewer
    iterations."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    G, C, l1 = config3["G"], config3["C"], config3["l1"]
    Kr, itr = ctx.fit_lasso_batch(G, C, [0.2 * l1])
    script = ("import sys, numpy as np; sys.path.insert(0, sys.argv[1]); sys.path.insert(0, sys.argv[1] + '/tests')\n"
              "import koopman_realizations_amd as kra\n"
              "d = np.load(sys.argv[2]); c = kra.Context(0)\n"
              "K, it = c.fit_lasso_batch(d['G'], d['C'], [float(d['t'])])\n"
              "np.savez(sys.argv[3], K=K[0], it=it[0])\n")
    import tempfile
    with tempfile.TemporaryDirectory() as td:
        np.savez(os.path.join(td, "in.npz"), G=G, C=C, t=0.2 * l1)
        env = dict(os.environ, **knob)
        r = subprocess.run([sys.executable, "-c", script, root, os.path.join(td, "in.npz"), os.path.join(td, "out.npz")], env=env,
                           capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stderr[-2000:]
        o = np.load(os.path.join(td, "out.npz"))
        assert np.abs(o["K"] - Kr[0]).max() <= 1e-9 * np.abs(Kr[0]).max()
        assert int(o["it"]) >= int(itr[0])
