"""GPU tests of edge cases and error behaviour at the C-ABI boundary."""
import numpy as np
import pytest

import koopman_realizations_amd as kra
from koopman_realizations_amd import _ffi as F
from oracle import koopman_oracle as ko
from test_gpu_fit import make_basis
from conftest import synth_pairs

pytestmark = pytest.mark.gpu


def test_empty_snapshot_set_gives_zero_grams_and_reports_singularity(ctx):
    dic = ko.build_dictionary("bilinear", 2, 1, ["poly"], [2])
    b = make_basis(ctx, dic)
    s = kra.Snapshots(ctx, np.zeros((0, 2)), np.zeros((0, 2)), np.zeros((0, 1)))
    G, C = kra.fit_gram(ctx, b, s)
    assert G.shape == (dic.W, dic.W) and not G.any() and not C.any()
    K = kra.fit(ctx, b, s)[0]                      # rank 0: like MATLAB's `\\` on an empty Px, zeros (and a warning)
    assert not K.any() and ctx.last_rank() == 0
    assert b.lift(F.LIFT_ECON, np.zeros((0, 2))).shape == (0, dic.N)


def test_dimension_mismatch_and_bad_arguments_are_rejected(ctx):
    dic = ko.build_dictionary("linear", 3, 2, ["poly"], [2])
    b = make_basis(ctx, dic)
    s = kra.Snapshots(ctx, np.zeros((10, 2)), np.zeros((10, 2)), np.zeros((10, 2)))     # nzeta 2 != 3
    with pytest.raises(kra.KoopmanHipError) as e:
        kra.fit_gram(ctx, b, s)
    assert e.value.code == F.KP_ERR_ARG
    with pytest.raises(kra.KoopmanHipError):
        kra.Basis(ctx, "linear", 40, 1, [("poly", np.zeros((0, 40), np.uint8))])   # > 32 variables
    with pytest.raises(kra.KoopmanHipError):
        kra.device.Mpc(ctx, "linear", np.eye(3), np.ones((3, 1)), 70, np.eye(3)[:1], 1.0, 1.0, [0.1])   # m*Np > 64


def test_single_snapshot_and_constant_columns(ctx):
    """One pair only; constant (zero-range) data columns scale to themselves (Ksysid.m:198-204)."""
    dic = ko.build_dictionary("nonlinear", 2, 1, ["poly"], [3])
    b = make_basis(ctx, dic)
    p = synth_pairs(1, 2, 1, seed=4)
    s = kra.Snapshots(ctx, p["alpha"], p["beta"], p["u"])
    G, C = kra.fit_gram(ctx, b, s)
    Px, Py = ko.px_py(dic, p)
    assert np.abs(G - Px.T @ Px).max() < 1e-14 and np.abs(C - Px.T @ Py).max() < 1e-14


def test_delay_embedding_fit_matches_oracle(ctx, golden):
    """delays = 1: zeta = [y, y_{-1}, u_{-1}] (Ksysid.m:868-907), nzeta = 15."""
    g = golden["arm_data"]
    lens = g["train_len"]; off = np.concatenate([[0], np.cumsum(lens)])
    train = [{"t": g["train_t"][a:b], "y": g["train_y"][a:b], "u": g["train_u"][a:b]} for a, b in zip(off[:2], off[1:3])]
    val = [{"t": g["val_t"], "y": g["val_y"], "u": g["val_u"]}]
    ks = kra.Ksysid({"train": train, "val": val}, ctx=ctx, model_type="linear", obs_type=["poly"], obs_degree=[1], delays=1,
                    snapshot_seed=3)
    assert ks.params["nzeta"] == 15 and ks.params["N"] == 16
    # oracle on the same (permuted) pairs
    dic = ko.build_dictionary("linear", 15, 3, ["poly"], [1])
    Px, Py = ko.px_py(dic, ks.snapshotPairs)
    # delayed coordinates make Px rank deficient here? (y and its delay are distinct columns: full rank)
    Kref = ko.koopman_ls(Px, Py)
    ks.train_models()
    cond = np.linalg.cond(Px)
    assert np.abs(ks.model["K"] - Kref).max() <= max(1e-9, 50 * cond ** 2 * 2.2e-16) * np.abs(Kref).max()
    merged = ko.merge_trials(train)
    sd, sc = ko.get_scale(merged)
    pairs = ko.snapshot_pairs(sd, 1)
    assert pairs["alpha"].shape[1] == 15 and ks.snapshotPairs["alpha"].shape[0] == pairs["alpha"].shape[0]
    # same multiset of rows (the mirror draws a seeded permutation)
    a = np.sort(ks.snapshotPairs["alpha"][:, 0]); bsorted = np.sort(pairs["alpha"][:, 0])
    assert np.allclose(a, bsorted)


def test_wide_dictionary_falls_back_to_general_kernel(ctx):
    """Fourier block (not a monomial dictionary) and > 96 columns: the general kernel is used."""
    p = synth_pairs(500, 2, 1, seed=8)
    dic = ko.build_dictionary("bilinear", 2, 1, ["poly", "fourier"], [3, 2])
    assert dic.N == 2 + 7 + 24 + 1
    b = make_basis(ctx, dic)
    s = kra.Snapshots(ctx, p["alpha"], p["beta"], p["u"])
    G, C = kra.fit_gram(ctx, b, s)
    Px, Py = ko.px_py(dic, p)
    assert np.abs(G - Px.T @ Px).max() <= 1e-12 * np.abs(Px.T @ Px).max()
    assert np.abs(C - Px.T @ Py).max() <= 1e-12 * np.abs(Px.T @ Px).max()


def test_lasso_on_a_rank_deficient_gram_uses_the_psd_guard(ctx):
    """Two identical state columns make a third of the degree-2 dictionary dependent: the Gram matrix is singular (its
    smallest eigenvalue is rounding noise of either sign).  solve_KoopmanQP adds 1e-6 to the diagonal when an eigenvalue is
    negative (Ksysid.m:1117-1120); the device does the same when the factorisation breaks down, and the answers are the
    optima of THAT problem: KKT residual of the guarded QP at rounding level for active budgets, the guarded least-squares
    solution for an inactive one."""
    rng = np.random.default_rng(4)
    Ns = 4000
    a = rng.uniform(-1, 1, (Ns, 3)); a[:, 2] = a[:, 0]
    u = rng.uniform(-1, 1, (Ns, 2))
    b_ = np.clip(a + 0.05 * np.tanh(np.hstack([a, u]) @ rng.standard_normal((5, 3))), -1, 1); b_[:, 2] = b_[:, 0]
    basis = kra.Basis(ctx, "bilinear", 3, 2, [("poly", kra.poly_exponent_table(3, 2)[3:])])
    snaps = kra.Snapshots(ctx, a, b_, u)
    G, C = kra.fit_gram(ctx, basis, snaps)
    assert np.linalg.matrix_rank(G) < basis.W
    Gg = G + 1e-6 * np.eye(basis.W)
    Kg = np.linalg.solve(Gg, C)
    l1 = np.abs(Kg).sum()
    Ks, its = ctx.fit_lasso_batch(G, C, [2.0 * l1, 0.8 * l1, 0.3 * l1, 0.05 * l1])
    assert all(np.isfinite(K).all() for K in Ks)
    assert np.abs(Ks[0] - Kg).max() <= 1e-6 * np.abs(Kg).max()                 # inactive budget: the guarded LS solution
    for K, fr in zip(Ks[1:], (0.8, 0.3, 0.05)):
        assert abs(np.abs(K).sum() / (fr * l1) - 1.0) <= 1e-9
        assert ko.lasso_kkt_residual(Gg, C, K, fr * l1) <= 1e-9 * np.abs(C).max()
