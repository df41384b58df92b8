"""GPU parity tests (through the C ABI): lasso (solve_KoopmanQP), get_model's M-projection,
validation rollouts, and the Ksysid host mirror end to end (example_sysid.m flow).
Tolerances: M-projection / models 1e-8 relative (two nested normal-equation solves);
rollouts 1e-9 absolute over 400 steps of a stable model; lasso 1e-6 relative (iterative,
both sides stop at a fixed-point tolerance; the reference's quadprog has 1e-8 tolerances)."""
import numpy as np
import pytest

import koopman_realizations_amd as kra
from koopman_realizations_amd import _ffi as F
from oracle import koopman_oracle as ko
from test_gpu_fit import make_basis
from conftest import synth_pairs

pytestmark = pytest.mark.gpu


def test_lasso_inactive_constraint_returns_least_squares(ctx):
    rng = np.random.default_rng(4)
    P = rng.standard_normal((300, 20)); Y = P @ (rng.standard_normal((20, 20)) * 0.3) + 0.01 * rng.standard_normal((300, 20))
    G, C = P.T @ P, P.T @ Y
    Kls = np.linalg.solve(G, C)
    K, it = ctx.fit_lasso(G, C, 2 * np.abs(Kls).sum())
    assert it == 0 and np.abs(K - Kls).max() < 1e-10


@pytest.mark.parametrize("frac", [0.7, 0.3])
def test_lasso_active_constraint_matches_oracle(ctx, frac):
    rng = np.random.default_rng(5)
    P = rng.standard_normal((400, 24)); Y = P @ (rng.standard_normal((24, 24)) * 0.3) + 0.01 * rng.standard_normal((400, 24))
    G, C = P.T @ P, P.T @ Y
    t = frac * np.abs(np.linalg.solve(G, C)).sum()
    K, it = ctx.fit_lasso(G, C, t, max_iter=50000, tol=1e-12)
    Ko = ko.koopman_lasso(G, C, t)
    assert abs(np.abs(K).sum() - t) < 1e-8 * t
    assert ko.lasso_kkt_residual(G, C, K, t) < 1e-6 * np.abs(C).max()
    assert np.abs(K - Ko).max() < 1e-6 * np.abs(Ko).max()


def test_fit_with_lasso_values(ctx):
    """train_models with a vector of lasso values (Ksysid.m:1372-1387): t = lasso*N."""
    p = synth_pairs(3000, 2, 1, seed=9)
    dic = ko.build_dictionary("bilinear", 2, 1, ["poly"], [2])
    b = make_basis(ctx, dic)
    snaps = kra.Snapshots(ctx, p["alpha"], p["beta"], p["u"])
    Px, Py = ko.px_py(dic, p)
    G, C = ko.gram(Px, Py)
    Kls = np.linalg.solve(G, C)
    l_act = 0.5 * np.abs(Kls).sum() / dic.N
    Ks = kra.fit(ctx, b, snaps, [np.inf, 1e4, l_act])
    assert np.abs(Ks[0] - Kls).max() < 1e-9 and np.abs(Ks[1] - Kls).max() < 1e-9
    Ko = ko.koopman_lasso(G, C, l_act * dic.N)
    assert np.abs(Ks[2] - Ko).max() < 1e-6 * np.abs(Ko).max()


def test_model_projection_matches_literal_get_model(ctx, arm):
    pairs = arm["pairs"]
    dic = ko.build_dictionary("linear", 6, 3, ["poly"], [3], pairs, dim_red=True)
    koop = ko.get_koopman(dic, pairs)
    ref = ko.get_model(dic, koop, 6)             # literal: L = (A Px' + B U')', M' = L \ Py
    Px, Py = ko.px_py(dic, pairs)
    G, C = ko.gram(Px, Py)
    A, B, M = ctx.model_project(koop["K"], G, C, dic.N, 3)
    for got, want in ((A, ref["A"]), (B, ref["B"]), (M, ref["M"])):
        assert np.abs(got - want).max() <= 1e-8 * max(1.0, np.abs(want).max())


@pytest.mark.parametrize("mt", ["linear", "bilinear"])
def test_rollout_matches_oracle(ctx, arm, mt):
    pairs = arm["pairs"]
    dic = ko.build_dictionary(mt, 6, 3, ["poly"], [3], pairs, dim_red=True)
    koop = ko.get_koopman(dic, pairs)
    mdl = ko.get_model(dic, koop, 6) if mt == "linear" else ko.get_blmodel(dic, koop, 6)
    val = {"t": arm["val"]["t"], "y": ko.scaledown(arm["scale"], "y", arm["val"]["y"]),
           "u": ko.scaledown(arm["scale"], "u", arm["val"]["u"])}
    res = ko.val_model(dic, mdl, val, 0)
    z0 = ko.econ_full(dic, val["y"][:1])[0]
    Y = ctx.rollout(mt, mdl["A"], mdl["B"], z0, val["u"], 6)
    Y[0] = val["y"][0]
    assert np.abs(Y - res["sim_y"]).max() < 1e-9
    # batch of 3 identical models == single
    Yb = ctx.rollout(mt, np.stack([mdl["A"]] * 3), np.stack([mdl["B"]] * 3), np.stack([z0] * 3), np.stack([val["u"]] * 3), 6)
    Yb[:, 0] = val["y"][0]
    assert (Yb[1] == Y).all() and (Yb[2] == Y).all()


@pytest.mark.parametrize("mt", ["linear", "bilinear"])
def test_rollout_wide_models_last_row_many_repetitions(ctx, mt):
    """The 256-thread rollout variant (N > 64) with n_out > 64: the last time step leaves the step loop before its
    barrier, so the collected outputs of that step need their own barrier before other waves copy them out (a data race
    in round 1).  Y[T-1] is checked over many repetitions and chunk-boundary lengths, T = 1 included."""
    rng = np.random.default_rng(11)
    N, m, n_out = 96, 2, 80
    A = 0.95 * np.linalg.qr(rng.standard_normal((N, N)))[0]
    B = 0.1 * rng.standard_normal((N, m)) if mt == "linear" else 0.02 * rng.standard_normal((N, N * m))
    z0 = rng.uniform(-1, 1, N)
    for T in (1, 2, 7, 197, 198, 256, 257, 400):     # chunk length here: 197 steps
        U = rng.uniform(-1, 1, (T, m))
        z = z0.copy(); want = np.empty((T, n_out))
        for t in range(T):
            want[t] = z[:n_out]
            Bz = B if mt == "linear" else B @ np.kron(np.eye(m), z[:, None])     # Ksysid.m:1285-1295
            z = A @ z + Bz @ U[t]
        for rep in range(40 if T in (1, 7, 198, 257) else 3):
            Y = ctx.rollout(mt, A, B, z0, U, n_out)
            assert np.abs(Y - want).max() < 1e-10, (T, rep, np.abs(Y - want).max(axis=1).argmax())


@pytest.mark.parametrize("mt", ["linear", "bilinear", "nonlinear"])
def test_ksysid_mirror_example_sysid_flow(ctx, golden, arm, mt):
    """example_sysid.m:22-65 through the host mirror: constructor, train_models, val_*."""
    g = golden["arm_data"]
    lens = g["train_len"]; off = np.concatenate([[0], np.cumsum(lens)])
    train = [{"t": g["train_t"][a:b], "y": g["train_y"][a:b], "u": g["train_u"][a:b]} for a, b in zip(off[:-1], off[1:])]
    val = [{"t": g["val_t"], "y": g["val_y"], "u": g["val_u"]}]
    ks = kra.Ksysid({"train": train, "val": val}, ctx=ctx, model_type=mt, obs_type=["poly"], obs_degree=[3], snapshots=np.inf,
                    lasso=[np.inf], delays=0, dim_red=True)
    np.testing.assert_allclose(ks.params["scale"]["u_factor"], arm["scale"]["u_factor"])
    assert ks.snapshotPairs["alpha"].shape == (11999, 6)
    assert ks.params["N"] == (88 if mt == "nonlinear" else 34)          # stored Z widths
    # the host mirror's PCA basis spans the oracle's (same columns up to rounding)
    dic = ko.build_dictionary(mt, 6, 3, ["poly"], [3], arm["pairs"], dim_red=True)
    # (the mirror's pca: covariance from the fused Gram kernel + Jacobi eigenvectors on the device; the oracle's: LAPACK
    #  SVD of the lifted matrix.  The last retained axes have small, close eigenvalues, where both are only determined to
    #  eps * lambda_1 / gap.)
    assert ks.basis["pcs"].shape == dic.pcs.shape
    assert np.abs(ks.basis["pcs"] - dic.pcs).max() < 1e-7
    assert np.abs(ks.basis["pcs"][:, :8] - dic.pcs[:, :8]).max() < 1e-10
    host = kra.Ksysid({"train": train, "val": val}, ctx=ctx, model_type=mt, obs_type=["poly"], obs_degree=[3], snapshots=np.inf,
                      lasso=[np.inf], delays=0, dim_red=True, _pca_host=True)
    assert np.abs(host.basis["pcs"] - dic.pcs).max() < 1e-9               # host cross-check path = the oracle's definition
    ks.train_models()
    dic = ko.Dictionary(dic.model_type, dic.nzeta, dic.m, dic.basis, ks.basis["pcs"])     # same axes for the model comparison
    Px, Py = ko.px_py(dic, arm["pairs"])
    Kref = ko.koopman_ls(Px, Py)      # least squares is invariant to the row permutation of the snapshot draw
    assert np.abs(ks.model["K"] - Kref).max() <= 2e-8 * np.abs(Kref).max()
    vd = ks.valdata[0]
    if mt == "linear":
        res = ks.val_model(ks.model, vd)
        ref = ko.val_model(dic, ko.get_model(dic, {"K": Kref, "Px": Px[:, :34], "Py": Py[:, :34], "u": arm["pairs"]["u"]}, 6), vd, 0)
    elif mt == "bilinear":
        res = ks.val_BLmodel(ks.model, vd)
        ref = ko.val_model(dic, ko.get_blmodel(dic, {"K": Kref}, 6), vd, 0)
    else:
        res = ks.val_NLmodel(ks.model, vd)
        ref = ko.val_model(dic, ko.get_nlmodel(dic, {"K": Kref}, 6), vd, 0)
    assert np.abs(res["sim"]["y"] - ref["sim_y"]).max() < 1e-5      # 400-step rollouts amplify the 1e-8 model difference
    e_ref = ko.get_error(ref["sim_y"], ref["real_y"], arm["scale"])
    assert abs(res["error"]["euclid_mean"] - e_ref["euclid_mean"]) < 1e-5
    # lift handles mirror the reference's lift.* (Ksysid.m:1615-1618)
    zeta = vd["y"][5]
    v = np.concatenate([zeta, vd["u"][5]]) if mt == "nonlinear" else zeta
    np.testing.assert_allclose(ks.lift.econ_full(v), ko.econ_full(dic, v[None, :])[0], atol=1e-9)


@pytest.mark.parametrize("kind,deg,nfull,dim_red", [("hermite", 3, 1 + 3 + 1, True), ("fourier_sparser", 1, 1 + 2 + 1, False)])
def test_ksysid_mirror_other_dictionaries(ctx, golden, kind, deg, nfull, dim_red):
    """README.txt:99-109 lists hermite and fourier_sparser among obs_type; constructor + fit + validation
    through the host mirror on a shipped random system (1-D state), against the oracle."""
    g = golden["rand_systems"]
    n = 1001
    train = [{"t": g["s0_train_t"][i * n:(i + 1) * n], "y": g["s0_train_y"][i * n:(i + 1) * n], "u": g["s0_train_u"][i * n:(i + 1) * n]}
             for i in range(9)]
    val = [{"t": g["s0_val_t"], "y": g["s0_val_y"], "u": g["s0_val_u"]}]
    ks = kra.Ksysid({"train": train, "val": val}, ctx=ctx, model_type="linear", obs_type=[kind], obs_degree=[deg],
                    snapshots=np.inf, lasso=[np.inf], delays=0, dim_red=dim_red)   # hermite repeats zeta (H1 = 2x): needs dim_red
    assert ks.basis_dev.nfull == nfull
    ks.train_models()
    dic = ko.Dictionary("linear", 1, 1, ko.make_basis(1, [kind], [deg]), ks.basis["pcs"] if dim_red else None)
    np.testing.assert_allclose(ks.lift.econ_full(np.array([0.3])), ko.econ_full(dic, np.array([[0.3]]))[0], atol=1e-12)
    res = ks.val_model(ks.model, ks.valdata[0])
    assert np.isfinite(res["error"]["mean"]).all()


def test_linear_lasso_with_delays_pins_the_shift_columns(ctx, golden):
    """Ksysid.m:1139-1164 through the host mirror: linear model, delays = 1, active L1 constraint."""
    g = golden["rand_systems"]
    n = 1001
    train = [{"t": g["s0_train_t"][i * n:(i + 1) * n], "y": g["s0_train_y"][i * n:(i + 1) * n], "u": g["s0_train_u"][i * n:(i + 1) * n]}
             for i in range(9)]
    val = [{"t": g["s0_val_t"], "y": g["s0_val_y"], "u": g["s0_val_u"]}]
    ks = kra.Ksysid({"train": train, "val": val}, ctx=ctx, model_type="linear", obs_type=["poly"], obs_degree=[2],
                    snapshots=np.inf, lasso=[0.5], delays=1, dim_red=False)
    p = ks.params
    assert p["nd"] == 1 and p["nzeta"] == 3
    koop = ks.get_Koopman(ks.snapshotPairs, 0.5, want_PxPy=False)
    K = koop["K"]
    N, m = p["N"], p["m"]
    c0, c1, ones = ko.delay_pins(p["n"], m, 1, N)
    pinned = np.zeros((N + m, c1 - c0))
    for r, c in ones:
        pinned[r, c - c0] = 1.0
    assert np.array_equal(K[:, c0:c1], pinned)
    dic = ko.build_dictionary("linear", p["nzeta"], m, ["poly"], [2])
    Px, Py = ko.px_py(dic, ks.snapshotPairs)
    G, C = ko.gram(Px, Py)
    Kref = ko.koopman_lasso_delays(G, C, 0.5 * N, p["n"], m, 1, N)
    assert np.abs(np.abs(K).sum() - 0.5 * N) < 1e-8            # the constraint is active
    assert np.abs(K - Kref).max() <= 1e-5 * max(1.0, np.abs(Kref).max())


@pytest.mark.parametrize("which", ["bilin", "lin"])
def test_device_pca_and_econ_lift_reproduce_stored_Z(ctx, golden, arm, which):
    """The reference's stored closed-loop results hold Z = lift.econ_full(scaledown.y(Y)) (Ksim.m:256): 300 x 34 golden
    vectors that pin scaling, pair selection, monomial order, the pca (sign rule, 99 % cut) and the econ lift.  Here the
    whole chain runs through the product: Ksysid mirror, covariance from the fused Gram kernel, Jacobi eigenvectors and
    the lift kernel on the device."""
    g = golden["arm_data"]; r = golden["arm_blockM"]
    lens = g["train_len"]; off = np.concatenate([[0], np.cumsum(lens)])
    train = [{"t": g["train_t"][a:b], "y": g["train_y"][a:b], "u": g["train_u"][a:b]} for a, b in zip(off[:-1], off[1:])]
    val = [{"t": g["val_t"], "y": g["val_y"], "u": g["val_u"]}]
    ks = kra.Ksysid({"train": train, "val": val}, ctx=ctx, model_type="bilinear" if which == "bilin" else "linear", obs_type=["poly"],
                    obs_degree=[3], snapshots=np.inf, lasso=[np.inf], delays=0, dim_red=True)
    assert ks.basis["pcs"].shape == (84, 27) and ks.params["N"] == 34
    Z = ks.lift.econ_full(ks.scaledown_y(r[which + "_Y"][:300]))
    assert np.abs(Z - r[which + "_Z"]).max() < 1e-11                      # measured 8e-14 (the oracle's LAPACK SVD chain: 1.4e-14)


@pytest.mark.parametrize("n", [1, 2, 7, 40, 41, 85, 220, 257, 600])
def test_sym_eig_matches_lapack(ctx, n):
    """kp_sym_eig (the eigensolver behind `pca`, Ksysid.m:1498): one-workgroup Jacobi up to n = 40, the multi-workgroup form
    (ping-pong S, one grid barrier per round; odd n: one index sits out every round) above - against LAPACK on a
    covariance with a spectrum decaying over six decades."""
    rng = np.random.default_rng(n)
    X = rng.standard_normal((max(4 * n, 50), n)) * np.logspace(0, -6, n)
    S = np.atleast_2d(np.cov(X, rowvar=False))
    w, V, sweeps = ctx.sym_eig(S)
    wl = np.linalg.eigvalsh(S)[::-1]
    scale = np.abs(wl).max()
    assert 0 < sweeps < 30
    assert np.abs(w - wl).max() <= 1e-13 * scale * max(1, n / 50)
    assert np.abs(S @ V - V * w).max() <= 1e-12 * scale and np.abs(V.T @ V - np.eye(n)).max() <= 1e-12


def test_train_models_with_a_vector_of_lasso_values_and_sampled_snapshots(ctx):
    """train_models(lasso vector) (Ksysid.m:1344-1389): one koopData / candidate per value, `model` = the first candidate; and the
    constructor's `snapshots` < Inf (:960-975): that many pairs drawn without replacement from the good ones."""
    from tests._loaded_system import make_trials
    trials = [{k: v for k, v in t.items() if k != "w"} for t in make_trials(6, 200, nw=1, seed=21)]
    lass = [0.3, 2.0, np.inf]
    ks = kra.Ksysid({"train": trials[:5], "val": trials[5:]}, ctx=ctx, model_type="bilinear", obs_type=["poly"], obs_degree=[2], lasso=lass,
                    snapshots=400)
    sp = ks.snapshotPairs
    assert sp["alpha"].shape == (400, 2) and sp["u"].shape == (400, 1)
    full = kra.Ksysid({"train": trials[:5], "val": trials[5:]}, ctx=ctx, model_type="bilinear", obs_type=["poly"], obs_degree=[2], lasso=lass)
    allp = {tuple(np.round(np.concatenate([a, b, u]), 12)) for a, b, u in zip(full.snapshotPairs["alpha"], full.snapshotPairs["beta"], full.snapshotPairs["u"])}
    mine = [tuple(np.round(np.concatenate([a, b, u]), 12)) for a, b, u in zip(sp["alpha"], sp["beta"], sp["u"])]
    assert len(set(mine)) == 400 and set(mine) <= allp                          # a subset of the good pairs, no repeats
    ks.train_models()
    assert isinstance(ks.candidates, list) and len(ks.candidates) == 3 and len(ks.koopData) == 3
    assert [c["lasso"] for c in ks.candidates] == [0.3, 2.0, 1e6] and ks.model is ks.candidates[0]
    dic = ko.build_dictionary("bilinear", 2, 1, ["poly"], [2])
    Px, Py = ko.px_py(dic, sp)
    G, C = ko.gram(Px, Py)
    Kls = np.linalg.solve(G, C)
    for kd, lv in zip(ks.koopData, [0.3, 2.0, 1e6]):
        K = kd["K"]
        t = lv * dic.N                                                         # :996
        if np.abs(Kls).sum() <= t:
            assert np.abs(K - Kls).max() < 1e-8 * max(1.0, np.abs(Kls).max())
        else:
            Ko = ko.koopman_lasso(G, C, t)
            assert abs(np.abs(K).sum() - t) < 1e-6 * t and np.abs(K - Ko).max() < 1e-5 * max(1.0, np.abs(Ko).max())
    assert np.abs(ks.candidates[1]["A"] - ks.koopData[1]["K"].T[:dic.N, :dic.N]).max() == 0      # get_BLmodel slices K' (:1261)


@pytest.mark.parametrize("mt", ["linear", "bilinear", "nonlinear"])
def test_koopdata_px_py_are_the_first_N_row_columns_and_materialise_lazily(ctx, golden, mt):
    """koopData.Px / .Py = Px(:, 1:N), Py(:, 1:N) (Ksysid.m:1085-1086): the mirror lifts the N econ columns (linear /
    bilinear rows start with psi(x), :1049-1063) at first access instead of the W-wide row block on every fit; same numbers
    as the oracle's literal rows."""
    g = golden["rand_systems"]
    n = 1001
    train = [{"t": g["s0_train_t"][i * n:(i + 1) * n], "y": g["s0_train_y"][i * n:(i + 1) * n], "u": g["s0_train_u"][i * n:(i + 1) * n]}
             for i in range(3)]
    val = [{"t": g["s0_val_t"], "y": g["s0_val_y"], "u": g["s0_val_u"]}]
    ks = kra.Ksysid({"train": train, "val": val}, ctx=ctx, model_type=mt, obs_type=["poly"], obs_degree=[3],
                    snapshots=np.inf, lasso=[np.inf], delays=0, dim_red=False)
    p = ks.params
    koop = ks.get_Koopman(ks.snapshotPairs)
    assert "Px" in koop and "Py" in koop and not dict.__contains__(koop, "Px")          # promised, not yet made
    assert set(koop.keys()) >= {"K", "Px", "Py", "u", "alpha"}
    dic = ko.build_dictionary(mt, p["nzeta"], p["m"], ["poly"], [3])
    Px, Py = ko.px_py(dic, ks.snapshotPairs)
    N = p["N"]
    assert koop["Px"].shape == (Px.shape[0], N) and dict.__contains__(koop, "Px")
    assert np.abs(koop["Px"] - Px[:, :N]).max() < 1e-13 and np.abs(koop["Py"] - Py[:, :N]).max() < 1e-13
    assert koop.get("Px") is koop["Px"] and koop.get("nothing", 7) == 7
    with pytest.raises(KeyError):
        koop["nothing"]
