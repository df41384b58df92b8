"""CPU tests: the oracle against the reference's own stored artefacts (golden vectors) and
its mathematical pins.  No GPU needed."""
import numpy as np
import pytest

from oracle import koopman_oracle as ko


def test_partitions_order_matches_reference_documentation():
    # partitions.m:206-219 order; SURVEY appendix A: n=3, d=2
    e = ko.partitions_ones(2, 3)
    assert e.tolist() == [[2, 0, 0], [1, 1, 0], [0, 2, 0], [1, 0, 1], [0, 1, 1], [0, 0, 2]]
    e = ko.poly_exponents(6, 3)
    assert e.shape == (83, 6)                      # N = C(9,3) = 84 with the constant (Ksysid.m:641)
    assert (e[:6] == np.eye(6, dtype=int)).all()   # first nzeta monomials are zeta (Ksysid.m:488)
    assert e[6].tolist() == [2, 0, 0, 0, 0, 0] and e[7].tolist() == [1, 1, 0, 0, 0, 0]
    for nv, d in [(1, 13), (2, 4), (9, 3)]:
        assert ko.poly_exponents(nv, d).shape[0] == np.prod(np.arange(nv + 1, nv + d + 1)) // np.prod(np.arange(1, d + 1)) - 1


def test_scale_and_pairs_match_arm_file(arm):
    # factors and the 11 999 pairs of the shipped arm data file
    np.testing.assert_allclose(arm["scale"]["u_factor"], [2.8108652, 2.80934499, 2.802723], rtol=1e-8)
    np.testing.assert_allclose(arm["scale"]["y_factor"][:2], [0.33333333, 0.3193511], rtol=1e-7)
    assert arm["pairs"]["alpha"].shape == (11999, 6)
    assert np.abs(arm["scaled"]["y"]).max() <= 1 + 1e-12


@pytest.mark.parametrize("which", ["bilin", "lin"])
def test_econ_lift_reproduces_stored_Z(arm, golden, which):
    """Golden vector: res_bilin.Z / res_lin.Z (300x34) = lift.econ_full(scaledown.y(Y(k,:)))
    stored by Ksim.run_trial_mpc (Ksim.m:256).  Pins scaling, pair selection, monomial
    order, pca (incl. sign convention and the 99 % cut) and the econ lift."""
    r = golden["arm_blockM"]
    dic = ko.build_dictionary("bilinear" if which == "bilin" else "linear", 6, 3, ["poly"], [3], arm["pairs"], dim_red=True)
    assert dic.pcs.shape == (84, 27) and dic.N == 34
    Z = ko.econ_full(dic, ko.scaledown(arm["scale"], "y", r[which + "_Y"][:300]))
    assert np.abs(Z - r[which + "_Z"]).max() < 5e-14


def test_nonlinear_dictionary_width_matches_stored_Z(arm, golden):
    dn = ko.build_dictionary("nonlinear", 6, 3, ["poly"], [3], arm["pairs"], dim_red=True)
    assert dn.basis.nfull == 220 and dn.N == int(golden["arm_blockM"]["nonlin_Zwidth"]) == 88


def test_px_layouts():
    rng = np.random.default_rng(1)
    z = rng.uniform(-1, 1, (5, 2)); u = rng.uniform(-1, 1, (5, 2))
    for mt in ("linear", "bilinear", "nonlinear"):
        dic = ko.build_dictionary(mt, 2, 2, ["poly"], [2])
        P = ko.lift_rows(dic, z, u)
        assert P.shape == (5, dic.W)
        if mt == "bilinear":   # [psi, u1 psi, u2 psi]  Ksysid.m:510-511
            N = dic.N
            np.testing.assert_allclose(P[:, N:2 * N], P[:, :N] * u[:, [0]])
            np.testing.assert_allclose(P[:, 2 * N:], P[:, :N] * u[:, [1]])
            assert np.all(P[:, N - 1] == 1)
        if mt == "linear":
            np.testing.assert_allclose(P[:, -2:], u)
        if mt == "nonlinear":  # psi over [zeta;u]
            np.testing.assert_allclose(P[:, :4], np.hstack([z, u]))


def test_fourier_and_gaussian_blocks():
    rng = np.random.default_rng(2)
    V = rng.uniform(-1, 1, (4, 2))
    c = rng.uniform(-1, 1, (2, 3))
    b = ko.make_basis(2, ["fourier", "gaussian"], [1, 3], [c])
    Pf = ko.lift_full(b, V)
    assert Pf.shape == (4, 2 + 8 + 3 + 1)
    # kron(poop(:,1), poop(:,2)) with poop = [1; cos; sin], minus the leading 1 (Ksysid.m:718-724)
    c1, s1 = np.cos(2 * np.pi * V[:, 0]), np.sin(2 * np.pi * V[:, 0])
    c2, s2 = np.cos(2 * np.pi * V[:, 1]), np.sin(2 * np.pi * V[:, 1])
    exp = np.stack([c2, s2, c1, c1 * c2, c1 * s2, s1, s1 * c2, s1 * s2], axis=1)
    np.testing.assert_allclose(Pf[:, 2:10], exp, atol=1e-15)
    np.testing.assert_allclose(Pf[:, 10], np.exp(-((V - c[:, 0]) ** 2).sum(axis=1)))


def test_hermite_and_fourier_sparser_blocks():
    """Literal forms of get_hermite (Ksysid.m:806-817) and get_sinusoid (:768-786) on 2 variables."""
    rng = np.random.default_rng(4)
    V = rng.uniform(-1, 1, (5, 2)); x, y = V[:, 0], V[:, 1]
    b = ko.make_basis(2, ["hermite"], [2])
    Pf = ko.lift_full(b, V)
    assert Pf.shape == (5, 2 + 5 + 1) and b.nfull == 8
    # hermiteH: H1 = 2x, H2 = 4x^2 - 2; rows of partitions: [1 0],[0 1],[2 0],[1 1],[0 2]
    want = np.stack([2 * x, 2 * y, 4 * x ** 2 - 2, 4 * x * y, 4 * y ** 2 - 2], axis=1)
    np.testing.assert_allclose(Pf[:, 2:7], want, atol=1e-14)
    assert abs(ko.hermite_h(3, np.array([0.5]))[0] - (8 * 0.125 - 12 * 0.5)) < 1e-15
    b = ko.make_basis(2, ["fourier_sparser"], [1])
    Pf = ko.lift_full(b, V)
    # multipliers over [sin x, sin y, cos x, cos y]: rows [1 0 0 0],[0 1 0 0],[0 0 1 0],[0 0 0 1]
    want = np.stack([np.sin(2 * np.pi * x), np.sin(2 * np.pi * y), np.cos(2 * np.pi * x), np.cos(2 * np.pi * y)], axis=1)
    np.testing.assert_allclose(Pf[:, 2:6], want, atol=1e-15)
    b = ko.make_basis(2, ["fourier_sparser"], [2])
    Pf = ko.lift_full(b, V)
    assert Pf.shape == (5, 2 + 4 + 10 + 1)
    ex = b.blocks[0][1]
    r = [i for i, e in enumerate(ex) if list(e) == [1, 0, 0, 1]][0]      # sin(2 pi x) cos(2 pi y)
    np.testing.assert_allclose(Pf[:, 2 + r], np.sin(2 * np.pi * x) * np.cos(2 * np.pi * y), atol=1e-15)
    r = [i for i, e in enumerate(ex) if list(e) == [2, 0, 0, 0]][0]      # sin(4 pi x)
    np.testing.assert_allclose(Pf[:, 2 + r], np.sin(4 * np.pi * x), atol=1e-15)


def test_delay_rows_of_the_lasso_qp():
    """Ksysid.m:1139-1164 literally: bd_pos over vec rows n*Nm+1 : Nm*(n(nd+1)+m nd) with ones at the three index
    families; the restatement pins the same entries of K, and the free columns satisfy the reduced problem's KKT."""
    n, m, nd = 2, 1, 2
    nzeta = n * (nd + 1) + m * nd
    N = nzeta + 1                                     # poly degree 1 + constant
    Nm, nnd, mnd = N + m, n * nd, m * nd
    bd = np.zeros(Nm * (nnd + mnd))                   # 1-based MATLAB indices below
    for i in range(1, nnd + 1):
        bd[(Nm + 1) * (i - 1) + 1 - 1] = 1
    for i in range(1, m + 1):
        bd[Nm * nnd + N + (Nm + 1) * (i - 1) + 1 - 1] = 1
    for i in range(1, m * (nd - 1) + 1):
        bd[Nm * (nnd + m) + nnd + (Nm + 1) * (i - 1) + 1 - 1] = 1
    vecK = np.full(Nm * Nm, np.nan)
    vecK[n * Nm: Nm * (n * (nd + 1) + mnd)] = bd      # Ad_pos rows n*Nm+1 : Nm*(n(nd+1)+mnd)
    Kpin = vecK.reshape(Nm, Nm, order="F")
    c0, c1, ones = ko.delay_pins(n, m, nd, N)
    assert (c0, c1) == (n, nzeta)
    want = np.full((Nm, Nm), np.nan); want[:, c0:c1] = 0.0
    for r, c in ones:
        want[r, c] = 1.0
    assert np.array_equal(np.isnan(want), np.isnan(Kpin)) and np.nanmax(np.abs(want - Kpin)) == 0
    assert len(ones) == nnd + m + m * (nd - 1)
    # state delays shift [x; xd] down by one block: K[i, n+i] = 1
    assert all((i, n + i) in ones for i in range(nnd))
    rng = np.random.default_rng(5)
    P = rng.standard_normal((60, Nm)); Y = rng.standard_normal((60, Nm))
    G, C = P.T @ P, P.T @ Y
    free = [j for j in range(Nm) if not (c0 <= j < c1)]
    t = 0.5 * np.abs(np.linalg.solve(G, C[:, free])).sum() + len(ones)
    K = ko.koopman_lasso_delays(G, C, t, n, m, nd, N)
    assert np.nanmax(np.abs(K[:, c0:c1] - want[:, c0:c1])) == 0
    assert abs(np.abs(K).sum() - t) < 1e-9           # budget used exactly: pinned ones + active free part
    assert ko.lasso_kkt_residual(G, C[:, free], K[:, free], t - len(ones)) < 1e-8


def test_ls_fit_recovers_exact_linear_system():
    rng = np.random.default_rng(3)
    A = rng.standard_normal((3, 3)) * 0.3; B = rng.standard_normal((3, 2))
    x = rng.uniform(-1, 1, (400, 3)); u = rng.uniform(-1, 1, (400, 2))
    y = x @ A.T + u @ B.T
    dic = ko.build_dictionary("linear", 3, 2, ["poly"], [1])
    koop = ko.get_koopman(dic, {"alpha": x, "beta": y, "u": u})
    mdl = ko.get_model(dic, koop, 3)
    np.testing.assert_allclose(mdl["A_raw"][:3, :3], A, atol=1e-10)
    np.testing.assert_allclose(mdl["B_raw"][:3], B, atol=1e-10)
    np.testing.assert_allclose(mdl["M"] @ mdl["A_raw"], mdl["A"])


def test_lasso_inactive_equals_ls_and_active_is_kkt():
    rng = np.random.default_rng(4)
    P = rng.standard_normal((200, 6)); Y = P @ rng.standard_normal((6, 6)) * 0.5 + 0.01 * rng.standard_normal((200, 6))
    G, C = ko.gram(P, Y)
    Kls = np.linalg.solve(G, C)
    K = ko.koopman_lasso(G, C, np.abs(Kls).sum() * 2)
    np.testing.assert_allclose(K, Kls, atol=1e-9)
    t = np.abs(Kls).sum() * 0.5
    K = ko.koopman_lasso(G, C, t)
    assert abs(np.abs(K).sum() - t) < 1e-9
    assert ko.lasso_kkt_residual(G, C, K, t) < 1e-8


def test_qp_oracle_kkt_on_random_problems():
    rng = np.random.default_rng(0)
    for _ in range(100):
        n = rng.integers(2, 31); mr = rng.integers(1, 100)
        M = rng.standard_normal((n, n)); H = M @ M.T + 0.1 * np.eye(n); f = rng.standard_normal(n) * 3
        A = rng.standard_normal((mr, n)); x0 = rng.standard_normal(n); b = A @ x0 + rng.random(mr) * 0.5
        b[0] = A[0] @ x0
        A = np.vstack([A, A[:2], -A[:1], np.zeros((1, n))]); b = np.concatenate([b, b[:2], -b[:1], [0.0]])
        x, lam, ok = ko.qp_solve(H, f, A, b)
        assert ok and ko.qp_kkt_residual(H, f, A, b, x, lam) < 1e-8
    # infeasible -> NaN (quadprog_gurobi.m:22-23)
    x, lam, ok = ko.qp_solve(np.eye(2), np.zeros(2), np.array([[1.0, 0], [-1.0, 0]]), np.array([-1.0, -1.0]))
    assert not ok and np.isnan(x).all()


# ---------------------------------------------------------------------------------------------------------------
# The reference's stored MATLAB closed loops pin fit -> model extraction -> QP assembly -> quadprog end to end.
# ---------------------------------------------------------------------------------------------------------------

def _stored_run_setup(arm, mt):
    """example_sysid.m / example_control.m settings of the stored block-M runs.  The stored inputs leave the
    +-7 pi/8 box of example_control.m:20 (max |U| = 3.3 > 2.75), so those runs were made with input_bounds = []:
    slope 0.1, costs 10 / 100 / 0.1*[3e-2, 2e-2, 1e-2], horizon 10, projection on the end effector."""
    sc = arm["scale"]
    dic = ko.build_dictionary(mt, 6, 3, ["poly"], [3], arm["pairs"], dim_red=True)
    koop = ko.get_koopman(dic, arm["pairs"])
    mdl = ko.get_blmodel(dic, koop, 6) if mt == "bilinear" else ko.get_model(dic, koop, 6)
    s = ko.MpcSetup(model_type=mt, A=mdl["A"], B=mdl["B"], m=3, Np=10, projmtx=mdl["C"][-2:, :], cost_running=10.0,
                    cost_terminal=100.0, cost_input=0.1 * np.array([3e-2, 2e-2, 1e-2]), input_bounds=None,
                    slope_lim=1e-1 * sc["u_factor"].mean(), smooth_lim=None, n=6)
    return dic, s


@pytest.mark.parametrize("mt,key,max_tol", [("bilinear", "bilin", 1e-3), ("linear", "lin", 5e-3)])
def test_stored_matlab_input_sequences_are_reproduced_teacher_forced(arm, golden, mt, key, max_tol):
    """Golden vectors: res_bilin.U / res_lin.U (301x3), the input sequences MATLAB's quadprog produced in the
    reference's own closed loops (Ksim.m:205-258).  Replayed teacher-forced: at step k the controller sees the
    stored state Y(k) and previous input U(k) (Ksim.m:153-166), the reference rows k..k+Np (:198-202), and must
    return the stored U(k+1) (:225-228, :252).  This pins `\\` (A, B, B_i from the least-squares fit), get_model /
    get_BLmodel, the cost and constraint assembly and the QP optimum to MATLAB's outputs over all 299 steps.
    Tolerance: median < 1e-6 (quadprog's interior-point tolerance is 1e-8 in scaled units); the maximum is set by a
    few steps on which H is flat along the stored deviation (cost_input ~ 1e-3 against output weights 10..100):
    the next test shows that on EVERY step the stored input is optimal to within quadprog's termination tolerance."""
    r = golden["arm_blockM"]; sc = arm["scale"]
    dic, s = _stored_run_setup(arm, mt)
    ysc = (golden["blockM_ref"]["y"] - sc["y_offset"][-2:]) / sc["y_factor"][-2:]           # scaledown_ref, Kmpc.m:135-142
    Y, U = r[key + "_Y"], r[key + "_U"]
    assert np.abs(U).max() > 7 * np.pi / 8                                                 # the stored runs had no input box
    d = np.empty(299)
    for k in range(299):
        z = ko.econ_full(dic, ko.scaledown(sc, "y", Y[k])[None, :])[0]
        Uo, kkt = ko.mpc_step(s, z, ko.scaledown(sc, "u", U[k]), ysc[k:k + 11])
        assert kkt < 1e-8
        d[k] = np.abs(ko.scaleup(sc, "u", Uo[1]) - U[k + 1]).max()
    assert np.median(d) < 1e-6, np.median(d)
    assert d.max() < max_tol, (d.max(), int(d.argmax()))
    assert (d < 1e-5).sum() >= 250


@pytest.mark.parametrize("mt,key", [("bilinear", "bilin"), ("linear", "lin")])
def test_stored_matlab_inputs_are_optimal_for_the_restated_qp(arm, golden, mt, key):
    """The size-independent form of the pin: for every step, re-solve the restated QP with the applied input row
    U(2,:) pinned to MATLAB's stored value.  The optimal cost rises by < 5e-8 of its magnitude on all 299 steps
    (measured: 1.0e-9 bilinear, 3.3e-8 linear) - i.e. MATLAB's answer is an optimum of the restated problem to within
    quadprog's own stopping tolerance, including the few steps where the input itself differs by 1e-3."""
    r = golden["arm_blockM"]; sc = arm["scale"]
    dic, s = _stored_run_setup(arm, mt)
    ysc = (golden["blockM_ref"]["y"] - sc["y_offset"][-2:]) / sc["y_factor"][-2:]
    Y, U = r[key + "_Y"], r[key + "_U"]
    E = np.zeros((6, 30)); E[:3, 3:6] = np.eye(3); E[3:, 3:6] = -np.eye(3)
    worst, solved = 0.0, 0
    for k in range(299):
        z = ko.econ_full(dic, ko.scaledown(sc, "y", Y[k])[None, :])[0]
        Hq, f, Aq, bq = ko.mpc_qp(s, z, ko.scaledown(sc, "u", U[k]), ysc[k:k + 11])
        x, lam, ok = ko.qp_solve(Hq, f, Aq, bq)
        us = ko.scaledown(sc, "u", U[k + 1])
        x2, _, ok2 = ko.qp_solve(Hq, f, np.vstack([Aq, E]), np.concatenate([bq, us + 1e-9, -us + 1e-9]))
        if not ok2:                     # stored input sits a hair outside the slope rows (interior-point tolerance)
            continue
        solved += 1
        obj = lambda v: 0.5 * v @ Hq @ v + f @ v
        worst = max(worst, (obj(x2) - obj(x)) / abs(obj(x)))
    assert solved >= 295 and worst < 5e-8, (solved, worst)


def test_stored_circle_runs_under_load_pin_lift_and_inputs_on_867_more_matlab_steps(arm, golden):
    """Golden vectors: simulations/circle_c0-0p7_r0p3_15sec/bilinear_..._16-43.mat res{1..3} - three more closed loops of the
    SAME N = 34 bilinear model (the plant carries loads W = -pi/3, 0, +pi/3; the controller does not know them), each with
    Y, U, R, Z.  (a) The stored lifted states Z (300 x 34 x 3) are reproduced to 2e-14.  (b) Teacher-forced replay of the stored
    inputs: the largest stored increment is 0.0789 = 0.01 * mean(u_factor) * u_factor(1), i.e. these runs used
    input_slopeConst = 1e-2 (everything else example_control.m, no input box: run 3 leaves it); the controller's reference
    window at step k is ref_sc(k : k+Np) (Ksim.m:198-202) and results.R(j) = ref(j-1) (:252 appends ref_sc(k) AFTER the step),
    so ref = R(2:end) and the windows are complete for the first 289 steps.  Medians 3.7e-6 / 1.6e-5 / 5.7e-8, maxima
    1.6e-3 - the flat directions of H again (slope rows bind less than in the block-M runs)."""
    c = golden["arm_circle"]; sc = arm["scale"]
    dic, s = _stored_run_setup(arm, "bilinear")
    s = ko.MpcSetup(**{**s.__dict__, "slope_lim": 1e-2 * sc["u_factor"].mean()})
    med = []
    for i in range(3):
        Y, U, R, Z = (c[f"run{i}_{k}"] for k in "YURZ")
        assert np.abs(ko.econ_full(dic, ko.scaledown(sc, "y", Y[:300])) - Z).max() < 2e-14
        assert abs(np.abs(np.diff(U, axis=0)).max() - 1e-2 * sc["u_factor"].mean() * sc["u_factor"][0]) < 1e-7     # quadprog interior point: 2e-9 inside the slope row
        ref = (R[1:] - sc["y_offset"][-2:]) / sc["y_factor"][-2:]                                   # scaledown_ref, Kmpc.m:135-142
        d = np.empty(289)
        for k in range(289):
            Uo, kkt = ko.mpc_step(s, Z[k], ko.scaledown(sc, "u", U[k]), ref[k:k + 11])
            assert kkt < 1e-8
            d[k] = np.abs(ko.scaleup(sc, "u", Uo[1]) - U[k + 1]).max()
        assert np.median(d) < 3e-5 and d.max() < 3e-3 and (d < 1e-3).sum() >= 285, (i, np.median(d), d.max(), (d < 1e-3).sum())
        med.append(np.median(d))
    assert min(med) < 1e-6


def test_stored_loaded_lift_has_the_kron_layout_of_econ_full_loaded(golden):
    """Golden vectors: ..._2020-06-21_23-31.mat res_loaded{1..3}.Z (300 x 96), the only artefact of the reference that touches the
    LOADED path (its model, N = 32, nw = 2, was trained on a data set that is not shipped).  lift.econ_full_loaded(zeta, w) =
    kron(eye(nw+1), psi) * [1; w] (Ksysid.m:1606-1612), written by Ksim.m:192-194, :256 with w = scaledown.w(What): the stored
    rows must be [psi, w_1 psi, w_2 psi] with psi ending in the constant 1 - so column 64 IS w_1, column 96 IS w_2, both are one
    affine map of the stored What (scaledown.w), and the oracle's loaded_lift(psi, w) rebuilds every stored row exactly."""
    c = golden["arm_circle"]
    for i in range(3):
        Z, What = c[f"loaded{i}_Z"], c[f"loaded{i}_What"]
        psi = Z[:, :32]
        assert np.array_equal(psi[:, 31], np.ones(300))                                         # the dictionary's constant
        w = np.stack([Z[:, 63], Z[:, 95]], axis=1)                                              # w_j * (the constant)
        assert np.abs(ko.loaded_lift(psi, w) - Z).max() < 1e-15
        # scaledown.w is affine: w = (What - offset) / factor.  Step k appends its estimate to results.What BEFORE it lifts
        # (Ksim.m:190-194, then :206-216), so row k of Z goes with row k + 1 of What
        for j in range(2):
            A = np.stack([What[1:301, j], np.ones(300)], axis=1)
            if np.ptp(What[1:301, j]) > 1e-9:
                coef = np.linalg.lstsq(A, w[:, j], rcond=None)[0]
                assert np.abs(A @ coef - w[:, j]).max() < 1e-12, (i, j)
