"""GPU parity tests (through the C ABI) of the MPC step: QP assembly and QP solve.
Oracle: oracle/koopman_oracle.py — the LITERAL assembly of Kmpc.m (Bhat rebuilt from dense
matrix powers, H = B'C'QCB + R, ...) and an exact dual active-set QP solve (the reference's
quadprog is a MathWorks built-in).  Pinned to MATLAB's own outputs by the teacher-forced replay of the
stored res_bilin.U / res_lin.U sequences (last test of this file and tests/test_oracle_golden.py), and by
the uniqueness of the optimum of a strictly convex QP (KKT residual checks) everywhere else.
Tolerances (f64): QP data 1e-11 relative; U 1e-8 absolute (inputs are O(1))."""
import numpy as np
import pytest

import koopman_realizations_amd as kra
from koopman_realizations_amd import _ffi as F
from koopman_realizations_amd.device import Mpc
from oracle import koopman_oracle as ko
from test_gpu_fit import make_basis

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def arm_models(arm):
    """Bilinear and linear arm models (example_sysid.m settings) fitted by the oracle."""
    out = {}
    for mt in ("bilinear", "linear"):
        dic = ko.build_dictionary(mt, 6, 3, ["poly"], [3], arm["pairs"], dim_red=True)
        koop = ko.get_koopman(dic, arm["pairs"])
        mdl = ko.get_blmodel(dic, koop, 6) if mt == "bilinear" else ko.get_model(dic, koop, 6)
        out[mt] = (dic, mdl)
    return out


def example_control_setup(arm, mt, dic, mdl, Np=10, with_bounds=True, smooth=False):
    """example_control.m:19-28 settings, scaled as the Kmpc constructor does."""
    sc = arm["scale"]
    bounds = np.tile([-7 * np.pi / 8, 7 * np.pi / 8], (3, 1))
    lohi = np.stack([ko.scaledown(sc, "u", bounds[:, 0]), ko.scaledown(sc, "u", bounds[:, 1])], axis=1)  # Kmpc.m:659
    return ko.MpcSetup(model_type=mt, A=mdl["A"], B=mdl["B"], m=3, Np=Np, projmtx=mdl["C"][-2:, :],
                       cost_running=10.0, cost_terminal=100.0, cost_input=0.1 * np.array([3e-2, 2e-2, 1e-2]),
                       input_bounds=lohi if with_bounds else None,
                       slope_lim=1e-1 * sc["u_factor"].mean(),               # Kmpc.m:684
                       smooth_lim=(0.05 ** 2 * 0.5 * sc["u_factor"].mean()) if smooth else None, n=6)


def make_mpc(ctx, s: ko.MpcSetup):
    ci = np.asarray(s.cost_input, dtype=float)
    r = np.full(s.m, float(ci)) if ci.ndim == 0 else ci
    lo = hi = None
    if s.input_bounds is not None:
        lo, hi = s.input_bounds[:, 0], s.input_bounds[:, 1]
    return Mpc(ctx, s.model_type, s.A, s.B, s.Np, s.projmtx, s.cost_running, s.cost_terminal, r, lo, hi, s.slope_lim, s.smooth_lim)


def sample_states(arm, golden, dic, k):
    """(z, u_prev, ref horizon) taken from the stored closed-loop run (state Y(k), input U(k), ref R)."""
    r = golden["arm_blockM"]
    y = ko.scaledown(arm["scale"], "y", r["bilin_Y"][k])
    z = ko.econ_full(dic, y[None, :])[0]
    u_prev = np.clip(ko.scaledown(arm["scale"], "u", r["bilin_U"][k]), -0.9, 0.9)
    ref = golden["blockM_ref"]["y"]
    ysc = (ref - arm["scale"]["y_offset"][-2:]) / arm["scale"]["y_factor"][-2:]      # scaledown_ref, Kmpc.m:135-142
    return y, z, u_prev, ysc[k:k + 11]


@pytest.mark.parametrize("mt", ["bilinear", "linear"])
@pytest.mark.parametrize("smooth", [False, True])
def test_qp_assembly_matches_literal_kmpc(ctx, arm, golden, arm_models, mt, smooth):
    dic, mdl = arm_models[mt]
    s = example_control_setup(arm, mt, dic, mdl, smooth=smooth)
    mpc = make_mpc(ctx, s)
    assert mpc.nvar == 30 and mpc.nrows == (126 if not smooth else 126 + 48)   # 66 + 54 + 6
    for k in (0, 17, 150, 295):
        y, z, u_prev, ref = sample_states(arm, golden, dic, k)
        U, st = mpc.step(z, u_prev, ko.pad_ref(ref, s.Np))
        Hq, f, Aq, bq = mpc.last_qp()
        Hr, fr, Ar, br = ko.mpc_qp(s, z, u_prev, ref)
        assert np.abs(Hq - Hr).max() <= 1e-11 * np.abs(Hr).max()
        assert np.abs(f - fr).max() <= 1e-11 * max(1.0, np.abs(fr).max())
        assert (Aq == Ar).all() and np.abs(bq - br).max() <= 1e-15
        x, lam, ok = ko.qp_solve(Hr, fr, Ar, br)
        assert ok and st == 0
        assert ko.qp_kkt_residual(Hr, fr, Ar, br, x, lam) < 1e-8
        assert np.abs(U - x.reshape(s.Np, s.m)).max() < 1e-8
        assert np.abs(U[0] - u_prev).max() < 1e-9          # pinned first input (Kmpc.m:865-870)


@pytest.mark.parametrize("Np", [3, 5, 7, 12, 16, 21])
def test_horizon_lengths_odd_and_even_variable_counts(ctx, arm, golden, arm_models, Np):
    """nvar = 3 Np: odd counts take the per-element workgroup inverse, even ones the 2 x 2 tile version, nvar > 32 the
    multi-tile loop; consecutive steps also exercise the warm start with odd / even active-set sizes."""
    dic, mdl = arm_models["bilinear"]
    s = example_control_setup(arm, "bilinear", dic, mdl, Np=Np)
    mpc = make_mpc(ctx, s)
    assert mpc.nvar == 3 * Np
    ref_all = golden["blockM_ref"]["y"]
    ysc = (ref_all - arm["scale"]["y_offset"][-2:]) / arm["scale"]["y_factor"][-2:]
    for k in (0, 1, 2, 40, 41, 150):
        y, z, u_prev, _ = sample_states(arm, golden, dic, k)
        ref = ysc[k:k + Np + 1]
        U, st = mpc.step(z, u_prev, ko.pad_ref(ref, s.Np))
        Hr, fr, Ar, br = ko.mpc_qp(s, z, u_prev, ref)
        x, lam, ok = ko.qp_solve(Hr, fr, Ar, br)
        assert ok and st == 0
        assert np.abs(U - x.reshape(s.Np, s.m)).max() < 1e-8


def test_fused_lift_step_and_iterated_linearisation(ctx, arm, golden, arm_models):
    dic, mdl = arm_models["bilinear"]
    s = example_control_setup(arm, "bilinear", dic, mdl)
    mpc = make_mpc(ctx, s)
    b = make_basis(ctx, dic)
    for k in (3, 120):
        y, z, u_prev, ref = sample_states(arm, golden, dic, k)
        Yr = ko.pad_ref(ref, s.Np)
        U1, st1 = mpc.step(z, u_prev, Yr)
        U2, z2, st2 = mpc.step_zeta(b, y, u_prev, Yr)
        assert st1 == 0 and st2 == 0
        assert np.abs(z2 - z).max() < 1e-13 and np.abs(U1 - U2).max() < 1e-9
        for iters in (2, 3):                                   # get_mpcInput_bilinear_iter, Kmpc.m:874-899
            Ui, sti = mpc.step(z, u_prev, Yr, iters=iters)
            Uo, kkt = ko.mpc_step(s, z, u_prev, ref, iters=iters)
            assert sti == 0 and np.abs(Ui - Uo).max() < 1e-7


def test_short_reference_is_padded_like_the_reference(ctx, arm, golden, arm_models):
    dic, mdl = arm_models["bilinear"]
    s = example_control_setup(arm, "bilinear", dic, mdl)
    mpc = make_mpc(ctx, s)
    y, z, u_prev, ref = sample_states(arm, golden, dic, 297)      # tail of the 301-point trajectory: 4 rows
    assert ref.shape[0] == 4
    U, st = mpc.step(z, u_prev, ko.pad_ref(ref, s.Np))
    Uo, _ = ko.mpc_step(s, z, u_prev, ref)
    assert st == 0 and np.abs(U - Uo).max() < 1e-8


def test_batch_matches_single(ctx, arm, golden, arm_models):
    dic, mdl = arm_models["bilinear"]
    s = example_control_setup(arm, "bilinear", dic, mdl)
    mpc = make_mpc(ctx, s)
    ks = list(range(0, 290, 7))
    Z, UP, YR, Us = [], [], [], []
    for k in ks:
        y, z, u_prev, ref = sample_states(arm, golden, dic, k)
        Z.append(z); UP.append(u_prev); YR.append(ko.pad_ref(ref, s.Np))
        Us.append(mpc.step(z, u_prev, YR[-1])[0])
    Ub, st = mpc.step_batch(np.array(Z), np.array(UP), np.array(YR))
    assert (st == 0).all()
    # same kernel and the same (unique) optimum; the single-problem path warm-starts from the previous step's active
    # set, so its pivoting order - and rounding - differs from the cold-started batch
    assert np.abs(Ub - np.array(Us)).max() < 1e-11


@pytest.mark.parametrize("mt", ["linear", "bilinear"])
def test_qp_solver_is_insensitive_to_rounding_level_noise_at_degenerate_vertices(ctx, arm, golden, arm_models, mt):
    """The stored runs sit on vertices with more tight rows than variables (slope rows of saturated moves; no input box,
    as in the stored runs).  The device active-set solve of those QPs must not depend on the rounding of their assembly:
    H and f perturbed at the 1e-9 relative level (b is left alone - the pinned first input is a pair of opposite
    inequalities, Kmpc.m:865-870) still solve, to the oracle's optimum of the perturbed problem."""
    dic, mdl = arm_models[mt]
    s = example_control_setup(arm, mt, dic, mdl, with_bounds=False)
    mpc = make_mpc(ctx, s)
    rng = np.random.default_rng(7)
    degenerate = 0
    r = golden["arm_blockM"]
    key = "lin" if mt == "linear" else "bilin"
    refs = golden["blockM_ref"]["y"]
    ysc = (refs - arm["scale"]["y_offset"][-2:]) / arm["scale"]["y_factor"][-2:]
    for k in range(5, 295, 12):                      # the stored run's own states and (unclipped) previous inputs
        z = ko.econ_full(dic, ko.scaledown(arm["scale"], "y", r[key + "_Y"][k])[None, :])[0]
        u_prev = ko.scaledown(arm["scale"], "u", r[key + "_U"][k])
        U, st = mpc.step(z, u_prev, ko.pad_ref(ysc[k:k + 11], s.Np))
        assert st == 0
        H, f, A, b = mpc.last_qp()
        for rep in range(2):
            fp = f * (1 + 1e-9 * rng.standard_normal(f.shape))
            E = 1e-9 * rng.standard_normal(H.shape) * np.sqrt(np.outer(np.diag(H), np.diag(H)))
            Hp = H + 0.5 * (E + E.T)
            x, stq = ctx.qp_solve(Hp, fp, A, b)
            xo, lam, ok = ko.qp_solve(Hp, fp, A, b)
            assert ok and stq == 0 and np.abs(x - xo).max() < 1e-8, (k, rep, stq)
        degenerate += int((np.abs(A @ xo - b) < 1e-9).sum() > H.shape[0])
    if mt == "linear":
        assert degenerate >= 5                       # the case this test is about does occur


def test_infeasible_qp_returns_nan_like_the_gurobi_shim(ctx, arm, golden, arm_models):
    dic, mdl = arm_models["bilinear"]
    s = example_control_setup(arm, "bilinear", dic, mdl)
    mpc = make_mpc(ctx, s)
    y, z, u_prev, ref = sample_states(arm, golden, dic, 10)
    u_bad = np.array([5.0, 0.0, 0.0])                           # pinned input outside the box: infeasible
    U, st = mpc.step(z, u_bad, ko.pad_ref(ref, s.Np))
    assert st == F.KP_ERR_QP_FAIL and np.isnan(U).all()         # quadprog_gurobi.m:22-23, Ksim.m:220


def test_generic_qp_shim_on_random_problems(ctx):
    rng = np.random.default_rng(0)
    for trial in range(40):
        n = int(rng.integers(2, 41)); mr = int(rng.integers(1, 120))
        M = rng.standard_normal((n, n)); H = M @ M.T + 0.1 * np.eye(n); f = rng.standard_normal(n) * 3
        A = rng.standard_normal((mr, n)); x0 = rng.standard_normal(n); b = A @ x0 + rng.random(mr) * 0.5
        b[0] = A[0] @ x0
        A = np.vstack([A, A[:2], -A[:1], np.zeros((1, n))]); b = np.concatenate([b, b[:2], -b[:1], [0.0]])
        x, st = ctx.qp_solve(H, f, A, b)
        xo, lam, ok = ko.qp_solve(H, f, A, b)
        assert ok and st == 0, trial
        assert np.abs(x - xo).max() < 1e-7 * max(1.0, np.abs(xo).max()), trial
    x, st = ctx.qp_solve(np.eye(2), np.zeros(2), np.array([[1.0, 0], [-1.0, 0]]), np.array([-1.0, -1.0]))
    assert st == F.KP_ERR_QP_FAIL and np.isnan(x).all()
    x, st = ctx.qp_solve(np.eye(3), np.array([1.0, -2, 3]), np.zeros((0, 3)), np.zeros(0))   # unconstrained
    assert st == 0 and np.abs(x - np.array([-1.0, 2, -3])).max() < 1e-14


def test_closed_loop_model_as_plant_tracks_blockM(ctx, arm, golden, arm_models):
    """300-step closed loop on the blockM reference (BASELINE config 3) with the identified model
    as the plant (Kmpc.run_simulation, Kmpc.m:403-512): GPU controller vs oracle controller."""
    dic, mdl = arm_models["bilinear"]
    s = example_control_setup(arm, "bilinear", dic, mdl)
    mpc = make_mpc(ctx, s)
    ref = golden["blockM_ref"]["y"]
    ysc = (ref - arm["scale"]["y_offset"][-2:]) / arm["scale"]["y_factor"][-2:]
    y = ko.scaledown(arm["scale"], "y", golden["arm_blockM"]["bilin_Y"][0])
    z = ko.econ_full(dic, y[None, :])[0]
    u_prev = np.zeros(3); u_prev = ko.scaledown(arm["scale"], "u", np.zeros(3))
    worst = 0.0
    err = []
    for k in range(120):
        refhor = ysc[k:k + s.Np + 1]
        U, st = mpc.step(z, u_prev, ko.pad_ref(refhor, s.Np))
        assert st == 0
        if k % 10 == 0:
            Uo, kkt = ko.mpc_step(s, z, u_prev, refhor)
            worst = max(worst, np.abs(U - Uo).max())
        z = mdl["A"] @ z + ko.beta_bilinear(mdl["B"], z, 3) @ u_prev        # plant = model (one-step delay, Ksim.m:240)
        u_prev = U[1]
        err.append(np.linalg.norm(mdl["C"][-2:] @ z - ysc[min(k + 1, len(ysc) - 1)]))
    assert worst < 1e-8
    assert np.mean(err[20:]) < 0.15          # the controller tracks the reference in scaled units


def test_kmpc_ksim_mirror_example_control_flow(ctx, golden):
    """example_control.m:31-40,59,68 through the host mirrors (Ksysid -> Kmpc -> Ksim), with the
    identified model as the plant.  Result struct has the reference's fields (Ksim.m:129-138)."""
    g = golden["arm_data"]
    lens = g["train_len"]; off = np.concatenate([[0], np.cumsum(lens)])
    train = [{"t": g["train_t"][a:b], "y": g["train_y"][a:b], "u": g["train_u"][a:b]} for a, b in zip(off[:-1], off[1:])]
    val = [{"t": g["val_t"], "y": g["val_y"], "u": g["val_u"]}]
    ks = kra.Ksysid({"train": train, "val": val}, ctx=ctx, model_type="bilinear", obs_type=["poly"], obs_degree=[3],
                    snapshots=np.inf, lasso=[np.inf], delays=0, dim_red=True).train_models()
    mpc = kra.Kmpc(ks, horizon=10, input_bounds=[-7 * np.pi / 8, 7 * np.pi / 8], input_slopeConst=1e-1, input_smoothConst=None,
                   state_bounds=None, cost_running=10, cost_terminal=100, cost_input=0.1 * np.array([3e-2, 2e-2, 1e-2]),
                   projmtx=ks.model["C"][-2:, :])
    assert mpc.dev.nvar == 30 and mpc.dev.nrows == 126
    sim = kra.Ksim(kra.ModelPlant(ks), mpc)
    ref = golden["blockM_ref"]["y"]
    y0 = golden["arm_blockM"]["bilin_Y"][0]
    res = sim.run_trial_mpc(ref[:80], x0=y0, u0=np.zeros(3))
    assert set(res) == {"T", "U", "Y", "K", "R", "X", "Z", "comp_time", "err"}
    assert res["U"].shape == (80, 3) and res["Z"].shape == (79, 34) and res["comp_time"].shape == (79,)
    assert np.abs(res["U"]).max() <= 7 * np.pi / 8 + 1e-9                     # input box respected
    du = np.abs(np.diff(res["U"], axis=0))
    lim = 1e-1 * ks.params["scale"]["u_factor"].mean() * ks.params["scale"]["u_factor"]
    assert (du <= lim + 1e-8).all()                                            # slope constraint (Kmpc.m:684)
    assert res["err"][40:].mean() < 0.12                                       # tracks the block-M in output units
    # first controller call reproduces the oracle's optimum for the same state
    dic = ko.Dictionary("bilinear", 6, 3, ko.make_basis(6, ["poly"], [3]), ks.basis["pcs"])
    sc = ks.params["scale"]
    s = ko.MpcSetup("bilinear", ks.model["A"], ks.model["B"], 3, 10, ks.model["C"][-2:, :], 10.0, 100.0,
                    0.1 * np.array([3e-2, 2e-2, 1e-2]),
                    np.stack([(np.full(3, -7 * np.pi / 8) - sc["u_offset"]) / sc["u_factor"],
                              (np.full(3, 7 * np.pi / 8) - sc["u_offset"]) / sc["u_factor"]], axis=1),
                    1e-1 * sc["u_factor"].mean(), None, None, 6)
    ysc = mpc.scaledown_ref(ref)
    z0 = ko.econ_full(dic, ks.scaledown_y(y0)[None, :])[0]
    Uo, kkt = ko.mpc_step(s, z0, ks.scaledown_u(np.zeros(3)), ysc[0:11])
    assert np.abs(ks.scaleup_u(Uo[1]) - res["U"][1]).max() < 1e-7


def _example_control(ctx, golden, mt, input_bounds=(-7 * np.pi / 8, 7 * np.pi / 8)):
    g = golden["arm_data"]; gp = golden["arm_plant"]
    lens = g["train_len"]; off = np.concatenate([[0], np.cumsum(lens)])
    train = [{"t": g["train_t"][a:b], "y": g["train_y"][a:b], "u": g["train_u"][a:b]} for a, b in zip(off[:-1], off[1:])]
    val = [{"t": g["val_t"], "y": g["val_y"], "u": g["val_u"]}]
    ks = kra.Ksysid({"train": train, "val": val}, ctx=ctx, model_type=mt, obs_type=["poly"], obs_degree=[3],
                    snapshots=np.inf, lasso=[np.inf], delays=0, dim_red=True).train_models()
    mpc = kra.Kmpc(ks, horizon=10, input_bounds=list(input_bounds), input_slopeConst=1e-1, input_smoothConst=None,
                   state_bounds=None, cost_running=10, cost_terminal=100, cost_input=0.1 * np.array([3e-2, 2e-2, 1e-2]),
                   projmtx=ks.model["C"][-2:, :])
    params = {k[2:]: (float(gp[k]) if gp[k].ndim == 0 else gp[k]) for k in gp.files if k.startswith("p_")}
    return ks, mpc, kra.Ksim(kra.Arm(params, output_type="markers"), mpc)


def test_example_control_true_arm_closed_loop_bilinear(ctx, golden):
    """example_control.m:15-69 end to end: fit on the arm data, build the bilinear controller, close
    the loop around the TRUE arm plant for the 15 s block-M.  The stored run of the reference
    (res_bilin) tracked with mean error 0.0203; the loop here must do as well, inside the same
    constraints."""
    ks, mpc, sim = _example_control(ctx, golden, "bilinear")
    res = sim.run_trial_mpc(golden["blockM_ref"]["y"], None, None)
    assert res["Y"].shape == (301, 6) and res["X"].shape == (301, 6) and res["Z"].shape == (300, 34)
    stored = float(golden["arm_plant"]["bilin_err"].mean())
    assert 0.7 * stored < res["err"].mean() < 1.1 * stored
    assert np.abs(res["U"]).max() <= 7 * np.pi / 8 + 1e-9
    du = np.abs(np.diff(res["U"], axis=0))
    assert (du <= 1e-1 * ks.params["scale"]["u_factor"].mean() * ks.params["scale"]["u_factor"] + 1e-8).all()
    # the end effector ends within 1 cm of the stored run's end point
    assert np.abs(res["Y"][-1, -2:] - golden["arm_blockM"]["bilin_Y"][-1, -2:]).max() < 1e-2


def test_free_running_closed_loop_reproduces_the_stored_matlab_run(ctx, golden):
    """The whole chain with nothing teacher-forced: fit on the device, Kmpc on the device, the arm plant integrated by
    the ode45 restatement, 300 closed-loop steps (Ksim.m:167-270).  With the constraint set the stored sequences were
    generated under (slope constraint only, see test_oracle_golden.py) the free-running loop stays on the stored MATLAB
    trajectory `res_bilin`: outputs to 1e-4, inputs to 2e-3 (of a +-2.7 range; differences of quadprog's 1e-8-level
    optimum are fed back through the plant 300 times), mean tracking error to 6 digits.  The linear loop is a chaotic
    limit cycle (stored mean error 0.74): it is compared over its first 100 steps only."""
    ks, mpc, sim = _example_control(ctx, golden, "bilinear", input_bounds=())
    res = sim.run_trial_mpc(golden["blockM_ref"]["y"], None, None)
    st = golden["arm_blockM"]
    assert np.abs(res["Y"] - st["bilin_Y"][:301]).max() < 1e-4
    assert np.abs(res["U"][:300] - st["bilin_U"][:300]).max() < 2e-3
    stored = float(golden["arm_plant"]["bilin_err"].mean())
    assert abs(res["err"].mean() - stored) < 1e-6 * stored * 10
    ks, mpc, sim = _example_control(ctx, golden, "linear", input_bounds=())
    res = sim.run_trial_mpc(golden["blockM_ref"]["y"][:112], None, None)
    assert np.abs(res["Y"][:100] - st["lin_Y"][:100]).max() < 1e-2
    assert np.abs(res["U"][:100] - st["lin_U"][:100]).max() < 5e-2


def test_example_control_linear_first_input_matches_stored_run(ctx, golden):
    """Same flow with the linear realization: the first controller output equals the stored U(2,:)
    of res_lin (same data, same least-squares model, same QP) to solver precision."""
    ks, mpc, sim = _example_control(ctx, golden, "linear")
    res = sim.run_trial_mpc(golden["blockM_ref"]["y"][:12], None, None)       # first call sees rows 1..11, unpadded
    assert np.abs(res["U"][1] - golden["arm_blockM"]["lin_U"][1]).max() < 1e-6
    assert np.abs(res["Y"][1] - golden["arm_blockM"]["lin_Y"][1]).max() < 1e-12


@pytest.mark.parametrize("mt", ["linear", "bilinear"])
@pytest.mark.parametrize("N,n,Np", [(3, 2, 8), (5, 2, 6), (4, 3, 5)])
def test_state_bounds_match_literal_kmpc(ctx, mt, N, n, Np):
    """State bounds (Kmpc.m:300-318 / :716-730) with the reference's placement of the kron block (first (Np+1) n columns
    of the stacked lifted state, :306): dense, z-dependent rows assembled on the device against the oracle's literal
    E, F, c -> L = F + E Bhat, M = E Ahat and the exact QP optimum."""
    rng = np.random.default_rng(100 * N + Np)
    m = 2
    A = 0.9 * np.linalg.qr(rng.standard_normal((N, N)))[0] + 0.02 * rng.standard_normal((N, N))
    B = 0.5 * rng.standard_normal((N, m)) if mt == "linear" else 0.5 * rng.standard_normal((N, N * m))
    proj = np.hstack([np.eye(1), np.zeros((1, N - 1))])
    lohi = np.stack([np.full(n, -0.55), np.full(n, 0.6)], axis=1)
    s = ko.MpcSetup(model_type=mt, A=A, B=B, m=m, Np=Np, projmtx=proj, cost_running=1.0, cost_terminal=5.0, cost_input=np.array([0.05, 0.02]),
                    input_bounds=np.tile([-1.0, 1.0], (m, 1)), slope_lim=0.4, smooth_lim=None, state_bounds=lohi, n=n)
    mpc = make_mpc(ctx, s)
    mpc.set_state_bounds(lohi[:, 0], lohi[:, 1])
    hit = 0
    for trial in range(10):
        z = 0.15 * rng.standard_normal(N); z[-1] = 0.5
        u_prev = 0.2 * rng.standard_normal(m)
        ref = np.full((Np + 1, 1), 0.8 * (-1) ** trial)
        U, st = mpc.step(z, u_prev, ko.pad_ref(ref, Np))
        Hr, fr, Ar, br = ko.mpc_qp(s, z, u_prev, ref)
        x, lam, ok = ko.qp_solve(Hr, fr, Ar, br)
        if not ok:
            assert st != 0 and np.isnan(U).all()                      # infeasible state bounds: NaN like the Gurobi shim
            continue
        assert st == 0
        assert np.abs(U - x.reshape(Np, m)).max() < 1e-8
        # do the state-bound rows matter for this problem?
        s0 = ko.MpcSetup(**{**s.__dict__, "state_bounds": None})
        H0, f0, A0, b0 = ko.mpc_qp(s0, z, u_prev, ref)
        x0, _, ok0 = ko.qp_solve(H0, f0, A0, b0)
        hit += ok0 and np.abs(x0 - x).max() > 1e-6
    if (N, n, Np) == (3, 2, 8):
        assert hit >= 1                                                 # here the bounded entries reach step 5: the bounds do bind
    # the same problems as ONE batched launch sequence (state bounds in kp_mpc_step_batch), and with two linearisation
    # passes (get_mpcInput_bilinear_iter, Kmpc.m:874-899: H, f follow the predicted lifted states, the constraint matrix
    # stays the one of the current state, :861)
    rng2 = np.random.default_rng(7 * N + Np)
    nbp = 6
    Zb = 0.15 * rng2.standard_normal((nbp, N)); Zb[:, -1] = 0.5
    Ub = 0.2 * rng2.standard_normal((nbp, m))
    refs = [np.full((Np + 1, 1), 0.8 * (-1) ** t_) for t_ in range(nbp)]
    Yb = np.stack([ko.pad_ref(r_, Np) for r_ in refs])
    Uall, stall = mpc.step_batch(Zb, Ub, Yb)
    for p_ in range(nbp):
        x, lam, ok = ko.qp_solve(*ko.mpc_qp(s, Zb[p_], Ub[p_], refs[p_]))
        if not ok:
            assert stall[p_] != 0 and np.isnan(Uall[p_]).all()
        else:
            assert stall[p_] == 0 and np.abs(Uall[p_] - x.reshape(Np, m)).max() < 1e-8
        if mt == "bilinear":
            U2, st2 = mpc.step(Zb[p_], Ub[p_], Yb[p_], iters=2)
            Uo, kkt = ko.mpc_step(s, Zb[p_], Ub[p_], refs[p_], iters=2)
            if np.isnan(Uo).any():
                assert st2 != 0 and np.isnan(U2).all()
            else:
                assert st2 == 0 and np.abs(U2 - Uo).max() < 1e-8
    mpc.set_state_bounds(None, None)
    U, st = mpc.step(z, u_prev, ko.pad_ref(ref, Np))
    x0, _, ok0 = ko.qp_solve(*ko.mpc_qp(ko.MpcSetup(**{**s.__dict__, "state_bounds": None}), z, u_prev, ref))
    assert ok0 and np.abs(U - x0.reshape(Np, m)).max() < 1e-8


@pytest.mark.parametrize("mt,key,max_tol", [("bilinear", "bilin", 1e-3), ("linear", "lin", 5e-3)])
def test_stored_matlab_input_sequences_replayed_teacher_forced_on_device(ctx, golden, mt, key, max_tol):
    """The reference's stored closed loops (res_bilin.U / res_lin.U, produced by MATLAB's `\\` and quadprog) replayed
    through the product path: device fit (Ksysid mirror: fused Gram + Cholesky + refinement) -> get_model / get_BLmodel
    -> kp_mpc_create -> one kp_mpc_step_zeta per stored state.  At step k the controller sees Y(k), U(k) and the
    reference rows k..k+Np (Ksim.m:153-166, :198-202) and must return the stored U(k+1) (:225-228).  The stored inputs
    leave example_control.m's +-7 pi/8 box, so those runs had input_bounds = []; everything else is example_control.m.
    Same tolerances as the oracle's replay (tests/test_oracle_golden.py): median < 1e-6, max set by the steps on which
    the QP is flat along the deviation; and the device must agree with the oracle's exact optimum to 1e-7 throughout."""
    g = golden["arm_data"]
    lens = g["train_len"]; off = np.concatenate([[0], np.cumsum(lens)])
    train = [{"t": g["train_t"][a:b], "y": g["train_y"][a:b], "u": g["train_u"][a:b]} for a, b in zip(off[:-1], off[1:])]
    val = [{"t": g["val_t"], "y": g["val_y"], "u": g["val_u"]}]
    ks = kra.Ksysid({"train": train, "val": val}, ctx=ctx, model_type=mt, obs_type=["poly"], obs_degree=[3],
                    snapshots=np.inf, lasso=[np.inf], delays=0, dim_red=True).train_models()
    mpc = kra.Kmpc(ks, horizon=10, input_bounds=[], input_slopeConst=1e-1, input_smoothConst=None, state_bounds=[],
                   cost_running=10, cost_terminal=100, cost_input=0.1 * np.array([3e-2, 2e-2, 1e-2]),
                   projmtx=ks.model["C"][-2:, :])
    r = golden["arm_blockM"]
    Y, U = r[key + "_Y"], r[key + "_U"]
    ref_sc = mpc.scaledown_ref(golden["blockM_ref"]["y"])
    sc = ks.params["scale"]
    s = ko.MpcSetup(model_type=mt, A=ks.model["A"], B=ks.model["B"], m=3, Np=10, projmtx=ks.model["C"][-2:, :], cost_running=10.0,
                    cost_terminal=100.0, cost_input=0.1 * np.array([3e-2, 2e-2, 1e-2]), input_bounds=None,
                    slope_lim=1e-1 * sc["u_factor"].mean(), smooth_lim=None, n=6)
    d = np.empty(299); dz = 0.0; dor = 0.0
    step = mpc.get_mpcInput if mt == "linear" else (lambda c, rh: mpc.get_mpcInput_bilinear_iter(c, rh, 1))
    for k in range(299):
        cur = {"y": ks.scaledown_y(Y[k])[None, :], "u": ks.scaledown_u(U[k])[None, :]}
        Uk, z = step(cur, ref_sc[k:k + 11])
        assert not np.isnan(Uk).any()
        d[k] = np.abs(ks.scaleup_u(Uk[1]) - U[k + 1]).max()
        dz = max(dz, np.abs(z - r[key + "_Z"][k]).max())                       # stored lifted state (Ksim.m:256)
        if k % 10 == 0:
            Uo, _ = ko.mpc_step(s, z, cur["u"][0], ref_sc[k:k + 11])
            dor = max(dor, np.abs(Uo - Uk).max())
    assert dz < 1e-11, dz
    assert dor < 1e-7, dor
    assert np.median(d) < 1e-6, np.median(d)
    assert d.max() < max_tol, (d.max(), int(d.argmax()))
    assert (d < 1e-5).sum() >= 245


def test_stored_circle_runs_replayed_on_device(ctx, golden):
    """The three stored circle closed loops (tests/test_oracle_golden.py::test_stored_circle_runs_... has the derivation of
    their settings: input_slopeConst = 1e-2, reference = R(2:end)) through the product path: device fit -> KmpcHip-shaped
    controller -> one kp_mpc_step_zeta per stored state, 3 x 289 steps.  Stored Z to 1e-11, device optimum = oracle optimum
    to 1e-7, stored MATLAB inputs with the oracle replay's tolerances."""
    g = golden["arm_data"]; c = golden["arm_circle"]
    lens = g["train_len"]; off = np.concatenate([[0], np.cumsum(lens)])
    train = [{"t": g["train_t"][a:b], "y": g["train_y"][a:b], "u": g["train_u"][a:b]} for a, b in zip(off[:-1], off[1:])]
    val = [{"t": g["val_t"], "y": g["val_y"], "u": g["val_u"]}]
    ks = kra.Ksysid({"train": train, "val": val}, ctx=ctx, model_type="bilinear", obs_type=["poly"], obs_degree=[3],
                    snapshots=np.inf, lasso=[np.inf], delays=0, dim_red=True).train_models()
    mpc = kra.Kmpc(ks, horizon=10, input_bounds=[], input_slopeConst=1e-2, input_smoothConst=None, state_bounds=[],
                   cost_running=10, cost_terminal=100, cost_input=0.1 * np.array([3e-2, 2e-2, 1e-2]), projmtx=ks.model["C"][-2:, :])
    sc = ks.params["scale"]
    s = ko.MpcSetup(model_type="bilinear", A=ks.model["A"], B=ks.model["B"], m=3, Np=10, projmtx=ks.model["C"][-2:, :], cost_running=10.0,
                    cost_terminal=100.0, cost_input=0.1 * np.array([3e-2, 2e-2, 1e-2]), input_bounds=None,
                    slope_lim=1e-2 * sc["u_factor"].mean(), smooth_lim=None, n=6)
    for i in range(3):
        Y, U, R, Z = (c[f"run{i}_{k}"] for k in "YURZ")
        ref_sc = mpc.scaledown_ref(R[1:])
        d = np.empty(289); dz = 0.0; dor = 0.0
        for k in range(289):
            cur = {"y": ks.scaledown_y(Y[k])[None, :], "u": ks.scaledown_u(U[k])[None, :]}
            Uk, z = mpc.get_mpcInput_bilinear_iter(cur, ref_sc[k:k + 11], 1)
            assert not np.isnan(Uk).any()
            d[k] = np.abs(ks.scaleup_u(Uk[1]) - U[k + 1]).max()
            dz = max(dz, np.abs(z - Z[k]).max())
            if k % 10 == 0:
                Uo, _ = ko.mpc_step(s, z, cur["u"][0], ref_sc[k:k + 11])
                dor = max(dor, np.abs(Uo - Uk).max())
        assert dz < 1e-11 and dor < 1e-7, (i, dz, dor)
        assert np.median(d) < 3e-5 and d.max() < 3e-3 and (d < 1e-3).sum() >= 285, (i, np.median(d), d.max(), (d < 1e-3).sum())


def test_stored_loaded_lift_rows_through_the_device_kernel(ctx, golden):
    """res_loaded{1..3}.Z (300 x 96, MATLAB's lift.econ_full_loaded) through the product's loaded lift: a loaded dictionary is the
    unloaded one declared bilinear with the load as pseudo-input, its row [psi, w_1 psi, w_2 psi].  The model that
    produced these rows is not shipped, so psi is fed through the identity dictionary [zeta; 1] on zeta = the stored psi(1:31) -
    the device kernel's Kronecker row layout is then checked against every stored row, exactly."""
    c = golden["arm_circle"]
    b = kra.Basis(ctx, "bilinear", 31, 2, [])                       # psi(zeta) = [zeta; 1]: N = 32; pseudo-input = the two loads
    assert b.N == 32 and b.W == 96
    for i in range(3):
        Z = c[f"loaded{i}_Z"]
        w = np.stack([Z[:, 63], Z[:, 95]], axis=1)
        rows = b.lift(F.LIFT_ROW, Z[:, :31], w)
        assert np.abs(rows - Z).max() < 1e-15, i
    b.close()


@pytest.mark.parametrize("mt,N,m,Np,nproj", [("bilinear", 7, 1, 5, 3), ("bilinear", 12, 2, 7, 5), ("linear", 9, 3, 4, 4), ("bilinear", 40, 2, 6, 1),
                                             ("bilinear", 120, 3, 10, 2), ("bilinear", 200, 3, 10, 2), ("linear", 30, 1, 9, 2)])
def test_assembly_and_solve_on_random_models_of_other_shapes(ctx, mt, N, m, Np, nproj):
    """Shapes the arm models do not have: 1 - 5 tracked outputs (the Hessian assembly is specialised for nproj <= 4 and generic
    beyond), one input (odd variable counts: the bordered pair sweep of the workgroup inverse), models whose constant blocks
    P | P_k B_i fit the step kernel's LDS staging (N = 120) and models where they do not (N = 200), linear models (no P_k B_i).
    QP data against the literal assembly of Kmpc.m:861-883, the optimum against the exact active-set solve, single steps
    (warm-started from the step before, staged constants) against the batched launch (cold, constants from L2)."""
    rng = np.random.default_rng(1000 * N + 10 * Np + nproj)
    A = 0.9 * np.linalg.qr(rng.standard_normal((N, N)))[0] + 0.02 * rng.standard_normal((N, N))
    B = 0.5 * rng.standard_normal((N, m)) if mt == "linear" else 0.5 / np.sqrt(N) * rng.standard_normal((N, N * m))
    proj = rng.standard_normal((nproj, N)) / np.sqrt(N)
    s = ko.MpcSetup(model_type=mt, A=A, B=B, m=m, Np=Np, projmtx=proj, cost_running=2.0, cost_terminal=7.0,
                    cost_input=0.05 + 0.01 * np.arange(m), input_bounds=np.tile([-0.6, 0.7], (m, 1)), slope_lim=0.25, smooth_lim=None, n=min(N, 3))
    mpc = make_mpc(ctx, s)
    Z, UP, YR, Us = [], [], [], []
    for trial in range(6):
        z = 0.3 * rng.standard_normal(N); z[-1] = 1.0
        u_prev = rng.uniform(-0.5, 0.5, m)
        ref = 0.6 * rng.standard_normal((Np + 1, nproj))
        U, st = mpc.step(z, u_prev, ko.pad_ref(ref, Np))
        Hd, fd, Ad, bd = mpc.last_qp()
        Hr, fr, Ar, br = ko.mpc_qp(s, z, u_prev, ref)
        assert np.abs(Hd - Hr).max() <= 1e-11 * np.abs(Hr).max() and np.abs(fd - fr).max() <= 1e-11 * max(1.0, np.abs(fr).max())
        assert np.array_equal(Hd, Hd.T)                                # (assembled from the upper triangle)
        assert np.abs(Ad - Ar).max() == 0 and np.abs(bd - br).max() <= 1e-14
        x, lam, ok = ko.qp_solve(Hr, fr, Ar, br)
        assert ok and st == 0 and np.abs(U - x.reshape(Np, m)).max() < 1e-8
        Z.append(z); UP.append(u_prev); YR.append(ko.pad_ref(ref, Np)); Us.append(U)
    Ub, stb = mpc.step_batch(np.array(Z), np.array(UP), np.array(YR))
    assert (stb == 0).all() and np.abs(Ub - np.array(Us)).max() < 1e-9
