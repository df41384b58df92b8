"""Loaded systems (Ksysid `loaded` = true, Ksysid.m:539-626 and the loaded branches of get_Koopman / get_model /
val_model, Kmpc.m:1298-1445): oracle restatement against the literal kron formulas (CPU), and the host mirror +
device kernels against the oracle (GPU)."""
import numpy as np
import pytest

from oracle import koopman_oracle as ko
from tests._loaded_system import make_trials


def merged(trials):
    d = {k: np.vstack([t[k].reshape(len(t["t"]), -1) for t in trials]) for k in ("t", "y", "u", "w")}
    d["t"] = d["t"].ravel()
    return d


def test_loaded_lift_is_the_kron_of_the_reference():
    rng = np.random.default_rng(1)
    psi = rng.standard_normal(7); w = rng.standard_normal(2); u = rng.standard_normal(3)
    full_loaded = np.kron(np.eye(3), psi[:, None]) @ np.concatenate([[1.0], w])          # Ksysid.m:1609-1611
    assert np.array_equal(ko.loaded_lift(psi[None, :], w[None, :])[0], full_loaded)
    with_input = np.kron(np.eye(4), full_loaded[:, None]) @ np.concatenate([[1.0], u])   # :1587-1590
    dic = ko.build_dictionary("bilinear", 2, 3, ["poly"], [2])
    zeta = rng.standard_normal(2)
    row = ko.lift_rows_loaded(dic, zeta, u, w)[0]
    psi2 = ko.econ_full(dic, zeta[None, :])[0]
    fl = np.kron(np.eye(3), psi2[:, None]) @ np.concatenate([[1.0], w])
    assert np.allclose(row, np.kron(np.eye(4), fl[:, None]) @ np.concatenate([[1.0], u]), rtol=0, atol=1e-15)
    assert with_input.shape == (7 * 3 * 4,)


@pytest.mark.parametrize("mt", ["linear", "bilinear", "nonlinear"])
def test_oracle_loaded_fit_shapes_and_rollout_consistency(mt):
    trials = make_trials(6, 120, nw=2)
    pairs = ko.snapshot_pairs_loaded(merged(trials), 0)
    assert pairs["w"].shape == (pairs["alpha"].shape[0], 2)
    dic = ko.build_dictionary(mt, 2, 1, ["poly"], [2])
    koop = ko.get_koopman_loaded(dic, pairs)
    NL = dic.N * 3
    W = {"linear": NL + 1, "bilinear": NL * 2, "nonlinear": NL}[mt]     # Ksysid.m:1019-1028
    assert koop["K"].shape == (W, W) and koop["Px"].shape[1] == NL
    mdl = {"linear": ko.get_model_loaded, "bilinear": ko.get_blmodel_loaded, "nonlinear": ko.get_nlmodel_loaded}[mt](dic, koop, 2)
    res = ko.val_model_loaded(dic, mdl, trials[0], 0)
    # one step of the rollout by hand (Ksysid.m:1667-1668 / :1761-1762 / :1858)
    z0 = trials[0]["y"][0]; u0 = trials[0]["u"][0]; w0 = trials[0]["w"][0]
    if mt == "nonlinear":
        y1 = mdl["Kf"] @ ko.loaded_lift(ko.econ_full(dic, np.concatenate([z0, u0])[None, :]), w0[None, :])[0]
    else:
        znow = ko.loaded_lift(ko.econ_full(dic, z0[None, :]), w0[None, :])[0]
        z1 = mdl["A"] @ znow + (ko.beta_bilinear(mdl["B"], znow, 1) @ u0 if mt == "bilinear" else mdl["B"] @ u0)
        y1 = z1[:2]
    assert np.allclose(res["sim_y"][1], y1[:2], rtol=0, atol=1e-13)


def test_oracle_load_estimators_recover_the_load():
    trials = make_trials(10, 200, nw=1, seed=3)
    pairs = ko.snapshot_pairs_loaded(merged(trials), 0)
    for mt, est in (("linear", ko.estimate_load_linear), ("bilinear", ko.estimate_load_bilinear)):
        dic = ko.build_dictionary(mt, 2, 1, ["poly"], [3])
        koop = ko.get_koopman_loaded(dic, pairs)
        mdl = (ko.get_model_loaded if mt == "linear" else ko.get_blmodel_loaded)(dic, koop, 2)
        for tr in trials[:4]:
            what, resnorm = est(dic, mdl, tr["y"][:40], tr["u"][:40], 1)
            assert abs(what[0] - tr["w"][0, 0]) < 0.15 and resnorm < 1e-2
            what2, _ = est(dic, mdl, tr["y"][:40], tr["u"][:40], 1, whatpast=np.array([0.0]))
            assert abs(what2[0]) <= 0.01 + 1e-9                              # slope constraint Kmpc.m:1344-1347


# ---- device ------------------------------------------------------------------------------------------------------

@pytest.fixture(scope="module")
def kra():
    import koopman_realizations_amd as k
    return k


@pytest.mark.gpu
@pytest.mark.parametrize("mt,deg,nw", [("linear", 3, 1), ("bilinear", 2, 2), ("bilinear", 3, 1), ("nonlinear", 2, 1), ("linear", 2, 2)])
def test_gpu_loaded_fit_model_and_validation_match_oracle(kra, mt, deg, nw):
    trials = make_trials(8, 150, nw=nw, seed=5)
    ks = kra.Ksysid({"train": trials[:6], "val": trials[6:]}, model_type=mt, obs_type=["poly"], obs_degree=[deg], loaded=True)
    ks.train_models()
    p = ks.params
    assert p["nw"] == nw
    # the oracle on the same scaled pairs
    sp = ks.snapshotPairs
    dic = ko.build_dictionary(mt, 2, 1, ["poly"], [deg])
    assert dic.N == p["N"]
    koop = ko.get_koopman_loaded(dic, sp)
    K = ks.koopData["K"]
    assert K.shape == koop["K"].shape
    assert np.abs(K - koop["K"]).max() <= 2e-8 * max(1.0, np.abs(koop["K"]).max())
    assert np.abs(ks.koopData["Px"] - koop["Px"]).max() < 1e-13 and np.abs(ks.koopData["Py"] - koop["Py"]).max() < 1e-13
    omdl = {"linear": ko.get_model_loaded, "bilinear": ko.get_blmodel_loaded, "nonlinear": ko.get_nlmodel_loaded}[mt](dic, koop, 2)
    for key in (("A", "B") if mt != "nonlinear" else ("Kf",)):
        assert np.abs(ks.model[key] - omdl[key]).max() <= 5e-7 * max(1.0, np.abs(omdl[key]).max()), key
    val = {"linear": ks.val_model, "bilinear": ks.val_BLmodel, "nonlinear": ks.val_NLmodel}[mt]
    for v in ks.valdata:
        res = val(ks.model, v)
        ores = ko.val_model_loaded(dic, ks.model, v, 0, mt)               # same model: isolates the rollout
        assert np.abs(res["sim"]["y"] - ores["sim_y"]).max() < 1e-9
    # a trial whose load changes half way: the rollout is cut into runs of constant load
    v = dict(ks.valdata[0]); w2 = v["w"].copy(); w2[70:] = -0.5 * w2[70:]; v["w"] = w2
    res = val(ks.model, v); ores = ko.val_model_loaded(dic, ks.model, v, 0, mt)
    assert np.abs(res["sim"]["y"] - ores["sim_y"]).max() < 1e-9


@pytest.mark.gpu
@pytest.mark.parametrize("mt,types,degs,dim_red,nw", [
    ("bilinear", ["poly"], [5], True, 1), ("linear", ["poly"], [5], True, 2), ("nonlinear", ["poly"], [4], True, 1),
    ("linear", ["poly"], [3], True, 1),                # 8 of 9 principal axes + zeta: rank deficient by construction
    ("bilinear", ["fourier"], [2], False, 1), ("linear", ["poly", "gaussian"], [2, 4], False, 1),
    ("nonlinear", ["hermite"], [2], False, 2), ("linear", ["fourier_sparser"], [2], False, 1)])
def test_gpu_loaded_any_dictionary_and_dim_red(kra, mt, types, degs, dim_red, nw):
    """Loaded systems with every kind of observable and with dim_red (Ksysid.m:539-626 + :1521-1565, :1580-1612): lift
    rows, K, the model matrices and the validation rollout against the oracle's literal kron construction."""
    trials = make_trials(8, 150, nw=nw, seed=11)
    nv = 2 + (1 if mt == "nonlinear" else 0)
    cen = [np.random.default_rng(3).uniform(-1, 1, (nv, d)) for t, d in zip(types, degs) if t == "gaussian"] or None
    ks = kra.Ksysid({"train": trials[:6], "val": trials[6:]}, gaussian_centres=cen, model_type=mt, obs_type=types, obs_degree=degs,
                    loaded=True, dim_red=dim_red)
    ks.train_models()
    sp = ks.snapshotPairs
    dic = ko.Dictionary(mt, 2, 1, ko.make_basis(nv, types, degs, cen), ks.basis["pcs"] if dim_red else None)
    assert dic.N == ks.params["N"]
    koop = ko.get_koopman_loaded(dic, sp)
    assert np.abs(ks.koopData["Px"] - koop["Px"]).max() < 1e-12 and np.abs(ks.koopData["Py"] - koop["Py"]).max() < 1e-12
    K = ks.koopData["K"]
    assert K.shape == koop["K"].shape
    Pxw = ko.lift_rows_loaded(dic, sp["alpha"], sp["u"], sp["w"]); Pyw = ko.lift_rows_loaded(dic, sp["beta"], sp["u"], sp["w"])
    sv = np.linalg.svd(Pxw, compute_uv=False)
    if sv[-1] > 1e-10 * sv[0]:
        assert np.abs(K - koop["K"]).max() <= max(2e-8, 50 * (sv[0] / sv[-1]) ** 2 * 2.2e-16) * max(1.0, np.abs(koop["K"]).max())
    else:                                              # `\` on a rank-deficient Px: a basic solution with the least-squares residual
        keep = np.abs(K).sum(axis=1) > 0               # (directions below sqrt(W 64 eps) of the largest are dropped in Gram space)
        assert 0 < keep.sum() < K.shape[0]
        r_dev = np.linalg.norm(Pxw @ K - Pyw)
        r_sub = np.linalg.norm(Pxw[:, keep] @ np.linalg.lstsq(Pxw[:, keep], Pyw, rcond=None)[0] - Pyw)
        assert r_dev <= r_sub * (1 + 1e-6) + 1e-9
    omdl = {"linear": ko.get_model_loaded, "bilinear": ko.get_blmodel_loaded, "nonlinear": ko.get_nlmodel_loaded}[mt](dic, dict(koop, K=K), 2)
    for key in (("A", "B") if mt != "nonlinear" else ("Kf",)):
        assert np.abs(ks.model[key] - omdl[key]).max() <= 5e-7 * max(1.0, np.abs(omdl[key]).max()), key
    val = {"linear": ks.val_model, "bilinear": ks.val_BLmodel, "nonlinear": ks.val_NLmodel}[mt]
    v = dict(ks.valdata[0]); w2 = v["w"].copy(); w2[70:] = -0.5 * w2[70:]; v["w"] = w2
    res = val(ks.model, v); ores = ko.val_model_loaded(dic, ks.model, v, 0, mt)
    ok = np.isfinite(ores["sim_y"]).all(axis=1) & (np.abs(ores["sim_y"]).max(axis=1) < 1e3)
    assert ok[:20].all() and np.abs(res["sim"]["y"][ok] - ores["sim_y"][ok]).max() < 1e-8
    # single-row lifts through the handles Kmpc uses
    z = ks.lift.econ_full_loaded(np.concatenate([sp["alpha"][3], sp["u"][3]]) if mt == "nonlinear" else sp["alpha"][3], sp["w"][3])
    assert np.abs(z - koop["Px"][3, :dic.N * (nw + 1)]).max() < 1e-12


@pytest.mark.gpu
@pytest.mark.parametrize("mt", ["linear", "bilinear"])
def test_gpu_loaded_lasso_matches_oracle(kra, mt):
    """solve_KoopmanQP on the loaded rows (Ksysid.m:1068-1081): t = lasso * N with the UNLOADED N (:996)."""
    trials = make_trials(8, 120, nw=1, seed=13)
    ks = kra.Ksysid({"train": trials[:6], "val": trials[6:]}, model_type=mt, obs_type=["poly"], obs_degree=[2], loaded=True, lasso=0.5)
    ks.train_models()
    dic = ko.build_dictionary(mt, 2, 1, ["poly"], [2])
    koop = ko.get_koopman_loaded(dic, ks.snapshotPairs, lasso=0.5, obj_lasso=0.5)
    K = ks.koopData["K"]
    assert abs(np.abs(K).sum() - 0.5 * dic.N) < 1e-6 * dic.N                      # the L1 budget is active
    assert np.abs(K - koop["K"]).max() < 1e-5 * max(1.0, np.abs(koop["K"]).max())


@pytest.mark.gpu
@pytest.mark.parametrize("mt", ["linear", "bilinear"])
def test_gpu_loaded_mpc_step_and_load_estimators(kra, mt):
    nw = 2 if mt == "linear" else 1
    trials = make_trials(10, 150, nw=nw, seed=7)
    ks = kra.Ksysid({"train": trials[:8], "val": trials[8:]}, model_type=mt, obs_type=["poly"], obs_degree=[2], loaded=True)
    ks.train_models()
    mpc = kra.Kmpc(ks, horizon=8, input_bounds=[-1.0, 1.0], input_slopeConst=0.5, cost_running=1.0, cost_terminal=10.0, cost_input=0.01,
                   projmtx=ks.model["C"][:1])
    dic = ko.build_dictionary(mt, 2, 1, ["poly"], [2])
    v = ks.valdata[0]
    # load estimators against the oracle's (same model, same window)
    est, oest = (mpc.estimate_load_linear, ko.estimate_load_linear) if mt == "linear" else (mpc.estimate_load_bilinear, ko.estimate_load_bilinear)
    for wp in (None, np.zeros(nw)):
        what, rn = est(v["y"][:30], v["u"][:30], wp)
        owhat, orn = oest(dic, ks.model, v["y"][:30], v["u"][:30], nw, 0, wp)
        assert np.abs(what - owhat).max() < 1e-8 and abs(rn - orn) < 1e-10
    if mt == "linear":
        assert what[-1] == 0.0                                              # the DEBUG pin of Kmpc.m:1350
    # one MPC step with the loaded lift against the oracle QP on the same lifted state
    traj = {"y": v["y"][10:11], "u": v["u"][10:11], "what": v["w"][10:11]}
    ref = np.full((9, 1), 0.2)
    U, z = (mpc.get_mpcInput if mt == "linear" else mpc.get_mpcInput_bilinear)(traj, ref)
    zo = ko.loaded_lift(ko.econ_full(dic, v["y"][10][None, :]), v["w"][10][None, :])[0]
    assert np.abs(z - zo).max() < 1e-13
    sc = ks.params["scale"]
    s = ko.MpcSetup(model_type=mt, A=ks.model["A"], B=ks.model["B"], m=1, Np=8, projmtx=ks.model["C"][:1], cost_running=1.0, cost_terminal=10.0,
                    cost_input=np.array([0.01]),
                    input_bounds=np.stack([(np.array([-1.0]) - sc["u_offset"]) / sc["u_factor"], (np.array([1.0]) - sc["u_offset"]) / sc["u_factor"]], axis=1),
                    slope_lim=0.5 * float(np.mean(sc["u_factor"])), smooth_lim=None, n=2)
    Hr, fr, Ar, br = ko.mpc_qp(s, zo, v["u"][10], ref)
    x, lam, ok = ko.qp_solve(Hr, fr, Ar, br)
    assert ok and np.abs(U.reshape(-1) - x).max() < 1e-8
