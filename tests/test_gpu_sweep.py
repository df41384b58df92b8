"""GPU test of the random-system sweep unit (evaluate_rand_models.m:47-143) on shipped
rand-systems data: every model type/degree of one system vs the oracle."""
import numpy as np
import pytest

import koopman_realizations_amd as kra
from koopman_realizations_amd import sweep
from oracle import koopman_oracle as ko

pytestmark = pytest.mark.gpu


def _system(golden, i):
    g = golden["rand_systems"]
    t, y, u = g[f"s{i}_train_t"], g[f"s{i}_train_y"], g[f"s{i}_train_u"]
    n = t.shape[0] // 1001
    train = [{"t": t[k * 1001:(k + 1) * 1001], "y": y[k * 1001:(k + 1) * 1001], "u": u[k * 1001:(k + 1) * 1001]} for k in range(n)]
    val = [{"t": g[f"s{i}_val_t"], "y": g[f"s{i}_val_y"], "u": g[f"s{i}_val_u"]}]
    return {"train": train, "val": val}


def _oracle_system(d, degrees):
    merged = ko.merge_trials(d["train"])
    sd, sc = ko.get_scale(merged)
    pairs = ko.snapshot_pairs(sd, 0)
    val = {"t": d["val"][0]["t"], "y": ko.scaledown(sc, "y", d["val"][0]["y"]), "u": ko.scaledown(sc, "u", d["val"][0]["u"])}
    out = {}
    for mt in ("linear", "bilinear", "nonlinear"):
        errs = []
        for j in range(1, degrees[mt] + 1):
            dic = ko.build_dictionary(mt, 1, 1, ["poly"], [j])
            koop = ko.get_koopman(dic, pairs)                       # lasso 4 is inactive (checked below)
            if mt == "nonlinear":
                assert np.abs(koop["K"]).sum() <= 4 * dic.N         # t = lasso*N (Ksysid.m:996): QP == least squares
            mdl = {"linear": ko.get_model, "bilinear": ko.get_blmodel, "nonlinear": ko.get_nlmodel}[mt](dic, koop, 1)
            r = ko.val_model(dic, mdl, val, 0)
            e = ko.get_error(r["sim_y"], r["real_y"])
            errs.append(float((e["mean"] / (np.abs(r["real_y"]).sum(axis=0) / r["real_y"].shape[0]))[0]))
        out[mt] = np.array(errs)
    return out


def test_eval_system_matches_oracle(ctx, golden):
    degrees = {"linear": 13, "bilinear": 6, "nonlinear": 4}
    d = _system(golden, 0)
    got = sweep.eval_system(d, ctx=ctx, degrees=degrees)
    want = _oracle_system(d, degrees)
    for mt in want:
        g, w = got[mt][0], want[mt]
        assert g.shape == w.shape
        # normal equations vs QR: cond up to ~3e5 at linear degree 13 => 1e-4 relative
        assert np.all(np.abs(g - w) <= 1e-4 * np.maximum(1.0, np.abs(w))), (mt, g, w)
    assert list(got["linear"][1]) == [j + 1 for j in range(1, 14)]          # size(basis.full,1) = N
    assert list(got["bilinear"][1]) == [2 * (j + 1) for j in range(1, 7)]    # size(basis.full_input,1)


def test_rand_sweep_table_single_rank(ctx, golden):
    systems = [_system(golden, i) for i in range(3)]
    tab = sweep.rand_models_sweep(systems, ctx=ctx, degrees={"linear": 3, "bilinear": 2, "nonlinear": 2})
    assert tab["linear"].shape == (3, 3) and tab["bilinear"].shape == (2, 3) and tab["nonlinear"].shape == (2, 3)
    # unstable models blow up to NaN in the reference too; the statistics drop them (evaluate_rand_models.m:149-171)
    mean, std = sweep.sweep_statistics(tab["linear"])
    assert np.isfinite(mean).all() and np.isfinite(tab["linear"]).all()
    assert sweep.sweep_percentiles(tab["linear"]).shape == (3, 5)


def test_generated_systems_flow_through_the_sweep(ctx):
    """Rsys -> data4sysid -> evaluate_rand_models body: data from the restated generator trains and
    validates like the shipped data sets (errors finite for the linear family and non-increasing on
    average from degree 1 to 3)."""
    from koopman_realizations_amd.rsys import Rsys
    r = Rsys(3, 3, 2, 2, seed=11)
    rng = np.random.default_rng(0)
    sets = Rsys.save_data(r.simulate_systems(2.0, 0.01, 4, rng.uniform(-1, 1, (4, 1))))
    tab = sweep.rand_models_sweep(sets, ctx=ctx, degrees={"linear": 3, "bilinear": 2, "nonlinear": 1})
    assert tab["linear"].shape == (3, 3) and np.isfinite(tab["linear"]).all() and np.isfinite(tab["bilinear"]).all()
    mean, _ = sweep.sweep_statistics(tab["linear"])
    assert mean[2] <= mean[0] * 1.05


def test_batched_sweep_matches_per_system_sweep(ctx, golden):
    """kp_fit_batch (one workgroup per system) + batched rollouts give the same error table as the per-system
    loop of evaluate_rand_models.m through the Ksysid mirror."""
    systems = [_system(golden, i) for i in range(3)]
    degrees = {"linear": 13, "bilinear": 6, "nonlinear": 4}
    want = sweep.rand_models_sweep(systems, ctx=ctx, degrees=degrees)
    got = sweep.rand_models_sweep_batched(systems, ctx, degrees=degrees)
    via_shard = sweep.rand_models_sweep(systems, ctx=ctx, degrees=degrees, batched=True)
    for mt in want:
        assert np.array_equal(np.nan_to_num(via_shard[mt], nan=-1.0), np.nan_to_num(got[mt], nan=-1.0))
        assert got[mt].shape == want[mt].shape
        w, g = want[mt], got[mt]
        both_nan = np.isnan(w) & np.isnan(g)
        # diverged rollouts (dropped by the statistics, evaluate_rand_models.m:155-157): BOTH sides must say so - above 10 on both,
        # or overflowed (beyond 1e6 / not finite) on one and not finite on the other; an error above 10 on one side only fails
        with np.errstate(invalid="ignore"):
            big = ((np.abs(w) > 10) & (np.abs(g) > 10)) | (((np.abs(w) > 1e6) | ~np.isfinite(w)) & ~np.isfinite(g))
        ok = both_nan | big | (np.abs(g - w) <= 1e-4 * np.maximum(1.0, np.abs(w)))
        assert ok.all(), (mt, g, w)


def test_device_resident_sweep_on_generated_systems(ctx):
    """64 DISTINCT generated systems (Rsys restatement, the data set's shape: 10 + 1 trials of 1001 samples) through
    the device-resident path (kp_traj_upload + one kp_sweep_eval per model type and degree: scaling, snapshot pairs, fit,
    model extraction, validation rollout and error on the device) against (a) the host-prepared batched path and (b) the
    numpy oracle's per-system evaluation.  Also pins the device's scaling and K for one dictionary."""
    from koopman_realizations_amd.rsys import Rsys
    from koopman_realizations_amd.device import Traj, Basis
    r = Rsys(64, 3, 3, 2, seed=21)
    systems = Rsys.save_data(r.simulate_systems_fast(10.0, 0.01, 11, np.zeros((1, 1))))
    degrees = {"linear": 13, "bilinear": 6, "nonlinear": 4}
    got = sweep.rand_models_sweep_batched(systems, ctx, degrees=degrees)                 # nested: one data pass per model type
    flat = sweep.rand_models_sweep_batched(systems, ctx, degrees=degrees, nested=False)  # one kp_sweep_eval per degree
    host = sweep._sweep_batched_host(systems, ctx, degrees)

    def close(g, w, tol):
        both_nan = np.isnan(w) & np.isnan(g)
        # diverged rollouts (dropped by the statistics, evaluate_rand_models.m:155-157): BOTH sides must say so - above 10 on both,
        # or overflowed (beyond 1e6 / not finite) on one and not finite on the other; an error above 10 on one side only fails
        with np.errstate(invalid="ignore"):
            big = ((np.abs(w) > 10) & (np.abs(g) > 10)) | (((np.abs(w) > 1e6) | ~np.isfinite(w)) & ~np.isfinite(g))
        # errors above 1 (the model does worse than predicting zero) come from unstable rollouts that amplify the last
        # digits of K over 1000 steps: compared at 1e-3 relative
        unstable = (np.abs(w) > 1) & (np.abs(g - w) <= 1e-3 * np.abs(w))
        return both_nan | big | unstable | (np.abs(g - w) <= tol * np.maximum(1.0, np.abs(w)))
    for mt in degrees:
        assert got[mt].shape == (degrees[mt], 64)
        for name, tab in (("nested", got[mt]), ("flat", flat[mt])):
            ok = close(tab, host[mt], 1e-7)
            assert ok.all(), (mt, name, np.argwhere(~ok)[:5], tab[~ok][:5], host[mt][~ok][:5])
    # the oracle's table of ALL 64 systems (scale -> pairs -> SVD lstsq -> model -> rollout -> error; 1.6 s per system, so it
    # is stored: tests/golden/sweep_oracle_seed21.npz, made by tests/golden/make_sweep_oracle.py).  Three systems are
    # recomputed here first: a stored table that no longer belongs to these generated systems would show.
    import os
    stored = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "sweep_oracle_seed21.npz"))
    for i in (0, 17, 63):
        want = _oracle_system(systems[i], degrees)
        for mt in degrees:
            assert np.allclose(stored[mt][:, i], want[mt], rtol=1e-6, atol=1e-9, equal_nan=True), (i, mt)
    for mt in degrees:
        # normal equations in the Chebyshev internal basis vs SVD lstsq at cond ~1e5: 1e-4; no escape for |err| > 1 beyond
        # the relative 1e-3 of `close` (unstable rollouts amplify the last digits of K over 1000 steps)
        ok = close(got[mt], stored[mt], 1e-4)
        assert ok.all(), (mt, np.argwhere(~ok)[:5], got[mt][~ok][:5], stored[mt][~ok][:5])
        assert np.isfinite(got[mt]).mean() > 0.9
    # scaling and K of one dictionary against the oracle
    raw = sweep._stack_raw(systems)
    traj = Traj(ctx, raw[0], raw[1], raw[2], raw[3], raw[4])
    sc = traj.scale()
    merged = ko.merge_trials(systems[5]["train"])
    sd, so = ko.get_scale(merged)
    assert np.allclose(sc[5], [so["y_offset"][0], so["y_factor"][0], so["u_offset"][0], so["u_factor"][0]], rtol=0, atol=1e-15)
    basis = Basis(ctx, "bilinear", 1, 1, [("poly", kra.poly_exponent_table(1, 4)[1:])], None)
    err, K, st = traj.sweep_eval(basis, want_K=True)
    pairs = ko.snapshot_pairs(sd, 0)
    assert pairs["alpha"].shape[0] == 10 * 1000 - 1                         # the last good pair is dropped (Ksysid.m:960)
    dic = ko.build_dictionary("bilinear", 1, 1, ["poly"], [4])
    Kref = ko.get_koopman(dic, pairs)["K"]
    assert np.abs(K[5] - Kref).max() <= 1e-8 * np.abs(Kref).max()
    basis.close()
    # K of every degree of the nested pass (Chebyshev internal basis, mapped back) against the SVD least squares: the
    # degree-13 dictionary has cond(Px) ~ 1e5, plain normal equations would be off by 1e-8 .. 1e-7 here
    b13 = Basis(ctx, "linear", 1, 1, [("poly", kra.poly_exponent_table(1, 13)[1:])], None)
    err13, st13 = traj.sweep_eval_nested(b13, 13)
    for dj in (0, 5, 12):
        dic = ko.build_dictionary("linear", 1, 1, ["poly"], [dj + 1])
        Kd = traj.nested_K(dj, dic.W)
        Kref = ko.get_koopman(dic, pairs)["K"]
        assert np.abs(Kd[5] - Kref).max() <= 2e-10 * np.abs(Kref).max(), (dj, np.abs(Kd[5] - Kref).max() / np.abs(Kref).max())
    bn = Basis(ctx, "nonlinear", 1, 1, [("poly", kra.poly_exponent_table(2, 4)[2:])], None)
    traj.sweep_eval_nested(bn, 4, 4.0)
    for dj in (1, 3):
        dic = ko.build_dictionary("nonlinear", 1, 1, ["poly"], [dj + 1])
        Kref = ko.get_koopman(dic, pairs)["K"]
        assert np.abs(traj.nested_K(dj, dic.W)[5] - Kref).max() <= 1e-10 * np.abs(Kref).max()
    b13.close(); bn.close()
    traj.close()


def test_gather_into_page_locked_host_arrays_gives_the_same_sweep(ctx):
    """sweep._stack_raw with a context writes the stacked trials into the context's page-locked arrays (kp_host_alloc),
    reused and regrown from call to call; blocks and error table equal those of the plain numpy gather."""
    from koopman_realizations_amd import rsys
    r = rsys.Rsys(12, 3, 3, 2, seed=5)
    systems = rsys.Rsys.save_data(r.simulate_systems_fast(2.0, 0.01, 5, np.zeros((1, 1))))
    plain = sweep._stack_raw(systems)
    for n in (4, 12, 7):                                    # grow, then reuse a larger block for a smaller gather
        pinned = sweep._stack_raw(systems[:n], ctx)
        ref = sweep._stack_raw(systems[:n])
        for a, b in zip(pinned, ref):
            assert np.array_equal(a, b)
    h = ctx.host_array("sweep_Y", (3, 5))
    h[:] = 1.0
    assert ctx.host_array("sweep_Y", (3, 5)).sum() == 15.0  # same block again
    tab_p = sweep.rand_models_sweep_arrays(*sweep._stack_raw(systems, ctx), ctx=ctx, degrees={"linear": 3, "bilinear": 2, "nonlinear": 2})
    tab_n = sweep.rand_models_sweep_arrays(*plain, ctx=ctx, degrees={"linear": 3, "bilinear": 2, "nonlinear": 2})
    for mt in tab_n:
        assert np.array_equal(tab_p[mt], tab_n[mt])


def test_fit_batch_matches_single_fits(ctx, golden):
    """K, G, C of kp_fit_batch against kp_fit_gram / kp_fit_solve system by system; singular systems are flagged."""
    from conftest import synth_pairs
    nb, Ns = 5, 777
    parts = [synth_pairs(Ns, 2, 1, seed=20 + i) for i in range(nb)]
    parts[3]["alpha"][:] = 0.25                                            # constant state: rank-deficient dictionary
    alpha = np.vstack([p["alpha"] for p in parts]); beta = np.vstack([p["beta"] for p in parts]); u = np.vstack([p["u"] for p in parts])
    for mt, deg in (("linear", 3), ("bilinear", 2), ("nonlinear", 2)):
        nv = 3 if mt == "nonlinear" else 2
        basis = kra.Basis(ctx, mt, 2, 1, [("poly", kra.poly_exponent_table(nv, deg)[nv:])], None)
        assert basis.W <= 16
        snaps = kra.Snapshots(ctx, alpha, beta, u)
        K, G, C, st = ctx.fit_batch(basis, snaps, nb)
        assert st[3] != 0 and np.isnan(K[3]).all() and (st[[0, 1, 2, 4]] == 0).all()
        for s in (0, 1, 2, 4):
            one = kra.Snapshots(ctx, parts[s]["alpha"], parts[s]["beta"], parts[s]["u"])
            G1, C1 = kra.fit_gram(ctx, basis, one)
            assert np.abs(G[s] - G1).max() <= 1e-12 * np.abs(G1).max() and np.abs(C[s] - C1).max() <= 1e-12 * np.abs(G1).max()
            K1 = ctx.fit_solve(G1, C1)
            assert np.abs(K[s] - K1).max() <= 1e-9 * max(1.0, np.abs(K1).max())


def test_model_project_batch_matches_single_and_oracle(ctx):
    """kp_model_project_batch (M-projection of get_model for many small linear models) against kp_model_project and
    the oracle's literal get_model (Ksysid.m:1206-1225) on the same snapshot pairs."""
    from conftest import synth_pairs
    nb, Ns, n, m = 6, 600, 2, 1
    parts = [synth_pairs(Ns, n, m, seed=40 + i) for i in range(nb)]
    alpha = np.vstack([p["alpha"] for p in parts]); beta = np.vstack([p["beta"] for p in parts]); u = np.vstack([p["u"] for p in parts])
    for deg in (1, 2, 3):
        basis = kra.Basis(ctx, "linear", n, m, [("poly", kra.poly_exponent_table(n, deg)[n:])], None)
        N = basis.N
        snaps = kra.Snapshots(ctx, alpha, beta, u)
        K, G, C, st = ctx.fit_batch(basis, snaps, nb)
        assert (st == 0).all()
        A, B, st2 = ctx.model_project_batch(K, G, C, N, m)
        assert (st2 == 0).all()
        dic = ko.build_dictionary("linear", n, m, ["poly"], [deg])
        for s in range(nb):
            A1, B1, M1 = ctx.model_project(K[s], G[s], C[s], N, m)
            assert np.abs(A[s] - A1).max() < 1e-9 and np.abs(B[s] - B1).max() < 1e-9
            koop = ko.get_koopman(dic, parts[s])
            om = ko.get_model(dic, koop, n)
            assert np.abs(A[s] - om["A"]).max() < 1e-7 * max(1.0, np.abs(om["A"]).max())
            assert np.abs(B[s] - om["B"]).max() < 1e-7 * max(1.0, np.abs(om["B"]).max())


def test_fit_batch_refinement_reaches_qr_accuracy_on_ill_conditioned_dictionaries(ctx):
    """Degree-13 monomials of a 1-D state (the sweep's largest linear dictionary): cond(Px) ~ 4e4, so the plain normal
    equations are good to ~1e-7 only; kp_fit_batch adds one refinement step with the residual Px'(Py - Px K) taken from
    the data and matches the QR / SVD least-squares solution (what MATLAB's `\\` returns, Ksysid.m:1069)."""
    rng = np.random.default_rng(0)
    nb, Ns = 4, 9000
    a = rng.uniform(-1, 1, (nb * Ns, 1)); u = rng.uniform(-1, 1, (nb * Ns, 1))
    b = np.clip(0.9 * a + 0.1 * np.sin(3 * a) + 0.05 * u, -1, 1)
    basis = kra.Basis(ctx, "linear", 1, 1, [("poly", kra.poly_exponent_table(1, 13)[1:])], None)
    snaps = kra.Snapshots(ctx, a, b, u)
    K, G, C, st = ctx.fit_batch(basis, snaps, nb)
    assert (st == 0).all()
    dic = ko.build_dictionary("linear", 1, 1, ["poly"], [13])
    for s in range(nb):
        pr = {"alpha": a[s * Ns:(s + 1) * Ns], "beta": b[s * Ns:(s + 1) * Ns], "u": u[s * Ns:(s + 1) * Ns]}
        Px, Py = ko.px_py(dic, pr)
        assert np.linalg.cond(Px) > 1e4
        Kq = np.linalg.lstsq(Px, Py, rcond=None)[0]
        assert np.abs(K[s] - Kq).max() <= 1e-9 * max(1.0, np.abs(Kq).max())


_RCCL_SCRIPT = r"""
import sys, faulthandler, numpy as np
faulthandler.dump_traceback_later(330, exit=True)       # a hang says WHERE before the parent's timeout fires
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, sys.argv[1] + "/tests")
import koopman_realizations_amd as kra
from koopman_realizations_amd import comm as kc, _ffi as F
from oracle import koopman_oracle as ko
from conftest import synth_pairs
c2 = kra.Context(0)
uid = kc.unique_id()
assert len(uid) == 128 and any(uid)
try:
    comm = kc.RcclComm(c2, 0, 1, uid, timeout=100.0)
except kc.RcclInitTimeout as e:                 # the box's RCCL bootstrap never completes: nothing of ours to test
    print("RCCL_INIT_TIMEOUT", e, flush=True)
    import os
    os._exit(0)
assert kc.all_gather_array(comm, np.arange(5.0)).tolist() == [[0.0, 1.0, 2.0, 3.0, 4.0]]
assert kc.all_gather_object(comm, {"k": [1, 2]}) == [{"k": [1, 2]}]
assert comm.all_reduce_sum(np.array([1.5, -2.0])).tolist() == [1.5, -2.0]
comm.barrier()
p = synth_pairs(4000, 3, 2, seed=4)
dic = ko.build_dictionary("bilinear", 3, 2, ["poly"], [2])
b = kra.Basis(c2, "bilinear", 3, 2, [("poly", kra.poly_exponent_table(3, 2)[3:])])
s = kra.Snapshots(c2, p["alpha"], p["beta"], p["u"])
Kref = kra.fit(c2, b, s)[0]
for _ in range(3):
    kra.fit(c2, b, s, fetch=False)
Kall = comm.all_gather_fit(2, b.W)
assert Kall.shape == (1, b.W, b.W) and np.abs(Kall[0] - Kref).max() <= 1e-11 * np.abs(Kref).max()
Ksh = kra.fit_sharded(c2, b, s)[0]                      # kp_fit_sharded: Gram kernel, all-reduce of [G | C], solve
assert np.abs(Ksh - Kref).max() <= 1e-12 * np.abs(Kref).max()
# the K stack of a shard through ONE ncclAllGather from the device result buffer (kp_comm_allgather_fits): the lasso grid's gather
from koopman_realizations_amd import sweep
l1 = np.abs(Kref).sum()
las3 = [np.inf, 0.5 * l1 / b.N, 0.1 * l1 / b.N]
ref3 = kra.fit(c2, b, s, las3)
got3 = sweep.lasso_sweep_device(c2, lambda ls: kra.fit(c2, b, s, ls, fetch=False), las3, b.W, comm)
assert len(got3) == 3 and all(np.array_equal(x, y) for x, y in zip(ref3, got3))
part = comm.all_gather_fits(1, 2, b.W)
assert part.shape == (1, 2, b.W, b.W) and np.array_equal(part[0, 1].T, ref3[2])
G, Cm = kra.fit_gram_sharded(c2, b, s)
Px, Py = ko.px_py(dic, p)
assert np.abs(G - Px.T @ Px).max() <= 1e-11 * np.abs(G).max()
comm.close()
# the context still works after the communicator is gone
assert np.abs(kra.fit(c2, b, s)[0] - Kref).max() == 0
c2.close()
print("RCCL_OK")
"""


def test_rccl_communicator_through_the_c_abi_single_rank():
    """kp_comm_* with a one-rank RCCL communicator: unique id, ncclCommInitRank, host all-gather / all-reduce, the
    device-to-device gather of a fit result and the sharded fit (all-reduce of [G | C] between Gram kernel and solve).
    The N-rank path is the same code with world > 1 (one process per GPU; the driver's scaling run).  Runs in a fresh
    torch-free interpreter, as bench.py's ranks do."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-c", _RCCL_SCRIPT, root], capture_output=True, text=True, timeout=400)
    if "RCCL_INIT_TIMEOUT" in r.stdout:
        pytest.skip("ncclCommInitRank of ONE rank does not return on this box (watchdog): " + r.stdout.strip()[-200:])
    assert r.returncode == 0 and "RCCL_OK" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]


_TWO_RANK_SCRIPT = r"""
import sys, os, faulthandler
faulthandler.dump_traceback_later(280, exit=True)
sys.path.insert(0, sys.argv[1])
import numpy as np
import koopman_realizations_amd as kra
from koopman_realizations_amd import comm as kc, sweep
ctx, comm = kc.init_from_env(kra.Context)
assert comm.world == 2
parts = kc.all_gather_array(comm, np.full(3, 1.0 + comm.rank))
assert parts.shape == (2, 3) and parts[1, 0] == 2.0
Ks = sweep.lasso_sweep(lambda l: np.full((2, 2), l), [0.1, 0.2, 0.3], comm, shape=(2, 2))
assert [k[0, 0] for k in Ks] == [0.1, 0.2, 0.3]
comm.barrier()
if comm.rank == 0:
    print("KIND", comm.kind, "|", getattr(comm, "fallback_reason", ""))
"""


def test_two_ranks_on_one_gpu_agree_on_the_file_fallback():
    """Two ranks forced onto one device: RCCL refuses the communicator (`invalid usage`), every rank votes through the
    rendezvous directory and all of them continue on the file backend - no rank is left blocked in a collective, and the
    result says which backend ran.  (With one GPU per rank the same launch keeps the RCCL communicator.)"""
    import os
    import subprocess
    import sys
    from koopman_realizations_amd import comm as kc
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, KP_FORCE_DEVICE="0")
    env.pop("KP_COMM_BACKEND", None)
    out = kc.spawn_ranks([sys.executable, "-c", _TWO_RANK_SCRIPT, root], 2, env=env, timeout=300)
    assert "KIND file | " in out and "ncclCommInitRank" in out, out


def test_a_system_whose_fit_fails_shows_as_nan_in_the_sweep_table(ctx):
    """evaluate_rand_models.m propagates whatever the fit returns; a system whose Gram matrix is singular (a stuck sensor:
    constant output) has no model, and the batched sweep reports NaN for it - not the rollout of an unsolved system."""
    from koopman_realizations_amd.rsys import Rsys
    r = Rsys(4, 3, 3, 2, seed=5)
    systems = Rsys.save_data(r.simulate_systems_fast(10.0, 0.01, 11, np.zeros((1, 1))))
    for tr in systems[2]["train"]:
        tr["y"] = np.zeros_like(tr["y"]); tr["u"] = np.zeros_like(tr["u"])
    systems[2]["val"][0]["y"] = np.zeros_like(systems[2]["val"][0]["y"])
    degrees = {"linear": 3, "bilinear": 2, "nonlinear": 2}
    tab = sweep.rand_models_sweep_batched(systems, ctx, degrees=degrees)
    for mt in degrees:
        assert np.isnan(tab[mt][:, 2]).all(), (mt, tab[mt][:, 2])
        assert np.isfinite(tab[mt][:, [0, 1, 3]]).all()


def test_nested_sweep_with_two_states_and_two_inputs(ctx):
    """The Gram pass of the nested sweep beyond the 1-D systems of evaluate_rand_models.m: 2 states and 2 inputs, so the
    dictionaries have 2 variables (4 for the nonlinear model: [zeta; u]), monomials with two factors, two input columns
    (linear) or two input blocks (bilinear).  K of every degree against the oracle's least squares on the same pairs."""
    from koopman_realizations_amd.device import Traj, Basis
    rng = np.random.default_rng(12)
    nb, k, T, n, m = 3, 3, 220, 2, 2
    systems = []
    for s_ in range(nb):
        A = np.array([[0.9, 0.1], [-0.1, 0.85]]) + 0.02 * rng.standard_normal((2, 2))
        B = 0.1 * rng.standard_normal((2, 2))
        trials = []
        for _ in range(k + 1):
            u = rng.uniform(-1, 1, (T, m)); y = np.zeros((T, n)); y[0] = rng.uniform(-0.5, 0.5, n)
            for t in range(T - 1):
                y[t + 1] = A @ y[t] + B @ u[t] + 0.05 * np.tanh(y[t] * u[t, ::-1])
            trials.append({"t": np.arange(T) * 0.01, "y": y, "u": u})
        systems.append({"train": trials[:k], "val": trials[k:]})
    raw = sweep._stack_raw(systems)
    assert raw is not None
    traj = Traj(ctx, *raw)
    for mt, D in (("linear", 3), ("bilinear", 1), ("nonlinear", 2)):
        nv = n + (m if mt == "nonlinear" else 0)
        basis = Basis(ctx, mt, n, m, [("poly", kra.poly_exponent_table(nv, D)[nv:])], None)
        assert basis.W <= 16
        err, st = traj.sweep_eval_nested(basis, D, np.inf)
        assert (st == 0).all() and np.isfinite(err).all()
        for i in range(nb):
            sd, _ = ko.get_scale(ko.merge_trials(systems[i]["train"]))
            pairs = ko.snapshot_pairs(sd, 0)
            for dj in range(D):
                dic = ko.build_dictionary(mt, n, m, ["poly"], [dj + 1])
                Kref = ko.get_koopman(dic, pairs)["K"]
                Kd = traj.nested_K(dj, dic.W)[i]
                assert np.abs(Kd - Kref).max() <= 1e-9 * np.abs(Kref).max(), (mt, i, dj, np.abs(Kd - Kref).max() / np.abs(Kref).max())
        basis.close()
    traj.close()


def test_nested_sweep_on_short_trials(ctx):
    """Tile edges of the sweep's Gram pass (128 pairs per tile, raw values fetched four tiles ahead): 131 pairs (a second tile
    with three of them), 257 pairs (two full tiles and one pair - 2 trials of 130 rows), and a system with ONE pair, whose fit is
    singular: status set, NaN errors, no crash."""
    from koopman_realizations_amd.device import Traj, Basis
    rng = np.random.default_rng(3)
    for (k, T) in ((3, 45), (2, 130)):
        systems = []
        for _ in range(2):
            trials = []
            for _t in range(k + 1):
                u = rng.uniform(-1, 1, (T, 1)); y = np.zeros((T, 1)); y[0] = rng.uniform(-0.5, 0.5)
                for t in range(T - 1):
                    y[t + 1] = 0.9 * y[t] + 0.2 * u[t] - 0.1 * y[t] ** 3
                trials.append({"t": np.arange(T) * 0.01, "y": y, "u": u})
            systems.append({"train": trials[:k], "val": trials[k:]})
        traj = Traj(ctx, *sweep._stack_raw(systems))
        basis = Basis(ctx, "linear", 1, 1, [("poly", kra.poly_exponent_table(1, 3)[1:])], None)
        err, st = traj.sweep_eval_nested(basis, 3, np.inf)
        assert (st == 0).all()
        for i in range(2):
            sd, _ = ko.get_scale(ko.merge_trials(systems[i]["train"]))
            pairs = ko.snapshot_pairs(sd, 0)
            assert pairs["alpha"].shape[0] == k * (T - 1) - 1
            for dj in range(3):
                dic = ko.build_dictionary("linear", 1, 1, ["poly"], [dj + 1])
                Kref = ko.get_koopman(dic, pairs)["K"]
                assert np.abs(traj.nested_K(dj, dic.W)[i] - Kref).max() <= 1e-9 * np.abs(Kref).max()
        basis.close(); traj.close()
    Y = rng.uniform(-1, 1, (2, 3, 1)); U = rng.uniform(-1, 1, (2, 3, 1))
    traj = Traj(ctx, Y, U, 1, rng.uniform(-1, 1, (2, 20, 1)), rng.uniform(-1, 1, (2, 20, 1)))
    basis = Basis(ctx, "linear", 1, 1, [("poly", kra.poly_exponent_table(1, 2)[1:])], None)
    err, st = traj.sweep_eval_nested(basis, 2, np.inf)
    assert (st != 0).all() and np.isnan(err).all()
    basis.close(); traj.close()


def test_trajectory_object_in_three_steps_equals_the_one_call_upload(ctx):
    """kp_traj_create / kp_traj_put / kp_traj_finish (each block on its way while the caller prepares the next) against
    kp_traj_upload: same scaling, same sweep table; an object that is not finished refuses to be used, and finish refuses an
    object with a block missing."""
    from koopman_realizations_amd.device import Traj, Basis
    from koopman_realizations_amd.rsys import Rsys
    r = Rsys(6, 3, 3, 2, seed=9)
    systems = Rsys.save_data(r.simulate_systems_fast(4.0, 0.01, 5, np.zeros((1, 1))))
    Y, U, k, Yv, Uv = sweep._stack_raw(systems)
    one = Traj(ctx, Y, U, k, Yv, Uv)
    three = Traj.begin(ctx, Y.shape[0], k, Y.shape[1] // k, 1, 1, Yv.shape[1])
    basis = Basis(ctx, "linear", 1, 1, [("poly", kra.poly_exponent_table(1, 3)[1:])], None)
    three.put("Y", Y); three.put("U", U)
    with pytest.raises(kra.KoopmanHipError):
        three.sweep_eval_nested(basis, 3, np.inf)                      # not finished
    with pytest.raises(kra.KoopmanHipError):
        three.finish()                                                 # Yv, Uv missing
    with pytest.raises(ValueError):
        three.put("Yv", Yv[:, :-1])                                    # wrong shape
    three.put("Yv", Yv); three.put("Uv", Uv)
    three.finish()
    assert np.array_equal(one.scale(), three.scale())
    e1, s1 = one.sweep_eval_nested(basis, 3, np.inf)
    e3, s3 = three.sweep_eval_nested(basis, 3, np.inf)
    assert np.array_equal(e1, e3, equal_nan=True) and np.array_equal(s1, s3)
    basis.close(); one.close(); three.close()
    # ... and the batched sweep (which uploads block by block) gives the table of the blocks uploaded in one call
    degrees = {"linear": 3, "bilinear": 2, "nonlinear": 2}
    a = sweep.rand_models_sweep_batched(systems, ctx, degrees=degrees)
    b = sweep.rand_models_sweep_arrays(Y, U, k, Yv, Uv, ctx=ctx, degrees=degrees)
    assert all(np.array_equal(a[mt], b[mt], equal_nan=True) for mt in degrees)


def test_config5_at_its_full_size_through_property_checks(ctx):
    """BASELINE configs[4] at its shape - evaluate_rand_models.m on 1024 DISTINCT generated systems (the population bench.py
    times), 23 fits + validation rollouts each - through size-independent properties: the nested form (one data pass per model
    type) equals the flat form (one kp_sweep_eval per degree) on every system; the same stacked trials dealt over the workers of a
    kp_multi object (the device listed three times: ragged chunks of 342 / 341 / 341 systems) give the single context's tables
    bit for bit; no status word is set; and the statistics of evaluate_rand_models.m:149-171 are finite."""
    import bench
    from koopman_realizations_amd.multi import Multi
    chunks = bench.gen_rand_systems(list(range(1024 // bench.RAND_CHUNK)))
    systems = [s_ for c in sorted(chunks) for s_ in chunks[c]]
    assert len(systems) == 1024
    degrees = dict(sweep.MAX_DEGREE)
    nested = sweep.rand_models_sweep_batched(systems, ctx, degrees=degrees)
    flat = sweep.rand_models_sweep_batched(systems, ctx, degrees=degrees, nested=False)
    for mt in nested:
        assert nested[mt].shape == (degrees[mt], 1024)
        ok = np.isfinite(flat[mt]) & np.isfinite(nested[mt])
        assert ok.mean() > 0.99
        # (the flat form solves the normal equations in the monomial basis, the nested one in the Chebyshev basis: cond up to 3e5 at
        # linear degree 13)
        assert np.all(np.abs(nested[mt][ok] - flat[mt][ok]) <= 1e-3 * np.maximum(1.0, np.abs(flat[mt][ok]))), mt
    mean, _ = sweep.sweep_statistics(nested["linear"])
    assert np.isfinite(mean).all()
    Y, U, k, Yv, Uv = sweep._stack_raw(systems)
    mg = Multi([0, 0, 0])
    try:
        tr = mg.traj_upload(Y, U, k, Yv, Uv)
        for mt, D in degrees.items():
            nv = 1 + (mt == "nonlinear")
            e = kra.poly_exponent_table(nv, D)[nv:]
            err, st = tr.sweep_eval_nested((mt, 1, 1, [("poly", e)], None), D, 4.0 if mt == "nonlinear" else np.inf)
            assert (st == 0).all()
            assert np.array_equal(err[:, :, 0], nested[mt], equal_nan=True), mt
        tr.close()
    finally:
        mg.close()
