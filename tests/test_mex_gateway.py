"""GPU tests that EXECUTE the MATLAB gateway: matlab/kp_mex.c, compiled unchanged against the functional mex.h stand-in
(tests/mex_shim), is driven command by command through its mexFunction and every result is compared with the direct
C-ABI call (the ctypes mirror).  The last test asserts that every command of the gateway's table has run.

This is the layer north_star calls the "thin C-ABI MEX/FFI": what a MATLAB host executes between KsysidHip.m / KmpcHip.m /
evaluate_rand_models_hip.m and libkoopman_hip.so.  The flows at the end replay what those .m files do, call for call."""
import os
import subprocess
import sys

import numpy as np
import pytest

import koopman_realizations_amd as kra
from koopman_realizations_amd import _ffi as F
from conftest import synth_pairs
import mexshim as ms

pytestmark = pytest.mark.gpu

USED = set()


def _f(x):
    return float(np.asarray(x).reshape(-1)[0])


def _i(x):
    return int(np.asarray(x).reshape(-1)[0])


def mex(cmd, *args, nargout=1):
    USED.add(cmd)
    return ms.kp_mex(cmd, *args, nargout=nargout)


def desc_of(model_type, nzeta, m, deg, pcs=None, extra=()):
    """The struct KsysidHip.hip_descriptor builds for obs_type {'poly'}."""
    nv = nzeta + (m if model_type == "nonlinear" else 0)
    e = kra.poly_exponent_table(nv, deg)[nv:].astype(np.uint8)
    return dict(model_type=np.int32(F.MODEL[model_type]), nzeta=np.int32(nzeta), m=np.int32(m),
                block_type=np.array([[0]], dtype=np.int32), block_count=np.array([[e.shape[0]]], dtype=np.int32),
                poly_exps=np.asfortranarray(e.T), gauss_centres=None, pcs=pcs), e


@pytest.fixture(scope="module")
def h():
    hh = mex("create", 0)
    yield hh
    mex("destroy", hh, nargout=0)


@pytest.fixture(scope="module")
def small(ctx, h):
    """bilinear poly-2 on 3 states, 2 inputs: W = 30."""
    p = synth_pairs(3000, 3, 2, seed=11)
    d, e = desc_of("bilinear", 3, 2, 2)
    b = mex("basis_create", h, d)
    bp = kra.Basis(ctx, "bilinear", 3, 2, [("poly", e)])
    s = mex("snapshots_upload", h, p["alpha"], p["beta"], p["u"])
    sp = kra.Snapshots(ctx, p["alpha"], p["beta"], p["u"])
    yield {"p": p, "d": d, "e": e, "b": b, "bp": bp, "s": s, "sp": sp, "W": bp.W, "N": bp.N}
    mex("snapshots_destroy", s, nargout=0)
    mex("basis_destroy", b, nargout=0)


def test_context_commands(ctx, h):
    assert _i(mex("device_count")) >= 1
    name, ncu, hbm = mex("device_info", h, nargout=3)
    info = ctx.info()
    assert name == info["name"] and _i(ncu) == info["num_cu"] and _i(hbm) == info["hbm_bytes"]
    assert isinstance(mex("last_error"), str) and isinstance(mex("last_error", h), str)
    assert _f(mex("timer_get", h, 0)) >= 0.0
    mex("synchronize", h, nargout=0)
    tab, names = mex("commands", nargout=2)
    assert tab.shape == (len(names.split("\n")), 3)
    h2 = mex("create", 0)                                   # the device's context is shared and reference counted
    assert h2 == h
    mex("destroy", h2, nargout=0)
    assert _i(mex("device_count")) >= 1                    # ... and still alive after one reference was dropped
    with pytest.raises(ms.MexError) as e:
        mex("create", 99)
    assert e.value.identifier in ("kp:usage", "kp:error")


def test_stale_and_foreign_handles_are_rejected(ctx, h):
    """A handle is a raw pointer in a uint64; MATLAB objects that carry one can be saved, loaded and outlive `clear kp_mex`
    (Ksysid.save_class, Ksysid.m:436-448).  The gateway keeps a registry of the handles it handed out: releasing one twice,
    using one after its release, or presenting a value it never produced is an ERROR MATLAB can catch - never a free() of a
    stale pointer (round-4 advisor finding on matlab/KpOwner.m)."""
    d, _ = desc_of("bilinear", 2, 1, 2)
    b = mex("basis_create", h, d)
    assert mex("basis_dims", b).ravel().tolist()[0] == 2
    mex("basis_destroy", b, nargout=0)
    for cmd, args in (("basis_destroy", (b,)), ("basis_dims", (b,)), ("mpc_destroy", (b,)), ("traj_destroy", (ms.Handle(0x7f0000001000),)),
                      ("snapshots_destroy", (ms.Handle(int(b) + 8),)), ("lift", (h, b, 1, np.zeros((3, 2)))), ("destroy", (ms.Handle(12345678),))):
        with pytest.raises(ms.MexError) as e:
            mex(cmd, *args, nargout=0)
        assert e.value.identifier == "kp:handle", (cmd, e.value.identifier)
    # the context is still usable, and a handle of one kind released through its own command still works afterwards
    b2 = mex("basis_create", h, d)
    assert mex("basis_dims", b2).ravel().tolist()[0] == 2
    # a LIVE handle of the wrong kind (round-5 advisor finding: the registry checked addresses only): a basis handle where an MPC,
    # a trajectory set, a snapshot object or a context is expected, and the context's handle where a basis is - refused, nothing
    # cast, nothing freed; both stay usable
    for cmd, args in (("mpc_destroy", (b2,)), ("mpc_dims", (b2,)), ("traj_destroy", (b2,)), ("snapshots_destroy", (b2,)), ("synchronize", (b2,)),
                      ("basis_destroy", (h,)), ("basis_dims", (h,)), ("lift", (b2, h, 1, np.zeros((3, 2))))):
        with pytest.raises(ms.MexError) as e:
            mex(cmd, *args, nargout=0)
        assert e.value.identifier == "kp:handle" and "wrong kind" in str(e.value), (cmd, str(e.value))
    assert mex("basis_dims", b2).ravel().tolist()[0] == 2
    mex("synchronize", h, nargout=0)
    mex("basis_destroy", b2, nargout=0)


def test_dictionary_lift_and_eig(ctx, h, small):
    assert mex("basis_dims", small["b"]).ravel().tolist() == [small["bp"].nvars, small["bp"].nfull, small["N"], small["W"]]
    assert mex("basis_desc_dims", small["d"]).ravel().tolist() == [small["bp"].nvars, small["bp"].nfull, small["N"], small["W"]]
    z = small["p"]["alpha"][:50]; u = small["p"]["u"][:50]
    for what in (0, 1, 2):
        got = mex("lift", h, small["b"], what, z, u)
        assert np.array_equal(got, small["bp"].lift(what, z, u))
    assert np.array_equal(mex("lift", h, small["b"], 1, z), small["bp"].lift(1, z))       # u omitted (econ lift)
    assert np.array_equal(mex("lift", h, small["b"], 1, z, None), small["bp"].lift(1, z))  # u = [] as KsysidHip passes it
    with pytest.raises(ms.MexError):
        mex("lift", h, small["b"], 2, z, u[:10])
    S = np.cov(np.random.default_rng(0).standard_normal((40, 12)), rowvar=False)
    V, lam, sweeps = mex("sym_eig", h, S, nargout=3)
    assert _i(sweeps) >= 1 and np.abs(V @ np.diag(lam.ravel()) @ V.T - S).max() <= 1e-12
    lam_p, V_p, _ = ctx.sym_eig(S)
    assert np.allclose(np.sort(lam.ravel())[::-1], lam_p, rtol=0, atol=1e-13)
    # a dictionary with a projection (dim_red): pcs travels as an nfull x k double matrix
    pcs = np.linalg.qr(np.random.default_rng(1).standard_normal((small["bp"].nfull, 4)))[0]
    d2, e2 = desc_of("bilinear", 3, 2, 2, pcs=pcs)
    b2 = mex("basis_create", h, d2)
    bp2 = kra.Basis(ctx, "bilinear", 3, 2, [("poly", e2)], pcs)
    assert mex("basis_dims", b2).ravel().tolist() == [3, bp2.nfull, 3 + 4 + 1, (3 + 4 + 1) * 3]
    assert np.array_equal(mex("lift", h, b2, 1, z), bp2.lift(1, z))
    mex("basis_destroy", b2, nargout=0)
    bp2.close()


def test_fit_commands(ctx, h, small):
    W, p = small["W"], small["p"]
    Kref = kra.fit(ctx, small["bp"], small["sp"])[0]
    K = mex("fit", h, small["b"], small["s"], np.inf)
    assert K.shape == (W, W) and np.array_equal(K, Kref)
    assert _i(mex("last_rank", h)) == W == ctx.last_rank()
    assert _f(mex("last_pivot_ratio", h)) == ctx.last_pivot_ratio()
    G, Cm = mex("fit_gram", h, small["b"], small["s"], nargout=2)
    Gp, Cp = kra.fit_gram(ctx, small["bp"], small["sp"])
    assert np.array_equal(G, Gp) and np.array_equal(Cm, Cp)
    assert np.array_equal(mex("fit_solve", h, G, Cm), ctx.fit_solve(Gp, Cp))
    assert np.array_equal(mex("fit_solve", h, G, Cm[:, :5]), ctx.fit_solve(Gp, Cp[:, :5]))
    t = 0.3 * np.abs(Kref).sum()
    Kl, it = mex("fit_lasso", h, G, Cm, t, nargout=2)
    Klp, itp = ctx.fit_lasso(Gp, Cp, t)
    assert np.array_equal(Kl, Klp) and _i(it) == itp and abs(np.abs(Kl).sum() - t) <= 1e-8 * t
    tv = np.array([0.5, 0.2, 5.0]) * np.abs(Kref).sum()
    Kb, itb = mex("fit_lasso_batch", h, G, Cm, tv, nargout=2)
    Kbp, itbp = ctx.fit_lasso_batch(Gp, Cp, tv)
    assert Kb.shape == (W, W, 3) and all(np.array_equal(Kb[:, :, i], Kbp[i]) for i in range(3)) and itb.ravel().tolist() == itbp.tolist()
    # a vector of lasso values in ONE call (train_models, Ksysid.m:1372-1387): W x W x n stack
    las = np.array([[np.inf, 0.5 * np.abs(Kref).sum() / small["N"], 0.1 * np.abs(Kref).sum() / small["N"]]])
    Ks = mex("fit", h, small["b"], small["s"], las)
    Ksp = kra.fit(ctx, small["bp"], small["sp"], las.ravel())
    assert Ks.shape == (W, W, 3) and all(np.array_equal(Ks[:, :, i], Ksp[i]) for i in range(3))
    Kr = mex("fit_refine", h, small["b"], small["s"], K, 1)
    assert np.array_equal(Kr, kra.fit_refine(ctx, small["bp"], small["sp"], Kref, 1))
    # sharded entry points without a communicator are the plain ones
    assert np.array_equal(mex("fit_sharded", h, small["b"], small["s"], np.inf), Kref)
    Gs, Cs = mex("fit_gram_sharded", h, small["b"], small["s"], nargout=2)
    assert np.array_equal(Gs, Gp) and np.array_equal(Cs, Cp)
    # the asynchronous pipeline: enqueue, fetch by batch index
    mex("fit_async_slots", h, 8, nargout=0)
    p2 = synth_pairs(3000, 3, 2, seed=12)
    s2 = mex("snapshots_upload", h, p2["alpha"], p2["beta"], p2["u"])
    sp2 = kra.Snapshots(ctx, p2["alpha"], p2["beta"], p2["u"])
    K2ref = kra.fit(ctx, small["bp"], sp2)[0]
    for snap in (small["s"], s2, small["s"]):
        mex("fit_async", h, small["b"], snap, nargout=0)
    mex("synchronize", h, nargout=0)
    got = [mex("fit_get_K", h, q, W) for q in range(3)]
    for g_, r_ in zip(got, (Kref, K2ref, Kref)):
        assert np.abs(g_ - r_).max() <= 1e-11 * np.abs(r_).max()
    with pytest.raises(ms.MexError) as e:
        mex("fit_get_K", h, 3, W)
    assert e.value.identifier == "kp:error" and "-1" in e.value.message
    mex("fit_async_slots", h, 128, nargout=0)
    # refill in place (kp_snapshots_update) and the MEX file's own resident object
    mex("snapshots_update", h, s2, p["alpha"], p["beta"], p["u"], nargout=0)
    assert np.array_equal(mex("fit", h, small["b"], s2, np.inf), Kref)
    sr = mex("snapshots_resident", h, p2["alpha"], p2["beta"], p2["u"])
    assert np.abs(mex("fit", h, small["b"], sr, np.inf) - K2ref).max() <= 1e-12 * np.abs(K2ref).max()
    sr2 = mex("snapshots_resident", h, p["alpha"], p["beta"], p["u"])
    assert sr2 == sr and np.array_equal(mex("fit", h, small["b"], sr2, np.inf), Kref)
    with pytest.raises(ms.MexError):
        mex("snapshots_upload", h, p["alpha"], p["beta"][:-1], p["u"])
    mex("snapshots_destroy", s2, nargout=0)
    sp2.close()


def test_rank_deficient_fit_warns_like_mldivide(h):
    """MATLAB's `\\` warns 'Rank deficient, rank = ...' and returns a basic solution (Ksysid.m:1069 on a dictionary with
    dependent columns); the gateway raises the warning through mexWarnMsgIdAndTxt and 'last_rank' reports the rank."""
    rng = np.random.default_rng(3)
    a = rng.uniform(-1, 1, (500, 2)); a[:, 1] = a[:, 0]              # two identical states: dependent monomials
    u = rng.uniform(-1, 1, (500, 1))
    d, _ = desc_of("linear", 2, 1, 2)
    b = mex("basis_create", h, d)
    s = mex("snapshots_upload", h, a, 0.9 * a, u)
    K = mex("fit", h, b, s, np.inf)
    warn = ms.last_warning
    assert warn[0] == "kp:rankDeficient" and "Rank deficient, rank = " in warn[1]
    r = _i(mex("last_rank", h))
    assert r < K.shape[0] and np.isfinite(K).all() and f"rank = {r}" in warn[1]
    mex("fit", h, b, s, np.inf)
    mex("snapshots_destroy", s, nargout=0)
    mex("basis_destroy", b, nargout=0)


def test_batch_fit_models_and_rollouts(ctx, h):
    nb, Ns = 6, 400
    rng = np.random.default_rng(5)
    a = rng.uniform(-1, 1, (nb * Ns, 1)); u = rng.uniform(-1, 1, (nb * Ns, 1))
    bb = 0.9 * a + 0.1 * u - 0.05 * a ** 3
    d, e = desc_of("linear", 1, 1, 4)
    b = mex("basis_create", h, d)
    bp = kra.Basis(ctx, "linear", 1, 1, [("poly", e)])
    s = mex("snapshots_upload", h, a, bb, u)
    sp = kra.Snapshots(ctx, a, bb, u)
    K, G, Cm, st = mex("fit_batch", h, b, s, nb, Ns, nargout=4)
    Kp, Gp, Cp, stp = ctx.fit_batch(bp, sp, nb)
    W, N = bp.W, bp.N
    assert K.shape == (W, W, nb) and st.ravel().tolist() == stp.tolist()
    for i in range(nb):
        assert np.array_equal(K[:, :, i], Kp[i]) and np.array_equal(G[:, :, i], Gp[i]) and np.array_equal(Cm[:, :, i], Cp[i])
    A, B, M, st2 = mex("model_project_batch", h, K, G, Cm, N, 1, nargout=4)
    Ap, Bp, st2p = ctx.model_project_batch(Kp, Gp, Cp, N, 1)
    assert A.shape == (N, N, nb) and all(np.array_equal(A[:, :, i], Ap[i]) and np.array_equal(B[:, :, i], Bp[i]) for i in range(nb))
    A1, B1, M1 = mex("model_project", h, K[:, :, 0], G[:, :, 0], Cm[:, :, 0], N, 1, nargout=3)
    A1p, B1p, M1p = ctx.model_project(Kp[0], Gp[0], Cp[0], N, 1)
    assert np.array_equal(A1, A1p) and np.array_equal(B1, B1p) and np.array_equal(M1, M1p)
    # rollouts: one model, then a batch (z0 N x batch, U T x m x batch)
    T = 60
    U = rng.uniform(-1, 1, (T, 1)); z0 = bp.lift(1, np.array([[0.3]]))[0]
    Y = mex("rollout", h, 0, A1, B1, z0, U, 1)
    assert np.array_equal(Y, ctx.rollout("linear", A1p, B1p, z0, U, 1))
    Ub = rng.uniform(-1, 1, (T, 1, nb)); z0b = np.stack([bp.lift(1, np.array([[0.1 * i]]))[0] for i in range(nb)], axis=1)
    Yb = mex("rollout", h, 0, A, B, z0b, Ub, 1)
    Ybp = ctx.rollout("linear", np.stack([A[:, :, i] for i in range(nb)]), np.stack([B[:, :, i] for i in range(nb)]), z0b.T,
                      np.transpose(Ub, (2, 0, 1)), 1)
    assert Yb.shape == (T, 1, nb) and all(np.array_equal(Yb[:, :, i], Ybp[i]) for i in range(nb))
    mex("snapshots_destroy", s, nargout=0); mex("basis_destroy", b, nargout=0)
    # nonlinear rollout
    dn, en = desc_of("nonlinear", 1, 1, 3)
    bn = mex("basis_create", h, dn)
    bnp = kra.Basis(ctx, "nonlinear", 1, 1, [("poly", en)])
    Kf = rng.standard_normal((1, bnp.N)) * 0.1
    Z = mex("rollout_nl", h, bn, Kf, np.array([[0.2]]), U)
    assert np.array_equal(Z, ctx.rollout_nl(bnp, Kf, np.array([0.2]), U))
    mex("basis_destroy", bn, nargout=0)


def _stacks(nb=12, k=4, T=201, Tv=151, seed=9):
    """Raw trials of nb 1-D systems (stable cubic maps driven by steps), as rows x n x nb stacks."""
    rng = np.random.default_rng(seed)
    Y = np.zeros((k * T, 1, nb)); U = np.zeros((k * T, 1, nb)); Yv = np.zeros((Tv, 1, nb)); Uv = np.zeros((Tv, 1, nb))
    for s in range(nb):
        a, c, g = rng.uniform(0.7, 0.95), rng.uniform(0.05, 0.2), rng.uniform(0.1, 0.5)
        def run(Tn):
            u = np.repeat(rng.uniform(-1, 1, Tn // 20 + 1), 20)[:Tn]
            y = np.zeros(Tn); y[0] = rng.uniform(-0.5, 0.5)
            for t in range(Tn - 1):
                y[t + 1] = a * y[t] - c * y[t] ** 3 + g * u[t]
            return y, u
        for j in range(k):
            y, u = run(T)
            Y[j * T:(j + 1) * T, 0, s] = y; U[j * T:(j + 1) * T, 0, s] = u
        Yv[:, 0, s], Uv[:, 0, s] = run(Tv)
    return Y, U, Yv, Uv, k


def test_sweep_commands(ctx, h):
    Y, U, Yv, Uv, k = _stacks()
    nb = Y.shape[2]
    from koopman_realizations_amd.device import Traj
    tp = Traj(ctx, np.transpose(Y, (2, 0, 1)), np.transpose(U, (2, 0, 1)), k, np.transpose(Yv, (2, 0, 1)), np.transpose(Uv, (2, 0, 1)))
    t = mex("traj_upload", h, Y, U, Yv, Uv, k)
    assert mex("traj_dims", t).ravel().tolist() == [nb, k, Y.shape[0] // k, 1, 1, Yv.shape[0]]
    assert np.array_equal(mex("traj_scale", t), tp.scale().T)
    # the same object in three steps
    t2 = mex("traj_create", h, nb, k, Y.shape[0] // k, 1, 1, Yv.shape[0])
    for w, blk in enumerate((Y, U, Yv, Uv)):
        mex("traj_put", t2, w, blk, nargout=0)
    mex("traj_finish", t2, nargout=0)
    assert np.array_equal(mex("traj_scale", t2), tp.scale().T)
    with pytest.raises(ms.MexError):
        mex("traj_put", t2, 0, Y, nargout=0)                 # a finished object holds scaled data
    with pytest.raises(ms.MexError):
        mex("traj_upload", h, Y, U[:-1], Yv, Uv, k)
    for mt, D, las in (("linear", 5, 1e6), ("bilinear", 3, 1e6), ("nonlinear", 3, 4.0)):
        d, e = desc_of(mt, 1, 1, D)
        b = mex("basis_create", h, d)
        bp = kra.Basis(ctx, mt, 1, 1, [("poly", e)])
        err, st = mex("sweep_eval_nested", h, t, b, las, D, nargout=2)
        errp, stp = tp.sweep_eval_nested(bp, D, las)
        assert err.shape == (1, nb, D) and np.array_equal(np.transpose(err, (2, 1, 0)), errp, equal_nan=True)
        assert np.array_equal(st.T, stp)
        W2 = _i(mex("basis_desc_dims", desc_of(mt, 1, 1, 2)[0])[0, 3])
        Kd = mex("sweep_nested_get_K", h, nb, bp.W, D, 1, W2)
        assert Kd.shape[2] == nb and np.isfinite(Kd).all()
        e1, K1, s1 = mex("sweep_eval", h, t2, b, las, nargout=3)
        e1p, K1p, s1p = tp.sweep_eval(bp, las, want_K=True)
        assert np.array_equal(e1.T, e1p, equal_nan=True) and s1.ravel().tolist() == s1p.tolist()
        assert all(np.array_equal(K1[:, :, i], K1p[i], equal_nan=True) for i in range(nb))
        mex("basis_destroy", b, nargout=0)
        bp.close()
    mex("traj_destroy", t, nargout=0); mex("traj_destroy", t2, nargout=0)
    tp.close()


@pytest.fixture(scope="module")
def mpc_model(ctx):
    """A bilinear model fitted on synthetic pairs (N = 10, m = 2) and the controller settings of example_control.m."""
    p = synth_pairs(4000, 3, 2, seed=21)
    e = kra.poly_exponent_table(3, 2)[3:].astype(np.uint8)
    bp = kra.Basis(ctx, "bilinear", 3, 2, [("poly", e)])
    K = kra.fit(ctx, bp, kra.Snapshots(ctx, p["alpha"], p["beta"], p["u"]))[0]
    N = bp.N
    UT = K.T
    A = np.asfortranarray(UT[:N, :N]); B = np.asfortranarray(UT[:N, N:])
    proj = np.hstack([np.eye(2), np.zeros((2, N - 2))])
    args = dict(Np=8, proj=proj, q_run=10.0, q_term=100.0, r=np.array([3e-3, 2e-3]), lo=np.array([-0.9, -0.9]), hi=np.array([0.9, 0.9]),
                slope=0.2, smooth=None)
    return {"A": A, "B": B, "N": N, "bp": bp, "e": e, **args}


def test_mpc_commands(ctx, h, mpc_model):
    mm = mpc_model
    mp = kra.Mpc(ctx, "bilinear", mm["A"], mm["B"], mm["Np"], mm["proj"], mm["q_run"], mm["q_term"], mm["r"], mm["lo"], mm["hi"], mm["slope"], None)
    m_ = mex("mpc_create", h, 1, mm["A"], mm["B"], mm["Np"], mm["proj"], mm["q_run"], mm["q_term"], mm["r"], mm["lo"], mm["hi"], mm["slope"], None)
    assert mex("mpc_dims", m_).ravel().tolist() == [mp.nvar, mp.nrows]
    d, _ = desc_of("bilinear", 3, 2, 2)
    b = mex("basis_create", h, d)
    rng = np.random.default_rng(2)
    for it in (1, 2):
        zeta = rng.uniform(-0.5, 0.5, 3); up = rng.uniform(-0.3, 0.3, 2); Yr = rng.uniform(-0.5, 0.5, 2 * (mm["Np"] + 1))
        U, z = mex("mpc_step_zeta", m_, b, zeta, up, Yr, it, nargout=2)
        Up, zp, stp = mp.step_zeta(mm["bp"], zeta, up, Yr, it)
        assert stp == 0 and np.abs(U - Up).max() <= 1e-10 and np.array_equal(z.ravel(), zp)
        U2, st2 = mex("mpc_step", m_, z, up, Yr, it, nargout=2)
        assert _i(st2) == 0 and np.abs(U2 - Up).max() <= 1e-10
    Udef = mex("mpc_step_zeta", m_, b, zeta, up, Yr)                      # iters omitted: 1
    assert np.abs(Udef - mp.step_zeta(mm["bp"], zeta, up, Yr, 1)[0]).max() <= 1e-10
    Hq, f, Aq, bq = mex("mpc_last_qp", m_, nargout=4)
    Hp, fp, Ap, bp_ = mp.last_qp()
    assert Hq.shape == Hp.shape and np.allclose(Hq, Hp, rtol=0, atol=1e-12) and Aq.shape == Ap.shape and bq.shape[0] == bp_.shape[0]
    us, counts = mex("mpc_last_profile", m_, nargout=2)
    assert us.shape == (1, 6) and us[0, 5] > 0 and counts.shape == (1, 2)
    us16 = mex("mpc_last_stamps", m_)
    assert us16.shape == (1, 16) and us16[0, 0] == 0 and abs(us16[0, 5] - us[0, 5]) < 1e-9 and 0 < us16[0, 10] < us16[0, 11] < us16[0, 5]
    # batch: columns = problems
    nb = 37
    Z = np.asfortranarray(mm["bp"].lift(1, rng.uniform(-0.5, 0.5, (nb, 3))).T)
    UP = rng.uniform(-0.3, 0.3, (2, nb)); YR = rng.uniform(-0.5, 0.5, (2 * (mm["Np"] + 1), nb))
    Ub, stb = mex("mpc_step_batch", m_, Z, UP, YR, nargout=2)
    Ubp, stbp = mp.step_batch(Z.T, UP.T, YR.T)
    assert Ub.shape == (mm["Np"], 2, nb) and stb.ravel().tolist() == stbp.tolist()
    assert all(np.abs(Ub[:, :, i] - Ubp[i]).max() <= 1e-10 for i in range(nb))
    # an infeasible problem comes back as NaN with the call succeeding (Ksim.m:220-222 tests any(isnan(U)))
    mex("mpc_set_state_bounds", m_, np.array([0.5, 0.5, 0.5]), np.array([0.4, 0.4, 0.4]), nargout=0)
    Ubad = mex("mpc_step_zeta", m_, b, zeta, up, Yr, 1)
    assert np.isnan(Ubad).all()
    mex("mpc_set_state_bounds", m_, None, None, nargout=0)               # n = 0 removes them
    assert np.isfinite(mex("mpc_step_zeta", m_, b, zeta, up, Yr, 1)).all()
    # the generic QP shim (quadprog_gurobi.m:1)
    Mq = rng.standard_normal((6, 6)); H = Mq @ Mq.T + np.eye(6); fq = rng.standard_normal(6)
    Aqp = np.vstack([np.eye(6), -np.eye(6)]); bqp = np.full(12, 0.3)
    x, stq = mex("qp_solve", h, H, fq, Aqp, bqp, nargout=2)
    xp, stqp = ctx.qp_solve(H, fq, Aqp, bqp)
    assert _i(stq) == stqp == 0 and np.array_equal(x.ravel(), xp)
    mex("mpc_destroy", m_, nargout=0)
    mex("basis_destroy", b, nargout=0)
    mp.close()


def test_multi_gpu_commands_with_one_device_listed_twice(ctx, h, small, mpc_model):
    """kp_mex('multi_*'): one caller, a worker thread + context per listed device.  On this one-GPU box the device is listed
    twice and three times: the fan-out, the ragged shards and the exchange of the snapshot-sharded fit are the real code."""
    p, W, N = small["p"], small["W"], small["N"]
    Kref = kra.fit(ctx, small["bp"], small["sp"])[0]
    l1 = np.abs(Kref).sum() / N
    las = np.array([np.inf, 0.5 * l1, 0.3 * l1, 0.2 * l1, 0.1 * l1])       # 5 values on 2 and 3 workers: ragged shards
    ref = kra.fit(ctx, small["bp"], small["sp"], las)
    for ids in ([0, 0], [0, 0, 0], [0]):
        g = mex("multi_create", np.array(ids, dtype=np.float64))
        assert _i(mex("multi_size", g)) == len(ids)
        Ks = mex("multi_fit", g, small["d"], p["alpha"], p["beta"], p["u"], las)
        assert Ks.shape == (W, W, 5) and all(np.array_equal(Ks[:, :, i], ref[i]) for i in range(5))
        tm = mex("multi_timers", g)
        assert tm.shape == (4, len(ids)) and (tm[3] > 0).all()
        # ONE fit sharded over snapshots: [G | C] of every worker summed on worker 0, then the same solve
        Ksh = mex("multi_fit_sharded", g, small["d"], p["alpha"], p["beta"], p["u"], np.array([np.inf, 0.3 * l1]))
        assert np.abs(Ksh[:, :, 0] - ref[0]).max() <= 1e-11 * np.abs(ref[0]).max()
        assert np.abs(Ksh[:, :, 1] - ref[2]).max() <= 1e-7 * np.abs(ref[2]).max()
        if len(ids) == 1:
            assert np.array_equal(Ksh[:, :, 0], ref[0])                       # one worker: the plain fit, bit for bit
        # the random-system sweep, systems dealt in contiguous chunks
        Y, U, Yv, Uv, k = _stacks(nb=11)
        t = mex("multi_traj_upload", g, Y, U, Yv, Uv, k)
        t1 = mex("traj_upload", h, Y, U, Yv, Uv, k)
        for mt, D, lv in (("linear", 5, 1e6), ("nonlinear", 3, 4.0)):
            d, _ = desc_of(mt, 1, 1, D)
            err, st = mex("multi_sweep_eval_nested", g, t, d, lv, D, nargout=2)
            b = mex("basis_create", h, d)
            err1, st1 = mex("sweep_eval_nested", h, t1, b, lv, D, nargout=2)
            assert np.array_equal(err, err1, equal_nan=True) and np.array_equal(st, st1)
            mex("basis_destroy", b, nargout=0)
        mex("multi_traj_destroy", t, nargout=0); mex("traj_destroy", t1, nargout=0)
        # batched MPC, problems dealt in contiguous chunks
        mm = mpc_model
        mmpc = mex("multi_mpc_create", g, 1, mm["A"], mm["B"], mm["Np"], mm["proj"], mm["q_run"], mm["q_term"], mm["r"], mm["lo"], mm["hi"], mm["slope"], None)
        m1 = mex("mpc_create", h, 1, mm["A"], mm["B"], mm["Np"], mm["proj"], mm["q_run"], mm["q_term"], mm["r"], mm["lo"], mm["hi"], mm["slope"], None)
        rng = np.random.default_rng(4)
        nb = 23
        Z = np.asfortranarray(mm["bp"].lift(1, rng.uniform(-0.5, 0.5, (nb, 3))).T)
        UP = rng.uniform(-0.3, 0.3, (2, nb)); YR = rng.uniform(-0.5, 0.5, (2 * (mm["Np"] + 1), nb))
        Um, sm = mex("multi_mpc_step_batch", mmpc, Z, UP, YR, nargout=2)
        U1, s1 = mex("mpc_step_batch", m1, Z, UP, YR, nargout=2)
        assert np.array_equal(Um, U1) and np.array_equal(sm, s1)
        mex("multi_mpc_set_state_bounds", mmpc, np.array([-5.0, -5.0]), np.array([5.0, 5.0]), nargout=0)
        mex("mpc_set_state_bounds", m1, np.array([-5.0, -5.0]), np.array([5.0, 5.0]), nargout=0)
        Um2, _ = mex("multi_mpc_step_batch", mmpc, Z, UP, YR, nargout=2)
        U12, _ = mex("mpc_step_batch", m1, Z, UP, YR, nargout=2)
        assert np.allclose(Um2, U12, rtol=0, atol=1e-12, equal_nan=True)
        mex("multi_mpc_destroy", mmpc, nargout=0); mex("mpc_destroy", m1, nargout=0)
        mex("multi_destroy", g, nargout=0)
    with pytest.raises(ms.MexError):
        mex("multi_size", ms.Handle(1234))                                    # not an object of this session


_COMM_SCRIPT = r"""
import sys, os, threading, numpy as np
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, sys.argv[1] + "/tests")
import mexshim as ms
from conftest import synth_pairs
import koopman_realizations_amd as kra
h = ms.kp_mex('create', 0)
uid = ms.kp_mex('comm_unique_id')
assert uid.shape == (1, 128) and uid.dtype == np.uint8 and uid.any()
done = threading.Event()
def dog():
    if not done.wait(100.0):
        print("RCCL_INIT_TIMEOUT", flush=True); os._exit(0)
threading.Thread(target=dog, daemon=True).start()
ms.kp_mex('comm_create', h, uid, 0, 1, nargout=0)
done.set()
assert ms.kp_mex('comm_info', h).ravel().tolist() == [0, 1]
v = np.arange(5.0)
assert np.array_equal(ms.kp_mex('comm_allgather', h, v), v.reshape(5, 1))
assert np.array_equal(ms.kp_mex('comm_allreduce_sum', h, np.array([1.5, -2.0])).ravel(), [1.5, -2.0])
e = kra.poly_exponent_table(3, 2)[3:].astype(np.uint8)
d = dict(model_type=np.int32(1), nzeta=np.int32(3), m=np.int32(2), block_type=np.array([[0]], dtype=np.int32),
         block_count=np.array([[e.shape[0]]], dtype=np.int32), poly_exps=np.asfortranarray(e.T), gauss_centres=None, pcs=None)
b = ms.kp_mex('basis_create', h, d)
W = int(ms.kp_mex('basis_dims', b)[0, 3])
p = synth_pairs(3000, 3, 2, seed=4)
s = ms.kp_mex('snapshots_upload', h, p['alpha'], p['beta'], p['u'])
Kref = ms.kp_mex('fit', h, b, s, np.inf)
for _ in range(3):
    ms.kp_mex('fit_async', h, b, s, nargout=0)
Kall = ms.kp_mex('comm_allgather_fit', h, 2, W)
assert Kall.shape == (W, W) and np.abs(Kall - Kref).max() <= 1e-11 * np.abs(Kref).max()
Kst = ms.kp_mex('comm_allgather_fits', h, 0, 3, W)
assert Kst.shape == (W, W, 3) and all(np.abs(Kst[:, :, i] - Kref).max() <= 1e-11 * np.abs(Kref).max() for i in range(3))
Kg = ms.kp_mex('comm_gather_fits', h, 0, 1, 2, W)
assert Kg.shape == (W, W, 2) and np.array_equal(Kg, Kst[:, :, 1:])
assert np.abs(ms.kp_mex('fit_sharded', h, b, s, np.inf) - Kref).max() <= 1e-12 * np.abs(Kref).max()     # all-reduce over one rank
ms.kp_mex('comm_destroy', h, nargout=0)
ms.kp_mex('comm_abandon', h, nargout=0)
assert ms.kp_mex('comm_info', h).ravel().tolist()[1] in (0, 1)
assert np.array_equal(ms.kp_mex('fit', h, b, s, np.inf), Kref)
ms.lib().shim_run_at_exit()
print("COMM_OK")
"""


def test_comm_commands_with_a_one_rank_communicator():
    """kp_mex('comm_*') over a one-rank RCCL communicator, in a fresh interpreter (as a MATLAB worker per GPU would run it)."""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, NCCL_SOCKET_IFNAME=os.environ.get("NCCL_SOCKET_IFNAME", "lo"), NCCL_IB_DISABLE=os.environ.get("NCCL_IB_DISABLE", "1"))
    r = subprocess.run([sys.executable, "-c", _COMM_SCRIPT, root], capture_output=True, text=True, timeout=400, env=env)
    for c in ("comm_unique_id", "comm_create", "comm_info", "comm_allgather", "comm_allreduce_sum", "comm_allgather_fit", "comm_allgather_fits",
              "comm_gather_fits", "comm_destroy", "comm_abandon"):
        USED.add(c)
    if "RCCL_INIT_TIMEOUT" in r.stdout:
        pytest.skip("ncclCommInitRank of ONE rank does not return on this box")
    assert r.returncode == 0 and "COMM_OK" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]


def test_flow_of_ksysidhip_get_koopman_and_train_models(ctx, h, golden):
    """The calls of matlab/KsysidHip.m, in its order, on the arm data with dim_red (the settings of example_sysid.m): constructor
    (create, basis_create, basis_dims), get_Koopman (snapshots_resident, fit, last_rank, last_pivot_ratio -> fit_refine), koopData
    lift, get_model (fit_gram, model_project), and the vector branch of train_models (ONE fit call for the whole lasso vector) -
    against the tested Python mirror (koopman_realizations_amd.Ksysid), which makes the same calls through ctypes."""
    g = golden["arm_data"]
    lens = g["train_len"]; off = np.concatenate([[0], np.cumsum(lens)])
    train = [{"t": g["train_t"][a:b], "y": g["train_y"][a:b], "u": g["train_u"][a:b]} for a, b in zip(off[:-1], off[1:])]
    data = {"train": train, "val": [{"t": g["val_t"], "y": g["val_y"], "u": g["val_u"]}]}
    ks = kra.Ksysid(data, ctx=ctx, model_type="linear", obs_type=["poly"], obs_degree=[2], snapshots=np.inf, lasso=[np.inf], delays=0, dim_red=True)
    ks.train_models()
    sp = ks.snapshotPairs
    nv = 6
    e = kra.poly_exponent_table(nv, 2)[nv:].astype(np.uint8)
    d = dict(model_type=np.int32(0), nzeta=np.int32(6), m=np.int32(3), block_type=np.array([[0]], dtype=np.int32),
             block_count=np.array([[e.shape[0]]], dtype=np.int32), poly_exps=np.asfortranarray(e.T), gauss_centres=None, pcs=ks.basis["pcs"])
    hc = mex("create", 0)
    b = mex("basis_create", hc, d)
    dims = mex("basis_dims", b).ravel()
    assert dims[2] == ks.params["N"]
    W = int(dims[3])
    s = mex("snapshots_resident", hc, sp["alpha"], sp["beta"], sp["u"])
    K = mex("fit", hc, b, s, np.inf)
    assert _i(mex("last_rank", hc)) == W
    ratio = _f(mex("last_pivot_ratio", hc))
    if ratio < 1e-5:                                            # KsysidHip.get_Koopman's rule (Ksysid.m:1069: `\` is a QR solve)
        K = mex("fit_refine", hc, b, s, K, 1)
    Kp = ks.koopData["K"]
    assert np.abs(K - Kp).max() <= 1e-13 * np.abs(Kp).max(), (ratio, np.abs(K - Kp).max())
    Px = mex("lift", hc, b, 1, sp["alpha"], None)
    assert np.array_equal(Px, ks.koopData["Px"])
    G, Cm = mex("fit_gram", hc, b, s, nargout=2)
    MA, MB, M = mex("model_project", hc, K, G, Cm, ks.params["N"], 3, nargout=3)     # the projected M*A, M*B (Ksysid.m:1224-1225)
    assert np.abs(MA - ks.model["A"]).max() <= 1e-10 * np.abs(ks.model["A"]).max()
    assert np.abs(MB - ks.model["B"]).max() <= 1e-10 * max(1.0, np.abs(ks.model["B"]).max())
    UT = K.T
    N_ = ks.params["N"]
    assert np.abs(M @ UT[:N_, :N_] - MA).max() <= 1e-9 * np.abs(MA).max()            # out.A = M * UT(1:N,1:N), as the parent forms it
    mex("basis_destroy", b, nargout=0)
    # vector branch of train_models (lasso property below 1e6): ONE gateway call, a W x W x n stack, value i = the parent's loop
    # (poly-3 with dim_red, the dictionary of example_sysid.m: cond(Px) 1.4e3)
    kb = kra.Ksysid(data, ctx=ctx, model_type="bilinear", obs_type=["poly"], obs_degree=[3], snapshots=np.inf, lasso=[1.0], delays=0, dim_red=True)
    e3 = kra.poly_exponent_table(nv, 3)[nv:].astype(np.uint8)
    db = dict(d, model_type=np.int32(1), pcs=kb.basis["pcs"], block_count=np.array([[e3.shape[0]]], dtype=np.int32), poly_exps=np.asfortranarray(e3.T))
    bb = mex("basis_create", hc, db)
    Kls = kra.fit(ctx, kb.basis_dev, kb._resident_snapshots(sp["alpha"], sp["beta"], sp["u"]))[0]
    las = np.array([[2.0, 0.5, 0.1]]) * np.abs(Kls).sum() / kb.params["N"]
    s = mex("snapshots_resident", hc, sp["alpha"], sp["beta"], sp["u"])
    Ks = mex("fit", hc, bb, s, las)
    kb.train_models(las.ravel())
    assert Ks.shape[2] == 3 == len(kb.koopData) and kb.model is kb.candidates[0]
    for i in range(3):
        Ki = kb.koopData[i]["K"]
        assert np.abs(Ks[:, :, i] - Ki).max() <= 1e-9 * np.abs(Ki).max()
    mex("basis_destroy", bb, nargout=0)
    # ... and on the poly-2 dictionary (cond(G) 3.5e10), where rounds 1-3 ended with "iteration cap reached": the values that the
    # projected-gradient iteration cannot finish come from the regularisation-path homotopy (kp_lasso_path.hip), budget met exactly
    kb2 = kra.Ksysid(data, ctx=ctx, model_type="bilinear", obs_type=["poly"], obs_degree=[2], snapshots=np.inf, lasso=[1.0], delays=0, dim_red=True)
    db2 = dict(d, model_type=np.int32(1), pcs=kb2.basis["pcs"])
    bb2 = mex("basis_create", hc, db2)
    Kls2 = kra.fit(ctx, kb2.basis_dev, kb2._resident_snapshots(sp["alpha"], sp["beta"], sp["u"]))[0]
    las2 = np.array([[0.5, 0.1]]) * np.abs(Kls2).sum() / kb2.params["N"]
    Ks2 = mex("fit", hc, bb2, s, las2)
    kb2.train_models(las2.ravel())
    for i in range(2):
        Ki = kb2.koopData[i]["K"]
        assert np.abs(Ks2[:, :, i] - Ki).max() <= 1e-9 * np.abs(Ki).max()
        assert abs(np.abs(Ks2[:, :, i]).sum() - las2[0, i] * kb2.params["N"]) <= 1e-11 * las2[0, i] * kb2.params["N"]
    mex("basis_destroy", bb2, nargout=0)
    mex("destroy", hc, nargout=0)


def test_every_gateway_command_has_been_executed():
    cmds = set(ms.commands())
    assert cmds - USED == set(), sorted(cmds - USED)
    assert ms.lib().shim_live_arrays() == 0
