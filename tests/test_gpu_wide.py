"""GPU parity tests of the WIDE path (W > 512): dictionaries the reference accepts as they come - `def_fourierLift` on the
arm's six states gives 728 functions (Ksysid.m:694-731; linear row 738 columns, bilinear row 2 940), poly-3 on a
delay-embedded state 816 (Ksysid.m:868-907) - and `K = Px \\ Py` on them (Ksysid.m:1069).  Device side: lifted panels in HBM +
TN products on the matrix pipe (csrc/kp_wide.hip, kp_tn_gemm.h), blocked Cholesky / substitution over all CUs
(csrc/kp_fit.hip `chol_solve_wide`).  Oracle: oracle/koopman_oracle.py on the same inputs.  Tolerances (f64):
  G, C   1e-12 relative to max|G|
  K      max(1e-9, 50 cond(Px)^2 eps) relative to max|K| against the SVD least-squares oracle (normal equations)
  solve  residual |G K - C| <= 1e-11 |G| |K| against a dense numpy solve on random SPD systems
"""
import os

import numpy as np
import pytest

import koopman_realizations_amd as kra
from koopman_realizations_amd import _ffi as F
from oracle import koopman_oracle as ko
from conftest import synth_pairs
from test_gpu_fit import make_basis

pytestmark = pytest.mark.gpu

EPS = 2.220446049250313e-16

WIDE_CASES = [
    ("linear", 6, 3, ["fourier"], [1], False),        # SURVEY row a7 at the size it names: 728 functions, W = 738
    ("bilinear", 6, 3, ["fourier"], [1], False),      # the same dictionary in a bilinear row: W = 2 940
    ("linear", 15, 3, ["poly"], [3], False),          # poly-3 on a delay-embedded arm state (nzeta = 15): 816 functions
    ("bilinear", 6, 3, ["poly"], [4], False),         # N = 210, W = 840: beyond the Kronecker kernel's 96 columns
    ("bilinear", 7, 1, ["poly"], [4], False),         # one input: N = 330, W = 660 - 3 weighted products per Gram
    ("bilinear", 6, 2, ["poly"], [4], False),         # two inputs: N = 210, W = 630 - 6 weighted products
]


@pytest.mark.parametrize("mt,nz,m,types,degs,dim_red", WIDE_CASES)
def test_wide_lift_and_gram_parity(ctx, mt, nz, m, types, degs, dim_red):
    Ns = 1003
    pairs = synth_pairs(Ns, nz, m, seed=7)
    dic = ko.build_dictionary(mt, nz, m, types, degs, pairs, dim_red)
    b = make_basis(ctx, dic)
    assert (b.nfull, b.N, b.W) == (dic.basis.nfull, dic.N, dic.W)
    assert b.W > 512
    Px, Py = ko.px_py(dic, pairs)
    np.testing.assert_allclose(b.lift(F.LIFT_ROW, pairs["alpha"], pairs["u"]), Px, atol=1e-13, rtol=0)
    snaps = kra.Snapshots(ctx, pairs["alpha"], pairs["beta"], pairs["u"])
    G, C = kra.fit_gram(ctx, b, snaps)
    Gr, Cr = ko.gram(Px, Py)
    scale = np.abs(Gr).max()
    assert np.abs(G - Gr).max() <= 1e-12 * scale
    assert np.abs(C - Cr).max() <= 1e-12 * scale
    assert (G == G.T).all()
    G2, C2 = kra.fit_gram(ctx, b, snaps)
    assert (G2 == G).all() and (C2 == C).all()          # fixed split / panel order: bitwise reproducible


def test_wide_gram_accumulates_over_panels(ctx):
    """The lifted panel is bounded (KP_WIDE_PANEL_MB); a snapshot matrix longer than one panel is accumulated panel by
    panel, ragged last panel included: same Grams as one panel to rounding."""
    pairs = synth_pairs(3001, 6, 3, seed=12)
    dic = ko.build_dictionary("linear", 6, 3, ["fourier"], [1])
    b = make_basis(ctx, dic)
    snaps = kra.Snapshots(ctx, pairs["alpha"], pairs["beta"], pairs["u"])
    G1, C1 = kra.fit_gram(ctx, b, snaps)
    os.environ["KP_WIDE_PANEL_MB"] = "1"                 # -> 1024-row panels: 3 panels, the last one 953 rows
    try:
        G3, C3 = kra.fit_gram(ctx, b, snaps)
    finally:
        del os.environ["KP_WIDE_PANEL_MB"]
    s = np.abs(G1).max()
    assert np.abs(G3 - G1).max() <= 1e-13 * s and np.abs(C3 - C1).max() <= 1e-13 * s
    assert (G3 == G3.T).all()
    Px, Py = ko.px_py(dic, pairs)
    Gr, Cr = ko.gram(Px, Py)
    assert np.abs(G3 - Gr).max() <= 1e-12 * s and np.abs(C3 - Cr).max() <= 1e-12 * s


def test_wide_bilinear_kronecker_form_equals_the_dense_products(ctx):
    """Bilinear wide dictionaries go through (m+1)(m+2)/2 weighted products of the N-wide panel of psi (block (a, b) of Px'Px is
    sum_k ut_a ut_b psi psi', ut = [1; u]); KP_WIDE_DENSE=1 keeps the dense products of the N(m+1)-wide panel.  Same Grams to
    rounding, over several panels with a ragged last one, exactly symmetric, and block (b, a) of C the bitwise copy of (a, b)."""
    pairs = synth_pairs(2500, 6, 3, seed=21)
    dic = ko.build_dictionary("bilinear", 6, 3, ["poly"], [4])
    b = make_basis(ctx, dic)
    N, m = b.N, 3
    assert b.W == N * (m + 1) == 840
    snaps = kra.Snapshots(ctx, pairs["alpha"], pairs["beta"], pairs["u"])
    os.environ["KP_WIDE_PANEL_MB"] = "2"                 # 1 248-row panels of psi: 3 panels
    try:
        Gk, Ck = kra.fit_gram(ctx, b, snaps)
        os.environ["KP_WIDE_DENSE"] = "1"
        Gd, Cd = kra.fit_gram(ctx, b, snaps)
    finally:
        os.environ.pop("KP_WIDE_DENSE", None)
        del os.environ["KP_WIDE_PANEL_MB"]
    s = np.abs(Gd).max()
    assert np.abs(Gk - Gd).max() <= 1e-13 * s and np.abs(Ck - Cd).max() <= 1e-13 * s
    assert (Gk == Gk.T).all()
    for a in range(m + 1):
        for c in range(a + 1, m + 1):
            assert (Ck[a * N:(a + 1) * N, c * N:(c + 1) * N] == Ck[c * N:(c + 1) * N, a * N:(a + 1) * N]).all()
            assert (Gk[a * N:(a + 1) * N, c * N:(c + 1) * N] == Gk[a * N:(a + 1) * N, c * N:(c + 1) * N].T).all()
    Px, Py = ko.px_py(dic, pairs)
    Gr, Cr = ko.gram(Px, Py)
    assert np.abs(Gk - Gr).max() <= 1e-12 * s and np.abs(Ck - Cr).max() <= 1e-12 * s


@pytest.mark.parametrize("W,nc,bs", [(513, 513, None), (600, 7, None), (777, 777, 128), (1000, 1000, None), (1000, 33, 352), (2940, 64, None)])
def test_wide_solve_random_spd(ctx, W, nc, bs):
    """kp_fit_solve beyond one workgroup's reach: blocked factorisation + substitution, ragged sizes, few and many
    right-hand sides, several block sizes."""
    rng = np.random.default_rng(W + nc)
    A = rng.standard_normal((W + 50, W))
    G = A.T @ A / W + 0.1 * np.eye(W)
    Cm = rng.standard_normal((W, nc))
    if bs:
        os.environ["KP_WIDE_BS"] = str(bs)
    try:
        K = ctx.fit_solve(G, Cm)
    finally:
        os.environ.pop("KP_WIDE_BS", None)
    Kr = np.linalg.solve(G, Cm)
    assert np.abs(K - Kr).max() <= 1e-10 * np.abs(Kr).max()
    assert np.abs(G @ K - Cm).max() <= 1e-11 * np.abs(G).sum(axis=1).max() * np.abs(K).max()


@pytest.mark.parametrize("mt,Ns", [("linear", 4000), ("bilinear", 7000)])
def test_wide_fit_fourier_on_six_states(ctx, mt, Ns):
    """SURVEY row a7 end to end: the 728-function fourier dictionary of `def_fourierLift` on six states, K = Px \\ Py."""
    pairs = synth_pairs(Ns, 6, 3, seed=5)
    dic = ko.build_dictionary(mt, 6, 3, ["fourier"], [1])
    b = make_basis(ctx, dic)
    assert b.W == (738 if mt == "linear" else 2940)
    snaps = kra.Snapshots(ctx, pairs["alpha"], pairs["beta"], pairs["u"])
    K = kra.fit(ctx, b, snaps)[0]
    assert ctx.last_rank() == b.W
    Px, Py = ko.px_py(dic, pairs)
    sv = np.linalg.svd(Px, compute_uv=False)
    cond = sv[0] / sv[-1]
    Kr = ko.koopman_ls(Px, Py)
    tol = max(1e-9, 50 * cond ** 2 * EPS)
    assert np.abs(K - Kr).max() <= tol * np.abs(Kr).max(), (cond, np.abs(K - Kr).max() / np.abs(Kr).max())
    # the pipelined entry (K_out == NULL) serves wide dictionaries synchronously: same K through kp_fit_get_K
    kra.fit(ctx, b, snaps, fetch=False)
    K2 = ctx.fit_result(0, b.W)
    assert (K2 == K).all()


def test_wide_rank_deficient_fit_returns_a_basic_solution_and_the_rank(ctx):
    """MATLAB's `\\` on a rank-deficient wide Px (Ksysid.m:1069): a repeated state makes products of its harmonics coincide
    (cos a sin b = sin a cos b, cos^2 + sin^2 = 1); the blocked pivoted Cholesky over many workgroups selects the column
    subset: same rank and residual as LAPACK's pivoted QR."""
    from test_gpu_fit import _pivoted_qr_basic_solution
    pairs = synth_pairs(3000, 6, 3, seed=3)
    pairs["alpha"][:, 5] = pairs["alpha"][:, 4]
    pairs["beta"][:, 5] = pairs["beta"][:, 4]
    dic = ko.build_dictionary("linear", 6, 3, ["fourier"], [1])
    b = make_basis(ctx, dic)
    snaps = kra.Snapshots(ctx, pairs["alpha"], pairs["beta"], pairs["u"])
    K = kra.fit(ctx, b, snaps)[0]
    Px, Py = ko.px_py(dic, pairs)
    Kq, r = _pivoted_qr_basic_solution(Px, Py)
    assert r < dic.W
    assert ctx.last_rank() == r
    assert (np.abs(K).sum(axis=1) == 0).sum() == dic.W - r
    res, resq = Px @ K - Py, Px @ Kq - Py
    assert np.abs(res - resq).max() < 1e-8 * max(1.0, np.abs(Py).max())


def test_mirror_with_a_delay_embedded_state_and_dim_red(ctx, golden):
    """Ksysid_setup.m lets the user pick delays = 1 with poly-3 and dim_red: nzeta = 15, 816 functions.  Rounds 1-4 took the
    `pca` of that dictionary on the host (too wide for the Gram kernels) and refused the BILINEAR model on it.  Now the
    covariance comes from the wide Gram path on the device (same principal axes as the host SVD of the lifted matrix), and
    the bilinear fit runs: its projected row goes through the lift kernel's 16-point form (the full lift of 64 points would
    not fit the LDS) and the wide Gram path - checked against the Grams of the device's own lifted rows formed on the host."""
    import warnings
    g = golden["arm_data"]
    lens = g["train_len"]; off = np.concatenate([[0], np.cumsum(lens)])
    train = [{"t": g["train_t"][a:b], "y": g["train_y"][a:b], "u": g["train_u"][a:b]} for a, b in zip(off[:-1], off[1:])]
    data = {"train": train, "val": [{"t": g["val_t"], "y": g["val_y"], "u": g["val_u"]}]}
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        kd = kra.Ksysid(data, ctx=ctx, model_type="linear", obs_type=["poly"], obs_degree=[3], snapshots=3000, lasso=[np.inf], delays=1, dim_red=True)
        kh = kra.Ksysid(data, ctx=ctx, model_type="linear", obs_type=["poly"], obs_degree=[3], snapshots=3000, lasso=[np.inf], delays=1, dim_red=True,
                        _pca_host=True)
    assert kd.basis_dev.nfull == 816 and kd.params["nzeta"] == 15
    assert kd.params["N"] == kh.params["N"]
    pd_, ph = kd.basis["pcs"], kh.basis["pcs"]
    # the leading principal axes agree (the trailing ones of the 99 % cut belong to nearly equal eigenvalues: compare the subspaces)
    k = min(10, pd_.shape[1])
    assert np.abs(np.abs(np.sum(pd_[:, :k] * ph[:, :k], axis=0)) - 1.0).max() < 1e-6
    sv = np.linalg.svd(pd_.T @ ph, compute_uv=False)
    assert sv.min() > 1 - 1e-6
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        kb = kra.Ksysid(data, ctx=ctx, model_type="bilinear", obs_type=["poly"], obs_degree=[3], snapshots=3000, lasso=[np.inf], delays=1, dim_red=True)
    sp = kb.snapshotPairs
    snaps = kra.Snapshots(ctx, sp["alpha"], sp["beta"], sp["u"])
    G, C = kra.fit_gram(ctx, kb.basis_dev, snaps)
    Px = kb.basis_dev.lift(F.LIFT_ROW, sp["alpha"], sp["u"]); Py = kb.basis_dev.lift(F.LIFT_ROW, sp["beta"], sp["u"])
    s = np.abs(Px.T @ Px).max()
    assert np.abs(G - Px.T @ Px).max() <= 1e-11 * s and np.abs(C - Px.T @ Py).max() <= 1e-11 * s
    snaps.close()


def test_lasso_on_a_wide_dictionary_meets_the_optimality_conditions(ctx):
    """solve_KoopmanQP (Ksysid.m:1095-1176) on the 728-function fourier dictionary (W = 738): the L1-constrained fit through the
    batched projected-gradient solver (the homotopy serves W <= 512).  Checked by the conditions of the convex QP: budget met,
    |G K - C| equal to one multiplier theta on the support and below it elsewhere."""
    p = synth_pairs(4000, 6, 3, seed=5)
    dic = ko.build_dictionary("linear", 6, 3, ["fourier"], [1])
    b = make_basis(ctx, dic)
    s = kra.Snapshots(ctx, p["alpha"], p["beta"], p["u"])
    Kls = kra.fit(ctx, b, s)[0]
    G, C = kra.fit_gram(ctx, b, s)
    l1 = np.abs(Kls).sum()
    for frac in (0.5, 0.1):
        t = frac * l1 / b.N
        K = kra.fit(ctx, b, s, lasso=[t])[0]
        assert abs(np.abs(K).sum() - t * b.N) <= 1e-9 * t * b.N
        g = G @ K - C
        supp = K != 0
        theta = np.abs(g[supp]).mean()
        assert np.abs(np.abs(g[supp]) - theta).max() <= 1e-5 * theta
        assert np.abs(g[~supp]).max() <= theta * (1 + 1e-6)
        assert (np.sign(K[supp]) == -np.sign(g[supp])).all()
    # an inactive budget returns the least-squares solution
    K = kra.fit(ctx, b, s, lasso=[2.0 * l1 / b.N])[0]
    assert np.array_equal(K, Kls)
