"""GPU parity tests (through the C ABI) of the EDMD fit: lift, fused lift+Gram, solve.
Oracle: oracle/koopman_oracle.py on the same inputs.  Tolerances (f64):
  lift            1e-13 abs (products of <= 13 factors in [-1,1]; different association)
  G, C            1e-12 relative to max|G| (sum of Ns products, different summation order)
  K               1e-9 relative to max|K| vs the QR/SVD least-squares oracle for cond(Px) <= ~1.5e3
                  (normal equations lose cond^2 * eps ~ 2e-10)
"""
import numpy as np
import pytest

import koopman_realizations_amd as kra
from koopman_realizations_amd import _ffi as F
from oracle import koopman_oracle as ko
from conftest import synth_pairs

pytestmark = pytest.mark.gpu


def make_basis(ctx, dic):
    blocks = []
    for kind, arg in dic.basis.blocks:
        if kind == "poly":
            blocks.append(("poly", arg[dic.basis.nvars:].astype(np.uint8)))
        else:
            blocks.append((kind, arg))
    return kra.Basis(ctx, dic.model_type, dic.nzeta, dic.m, blocks, dic.pcs)


CASES = [
    ("linear", 6, 3, ["poly"], [2], False),
    ("bilinear", 6, 3, ["poly"], [2], False),
    ("nonlinear", 6, 3, ["poly"], [2], False),
    ("bilinear", 6, 3, ["poly"], [3], False),      # BASELINE config 2 dictionary, W = 336
    ("linear", 1, 1, ["poly"], [13], False),       # rand-systems sweep, deepest linear
    ("bilinear", 1, 1, ["poly"], [6], False),
    ("nonlinear", 1, 1, ["poly"], [4], False),
    ("bilinear", 3, 2, ["poly", "fourier"], [2, 1], False),
    ("linear", 3, 2, ["gaussian", "poly"], [5, 2], False),
    ("bilinear", 6, 3, ["poly"], [3], True),       # dim_red (pcs projection)
    ("nonlinear", 6, 3, ["poly"], [2], True),
    ("linear", 6, 3, ["poly"], [3], True),         # example_sysid.m's first model: its Grams are sub-blocks of the bilinear ones
    ("linear", 3, 2, ["poly"], [3], True),
    ("linear", 2, 1, ["poly"], [2], True),
    ("linear", 10, 2, ["poly"], [3], True),        # 286 full functions (a delay-embedded state): kp_gram2 + congruence; no kernel projects that wide per pair
    ("nonlinear", 6, 3, ["poly"], [3], True),      # example_sysid.m's third model: 220 full functions on [zeta; u]
    ("bilinear", 3, 2, ["poly"], [3], False),                # two inputs: 6 Kronecker weights
    ("bilinear", 9, 3, ["poly"], [2], False),                # delay-embedded width (nzeta = 9), N = 55
    ("nonlinear", 6, 3, ["poly"], [3], False),               # N = W = 220: dense 16 x 16 tile kernel, 14 tiles per side
    ("linear", 6, 3, ["poly"], [4], False),                  # W = 213, monomials with 4 distinct variables
    ("linear", 2, 1, ["hermite"], [3], True),                # Ksysid.m:820-851
    ("bilinear", 2, 1, ["fourier_sparser"], [2], False),     # Ksysid.m:734-786
    ("nonlinear", 2, 1, ["poly", "hermite", "fourier_sparser"], [2, 2, 1], True),
    # fourier / gaussian blocks as table entries of the Kronecker kernel (kp_gram3_kernel<.,.,false,true>)
    ("bilinear", 3, 3, ["fourier"], [1], False),             # (1 + 2)^3 - 1 = 26 functions, products of <= 3 harmonics
    ("bilinear", 6, 3, ["gaussian"], [20], False),           # def_gaussianLift with 20 centres (Ksysid.m:790-817)
    ("bilinear", 2, 1, ["fourier", "gaussian", "poly"], [2, 7, 3], False),   # second harmonics by the recurrence, all three kinds
    # round 6 (the Kronecker kernel's in-loop power table and three-step lift): fourth powers; powers beyond 4 (served by kp_gram2);
    # the widest shape it admits - 15 raw rows per side (the table falls back to its dense row stride), 91 columns = 191 lift items
    ("bilinear", 3, 3, ["poly"], [4], False),
    ("bilinear", 2, 1, ["poly"], [5], False),
    ("bilinear", 12, 3, ["poly"], [2], False),
]


@pytest.mark.parametrize("mt,nz,m,types,degs,dim_red", CASES)
def test_lift_and_gram_parity(ctx, mt, nz, m, types, degs, dim_red):
    Ns = 1003  # ragged: not a multiple of the 8-snapshot tile
    pairs = synth_pairs(Ns, nz, m, seed=7)
    rng = np.random.default_rng(11)
    nv = nz + (m if mt == "nonlinear" else 0)
    centres = [rng.uniform(-1, 1, (nv, d)) for t, d in zip(types, degs) if t == "gaussian"]
    dic = ko.build_dictionary(mt, nz, m, types, degs, pairs, dim_red, centres)
    b = make_basis(ctx, dic)
    assert (b.nfull, b.N, b.W) == (dic.basis.nfull, dic.N, dic.W)
    V = np.hstack([pairs["alpha"], pairs["u"]]) if mt == "nonlinear" else pairs["alpha"]
    full = b.lift(F.LIFT_FULL, pairs["alpha"], pairs["u"] if mt == "nonlinear" else None)
    np.testing.assert_allclose(full, ko.lift_full(dic.basis, V), atol=1e-13, rtol=0)
    econ = b.lift(F.LIFT_ECON, pairs["alpha"], pairs["u"] if mt == "nonlinear" else None)
    np.testing.assert_allclose(econ, ko.econ_full(dic, V), atol=1e-13, rtol=0)
    Px, Py = ko.px_py(dic, pairs)
    np.testing.assert_allclose(b.lift(F.LIFT_ROW, pairs["alpha"], pairs["u"]), Px, atol=1e-13, rtol=0)
    snaps = kra.Snapshots(ctx, pairs["alpha"], pairs["beta"], pairs["u"])
    G, C = kra.fit_gram(ctx, b, snaps)
    Gr, Cr = ko.gram(Px, Py)
    scale = np.abs(Gr).max()
    assert np.abs(G - Gr).max() <= 1e-12 * scale
    assert np.abs(C - Cr).max() <= 1e-12 * scale
    assert (G == G.T).all()          # both triangles written from the same accumulator
    G2, C2 = kra.fit_gram(ctx, b, snaps)
    assert (G2 == G).all() and (C2 == C).all()   # fixed-order reduction: bitwise reproducible


@pytest.mark.parametrize("nz,m,deg,k", [(2, 2, 3, 8), (2, 1, 3, 9), (2, 2, 3, 10), (3, 1, 2, 7), (1, 3, 4, 4), (4, 2, 2, 13)])
def test_gram_with_as_many_principal_axes_as_functions(ctx, nz, m, deg, k):
    """dim_red dictionaries whose econ layout [zeta | k axes | 1] is as wide as (or wider than) the full dictionary:
    the in-kernel projection writes the axes over the column that held the full dictionary's constant."""
    pairs = synth_pairs(1003, nz, m, seed=21)
    dic = ko.build_dictionary("bilinear", nz, m, ["poly"], [deg])
    assert k <= dic.basis.nfull
    dic.pcs = np.linalg.qr(np.random.default_rng(k).standard_normal((dic.basis.nfull, dic.basis.nfull)))[0][:, :k].copy()
    b = make_basis(ctx, dic)
    assert b.N == nz + k + 1
    Px, Py = ko.px_py(dic, pairs)
    np.testing.assert_allclose(b.lift(F.LIFT_ROW, pairs["alpha"], pairs["u"]), Px, atol=1e-13, rtol=0)
    snaps = kra.Snapshots(ctx, pairs["alpha"], pairs["beta"], pairs["u"])
    G, C = kra.fit_gram(ctx, b, snaps)
    Gr, Cr = ko.gram(Px, Py)
    assert np.abs(G - Gr).max() <= 1e-12 * np.abs(Gr).max() and np.abs(C - Cr).max() <= 1e-12 * np.abs(Gr).max()


@pytest.mark.parametrize("Ns", [1, 7, 8, 9, 64, 2048])
def test_gram_edge_sizes(ctx, Ns):
    pairs = synth_pairs(Ns, 2, 1, seed=3)
    dic = ko.build_dictionary("bilinear", 2, 1, ["poly"], [2])
    b = make_basis(ctx, dic)
    snaps = kra.Snapshots(ctx, pairs["alpha"], pairs["beta"], pairs["u"])
    G, C = kra.fit_gram(ctx, b, snaps)
    Px, Py = ko.px_py(dic, pairs)
    Gr, Cr = ko.gram(Px, Py)
    assert np.abs(G - Gr).max() <= 1e-12 * max(1.0, np.abs(Gr).max())
    assert np.abs(C - Cr).max() <= 1e-12 * max(1.0, np.abs(Gr).max())


@pytest.mark.parametrize("nz,m,deg,Ns", [(1, 1, 6, 9000), (2, 1, 2, 9008), (2, 1, 6, 20000), (6, 3, 2, 20000), (6, 3, 3, 20001)])
def test_gram_small_dictionaries_many_splits(ctx, nz, m, deg, Ns):
    """Small dictionaries put few quads on a wave and several tiles on every snapshot split: exercises the
    lift pipeline of the Kronecker kernel with 1..3 quad steps per tile and odd/even tile counts."""
    pairs = synth_pairs(Ns, nz, m, seed=9)
    dic = ko.build_dictionary("bilinear", nz, m, ["poly"], [deg])
    b = make_basis(ctx, dic)
    snaps = kra.Snapshots(ctx, pairs["alpha"], pairs["beta"], pairs["u"])
    G, C = kra.fit_gram(ctx, b, snaps)
    Px, Py = ko.px_py(dic, pairs)
    Gr, Cr = ko.gram(Px, Py)
    assert np.abs(G - Gr).max() <= 1e-12 * np.abs(Gr).max()
    assert np.abs(C - Cr).max() <= 1e-12 * np.abs(Gr).max()


def test_gram_linearity_in_snapshots(ctx):
    """Size-independent property: Gram over a concatenation = sum of the Grams."""
    dic = ko.build_dictionary("bilinear", 6, 3, ["poly"], [3])
    b = make_basis(ctx, dic)
    p = synth_pairs(100000, seed=5)                      # BASELINE config 2 size
    sA = kra.Snapshots(ctx, p["alpha"], p["beta"], p["u"])
    h = 37123
    s1 = kra.Snapshots(ctx, p["alpha"][:h], p["beta"][:h], p["u"][:h])
    s2 = kra.Snapshots(ctx, p["alpha"][h:], p["beta"][h:], p["u"][h:])
    G, C = kra.fit_gram(ctx, b, sA)
    G1, C1 = kra.fit_gram(ctx, b, s1)
    G2, C2 = kra.fit_gram(ctx, b, s2)
    assert np.abs(G - (G1 + G2)).max() <= 1e-12 * np.abs(G).max()
    assert np.abs(C - (C1 + C2)).max() <= 1e-12 * np.abs(G).max()
    # constant observable: G[N-1, N-1] counts the snapshots exactly
    assert G[dic.N - 1, dic.N - 1] == 100000.0


def test_solve_parity_at_every_panel_count(ctx):
    """Every padded order 16 .. 352 of the left-looking Cholesky (its tile dealing, the overlapped diagonal block and the
    late-panel (tile pair, k-step) ranges change with the number of panels) and of the right-looking block substitution,
    plus the orders around the switch to round 1's kernels (> 352)."""
    rng = np.random.default_rng(99)
    for W in list(range(9, 353, 16)) + [352, 353, 368, 400]:
        P = rng.standard_normal((3 * W + 20, W)); Y = rng.standard_normal((3 * W + 20, 5))
        G, C = P.T @ P, P.T @ Y
        K = ctx.fit_solve(G, C)
        Kref = np.linalg.solve(G, C)
        assert ctx.last_rank() == W
        assert np.abs(K - Kref).max() <= 1e-10 * np.abs(Kref).max(), W


@pytest.mark.parametrize("W,nc", [(5, 5), (16, 16), (17, 3), (32, 1), (33, 7), (48, 48), (136, 136), (336, 336), (500, 16)])
def test_solve_parity(ctx, W, nc):
    rng = np.random.default_rng(W)
    P = rng.standard_normal((4 * W + 10, W)); Y = rng.standard_normal((4 * W + 10, nc))
    G, C = P.T @ P, P.T @ Y
    K = ctx.fit_solve(G, C)
    Kref = np.linalg.lstsq(P, Y, rcond=None)[0]
    assert np.abs(K - Kref).max() <= 1e-11 * np.abs(Kref).max()


def _pivoted_qr_basic_solution(Px, Py):
    """MATLAB's `\\` on a rank-deficient rectangular system (Ksysid.m:1069): Householder QR with column pivoting, rank from
    the diagonal of R (tolerance max(size) eps |R_11|), basic solution over the first r pivot columns."""
    import scipy.linalg as sla
    Q, R, piv = sla.qr(Px, mode="economic", pivoting=True)
    d = np.abs(np.diag(R))
    r = int((d > max(Px.shape) * np.finfo(float).eps * d[0]).sum())
    K = np.zeros((Px.shape[1], Py.shape[1]))
    K[piv[:r]] = sla.solve_triangular(R[:r, :r], Q[:, :r].T @ Py)
    return K, r


def test_solve_of_rank_deficient_systems_returns_a_basic_solution_and_the_rank(ctx):
    """kp_fit_solve on singular Gram matrices: no error, a basic solution (zero rows outside the selected columns), the
    rank, and the residual of LAPACK's pivoted-QR basic solution."""
    rng = np.random.default_rng(0)
    P = rng.standard_normal((50, 8)); Y = rng.standard_normal((50, 3))
    P[:, 7] = P[:, 0] + P[:, 1]
    K = ctx.fit_solve(P.T @ P, P.T @ Y)
    assert ctx.last_rank() == 7
    assert (np.abs(K).sum(axis=1) == 0).sum() == 1                       # one column of Px is left out
    Kq, r = _pivoted_qr_basic_solution(P, Y)
    assert r == 7
    assert np.abs((P @ K - Y) - (P @ Kq - Y)).max() < 1e-10
    # a full-rank system right after it reports its full rank
    P2 = rng.standard_normal((60, 8))
    ctx.fit_solve(P2.T @ P2, P2.T @ rng.standard_normal((60, 2)))
    assert ctx.last_rank() == 8
    # an indefinite "Gram" matrix: the pivoting stops at the first non-positive pivot
    K = ctx.fit_solve(np.diag([1.0, 2.0, 4.0, -1.0]), np.eye(4))
    assert ctx.last_rank() == 3 and np.allclose(K, np.diag([1.0, 0.5, 0.25, 0.0]))


@pytest.mark.parametrize("mt,deg,rank", [("bilinear", 3, 252), ("linear", 2, 28), ("bilinear", 2, 100)])
def test_fit_arm_data_without_dim_red_is_rank_deficient_like_the_reference(ctx, arm, mt, deg, rank):
    """The arm's marker coordinates satisfy link-length identities, so the full polynomial dictionaries are exactly rank
    deficient (poly-3 bilinear 252 of 336, poly-2 lift 25 of 28).  MATLAB's `\\` (Ksysid.m:1069) returns a
    basic solution from pivoted QR; the device returns one from pivoted Cholesky: same rank, same residual (every
    basic solution over a column subset spanning range(Px) has the residual of the projection)."""
    p = arm["pairs"]
    dic = ko.build_dictionary(mt, 6, 3, ["poly"], [deg])
    b = make_basis(ctx, dic)
    snaps = kra.Snapshots(ctx, p["alpha"], p["beta"], p["u"])
    K = kra.fit(ctx, b, snaps)[0]
    Px, Py = ko.px_py(dic, p)
    Kq, r = _pivoted_qr_basic_solution(Px, Py)
    assert r == rank
    assert ctx.last_rank() == rank
    assert (np.abs(K).sum(axis=1) == 0).sum() == dic.W - rank
    res, resq = Px @ K - Py, Px @ Kq - Py
    assert np.abs(res - resq).max() < 1e-8 * max(1.0, np.abs(Py).max())


@pytest.mark.parametrize("mt,deg,dim_red", [("bilinear", 3, True), ("linear", 3, True), ("nonlinear", 3, True), ("linear", 2, True)])
def test_fit_arm_data_matches_oracle_lstsq(ctx, arm, mt, deg, dim_red):
    """example_sysid.m configuration on the shipped arm data (11 999 pairs, dim_red)."""
    pairs = arm["pairs"]
    dic = ko.build_dictionary(mt, 6, 3, ["poly"], [deg], pairs, dim_red)
    b = make_basis(ctx, dic)
    snaps = kra.Snapshots(ctx, pairs["alpha"], pairs["beta"], pairs["u"])
    K = kra.fit(ctx, b, snaps)[0]
    Px, Py = ko.px_py(dic, pairs)
    Kref = ko.koopman_ls(Px, Py)
    cond = np.linalg.cond(Px)
    tol = max(1e-9, 50 * cond ** 2 * 2.2e-16)
    assert np.abs(K - Kref).max() <= tol * np.abs(Kref).max(), (cond, np.abs(K - Kref).max() / np.abs(Kref).max())


def test_fit_synthetic_config2(ctx):
    """BASELINE config 2: bilinear poly-3, 1e5 synthetic snapshots, W = 336."""
    p = synth_pairs(100000, seed=0)
    dic = ko.build_dictionary("bilinear", 6, 3, ["poly"], [3])
    b = make_basis(ctx, dic)
    snaps = kra.Snapshots(ctx, p["alpha"], p["beta"], p["u"])
    K = kra.fit(ctx, b, snaps)[0]
    Px, Py = ko.px_py(dic, p)
    Kref = ko.koopman_ls(Px, Py)
    assert np.abs(K - Kref).max() <= 1e-9 * np.abs(Kref).max()


def test_async_fit_pipeline_matches_synchronous_result(ctx):
    """kp_fit with K_out = NULL is asynchronous (solve of fit i overlaps the Gram of fit i+1 on a
    second stream); results and error reporting must equal the synchronous path."""
    dic = ko.build_dictionary("bilinear", 6, 3, ["poly"], [3])
    b = make_basis(ctx, dic)
    sets = [synth_pairs(20000, seed=20 + i) for i in range(3)] + [synth_pairs(23000, seed=30)]
    snaps = [kra.Snapshots(ctx, p["alpha"], p["beta"], p["u"]) for p in sets]
    Ksync = [kra.fit(ctx, b, s)[0] for s in snaps]
    W = b.W
    tol = lambda Kref: 1e-11 * np.abs(Kref).max()
    for s, Kref in zip(snaps, Ksync):           # one fit per batch: kp_fit_get_K synchronises, which closes the batch
        kra.fit(ctx, b, s, fetch=False)
        # the async path leaves CUs free for the overlapped solve => different snapshot split, different
        # (still fixed) summation order of the partial Grams: equal to rounding, not bitwise
        assert np.abs(ctx.fit_result(0, W) - Kref).max() <= tol(Kref)
    # a batch of pipelined fits: EVERY K stays retrievable (ring of result slots), in issue order
    for s in snaps[:3]:
        kra.fit(ctx, b, s, fetch=False)
    ctx.synchronize()
    for i in range(3):
        assert np.abs(ctx.fit_result(i, W) - Ksync[i]).max() <= tol(Ksync[i])
    Kst = ctx.fit_results(0, 3, W)               # the same three through the page-locked stack (column-major blocks)
    for i in range(3):
        assert np.array_equal(Kst[i].T, ctx.fit_result(i, W))
    with pytest.raises(kra.KoopmanHipError):
        ctx.fit_result(3, W)                     # not part of the batch
    # ring shorter than the batch: the oldest results are gone and say so
    ctx.fit_async_slots(2)
    for s in snaps[:3]:
        kra.fit(ctx, b, s, fetch=False)
    with pytest.raises(kra.KoopmanHipError) as e:
        ctx.fit_result(0, W)
    assert e.value.code == F.KP_ERR_ARG
    for i in (1, 2):
        assert np.abs(ctx.fit_result(i, W) - Ksync[i]).max() <= tol(Ksync[i])
    ctx.fit_async_slots(128)
    # another snapshot count while fits are in flight: the pipeline drains first (shared partial / G|C buffers are
    # sized per call), which closes the batch; the new fit is number 0 of the next one
    kra.fit(ctx, b, snaps[0], fetch=False)
    kra.fit(ctx, b, snaps[3], fetch=False)
    kra.fit(ctx, b, snaps[3], fetch=False)
    assert np.abs(ctx.fit_result(0, W) - Ksync[3]).max() <= tol(Ksync[3])
    assert np.abs(ctx.fit_result(1, W) - Ksync[3]).max() <= tol(Ksync[3])
    # deferred failure: a rank-deficient dictionary is reported by kp_synchronize
    z = np.zeros((64, 6)); u = np.random.default_rng(0).uniform(-1, 1, (64, 3))
    bad = kra.Snapshots(ctx, z, z, u)           # all-zero states: Gram is singular
    kra.fit(ctx, b, bad, fetch=False)
    with pytest.raises(kra.KoopmanHipError) as e:
        ctx.synchronize()
    assert e.value.code == F.KP_ERR_NOT_SPD
    ctx.synchronize()                            # the sticky flag was cleared


def test_queued_fits_survive_a_wider_dictionary_arriving_without_a_synchronise(ctx):
    """Deferred, batched solves: the [G | C] ring holds queued Gram pairs that are not solved yet.  A fit with a much wider
    dictionary (ring buffer has to grow) issued WITHOUT synchronising must first drain the queue - not reallocate the ring
    under it: the three narrow fits' results of the closed batch are lost by design (new batch), but the wide fit and a
    following narrow batch must be right and no spurious NOT_SPD may be reported."""
    small = make_basis(ctx, ko.build_dictionary("linear", 6, 3, ["poly"], [1]))          # W = 10
    wide = make_basis(ctx, ko.build_dictionary("bilinear", 6, 3, ["poly"], [3]))         # W = 336 > 6 x 10
    p = synth_pairs(6000, seed=41)
    snaps = kra.Snapshots(ctx, p["alpha"], p["beta"], p["u"])
    Ks_ref = kra.fit(ctx, small, snaps)[0]
    Kw_ref = kra.fit(ctx, wide, snaps)[0]
    ctx2 = kra.Context(0)                       # fresh context: its ring starts at the narrow size
    try:
        small2 = make_basis(ctx2, ko.build_dictionary("linear", 6, 3, ["poly"], [1]))
        wide2 = make_basis(ctx2, ko.build_dictionary("bilinear", 6, 3, ["poly"], [3]))
        snaps2 = kra.Snapshots(ctx2, p["alpha"], p["beta"], p["u"])
        for _ in range(3):
            kra.fit(ctx2, small2, snaps2, fetch=False)
        kra.fit(ctx2, wide2, snaps2, fetch=False)          # no synchronise in between
        assert np.abs(ctx2.fit_result(0, wide2.W) - Kw_ref).max() <= 1e-10 * np.abs(Kw_ref).max()
        for _ in range(3):
            kra.fit(ctx2, small2, snaps2, fetch=False)
        ctx2.synchronize()                                  # would raise NOT_SPD if a queued pair had been clobbered
        for i in range(3):
            assert np.abs(ctx2.fit_result(i, small2.W) - Ks_ref).max() <= 1e-10 * np.abs(Ks_ref).max()
    finally:
        ctx2.close()


def test_snapshot_objects_refilled_from_the_host_while_fits_are_in_flight(ctx):
    """kp_snapshots_update: new pairs into an existing object through the pinned staging ring and the copy stream.  Two
    objects filled alternately while the fits of the other one are queued: every K equals the fit of a freshly uploaded
    object; the host arrays may be overwritten as soon as update() returns; the snapshot count may shrink and grow."""
    dic = ko.build_dictionary("bilinear", 6, 3, ["poly"], [3])
    b = make_basis(ctx, dic)
    W = b.W
    sets = [synth_pairs(20000, seed=40 + i) for i in range(5)]
    Kref = []
    for p in sets:
        s = kra.Snapshots(ctx, p["alpha"], p["beta"], p["u"])
        kra.fit(ctx, b, s, fetch=False)
        Kref.append(ctx.fit_result(0, W).copy())
        s.close()
    ring = [kra.Snapshots(ctx, sets[0]["alpha"], sets[0]["beta"], sets[0]["u"]) for _ in range(2)]
    order = [1, 2, 3, 4, 0, 2, 4, 1]
    for i, k in enumerate(order):
        a, be, u = (np.asfortranarray(sets[k][f]).copy(order="F") for f in ("alpha", "beta", "u"))
        ring[i % 2].update(a, be, u)
        a[:] = np.nan; be[:] = np.nan; u[:] = np.nan          # the caller's arrays are free again
        kra.fit(ctx, b, ring[i % 2], fetch=False)
    ctx.synchronize()
    for i, k in enumerate(order):
        assert np.abs(ctx.fit_result(i, W) - Kref[k]).max() <= 1e-13 * np.abs(Kref[k]).max(), (i, k)
    # another snapshot count: fewer rows reuse the arrays, more rows reallocate them (drains the pipeline)
    small, big = synth_pairs(7001, seed=50), synth_pairs(26003, seed=51)
    for p in (small, big, small):
        Ks = kra.fit(ctx, b, kra.Snapshots(ctx, p["alpha"], p["beta"], p["u"]))[0]
        ring[0].update(p["alpha"], p["beta"], p["u"])
        assert ring[0].Ns == p["alpha"].shape[0]
        Ku = kra.fit(ctx, b, ring[0])[0]
        assert np.array_equal(Ku, Ks)
        G, Cm = kra.fit_gram(ctx, b, ring[0])
        Px, Py = ko.px_py(dic, p)
        assert np.abs(G - Px.T @ Px).max() <= 1e-11 * np.abs(G).max() and np.abs(Cm - Px.T @ Py).max() <= 1e-11 * np.abs(Cm).max()
    with pytest.raises(ValueError):
        ring[0].update(small["alpha"][:, :5], small["beta"][:, :5], small["u"])
    # an object that was never refilled keeps working beside the streaming ones (no events recorded for it)
    s0 = kra.Snapshots(ctx, sets[0]["alpha"], sets[0]["beta"], sets[0]["u"])
    kra.fit(ctx, b, s0, fetch=False)
    assert np.abs(ctx.fit_result(0, W) - Kref[0]).max() <= 1e-13 * np.abs(Kref[0]).max()


_POOL_SCRIPT = r"""
import sys, numpy as np
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, sys.argv[1] + "/tests")
import koopman_realizations_amd as kra
from oracle import koopman_oracle as ko
from conftest import synth_pairs
from test_gpu_fit import make_basis
ctx = kra.Context(0)
dic = ko.build_dictionary("bilinear", 6, 3, ["poly"], [2])
b = make_basis(ctx, dic)
sets = [synth_pairs(30011 + 977 * i, seed=60 + i) for i in range(4)]
ring = [kra.Snapshots(ctx, sets[0]["alpha"], sets[0]["beta"], sets[0]["u"]) for _ in range(2)]
for rep in range(3):
    for i, p in enumerate(sets):
        s = ring[i % 2].update(p["alpha"], p["beta"], p["u"])
        G, C = kra.fit_gram(ctx, b, s)
        Px, Py = ko.px_py(dic, p)
        assert np.abs(G - Px.T @ Px).max() <= 1e-11 * np.abs(G).max() and np.abs(C - Px.T @ Py).max() <= 1e-11 * np.abs(C).max(), (rep, i)
print("POOL_OK")
"""


def test_staged_upload_with_copy_threads_and_small_chunks():
    """The chunked host -> HBM path with its worker threads forced on (4 threads, 64 KB chunks, no size threshold: ~60
    chunks per refill handed between threads) - the configuration large snapshot matrices use - against the oracle's Grams."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, KP_COPY_THREADS="4", KP_COPY_CHUNK_KB="64", KP_COPY_POOL_MIN_MB="0")
    r = subprocess.run([sys.executable, "-c", _POOL_SCRIPT, root], capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0 and "POOL_OK" in r.stdout, r.stdout[-1500:] + r.stderr[-1500:]


_LONE_FIT_SCRIPT = r"""
import sys
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, sys.argv[1] + "/tests")
import numpy as np
import koopman_realizations_amd as kra
from conftest import synth_pairs
from oracle import koopman_oracle as ko
n_fits, n_slots = int(sys.argv[2]), int(sys.argv[3])
ctx = kra.Context(0)
dic = ko.build_dictionary("bilinear", 3, 2, ["poly"], [2])
b = kra.Basis(ctx, "bilinear", 3, 2, [("poly", dic.basis.blocks[0][1][3:].astype(np.uint8))])
sets = [synth_pairs(3000, 3, 2, seed=70 + i) for i in range(3)]
snaps = [kra.Snapshots(ctx, p["alpha"], p["beta"], p["u"]) for p in sets]
Kref = [kra.fit(ctx, b, s)[0] for s in snaps]
ctx.fit_async_slots(n_slots)
for q in range(n_fits):
    kra.fit(ctx, b, snaps[q % 3], fetch=False)
ctx.synchronize()
for q in range(max(0, n_fits - n_slots), n_fits):
    K = ctx.fit_result(q, b.W)
    assert np.abs(K - Kref[q % 3]).max() <= 1e-11 * np.abs(Kref[q % 3]).max(), q
print("LONE_OK")
"""


@pytest.mark.parametrize("n_fits,n_slots,env", [(129, 200, {}), (257, 300, {}), (7, 128, {"KP_SOLVE_BATCH": "1"}), (5, 64, {"KP_SOLVE_BATCH": "3"})])
def test_a_lone_queued_fit_lands_in_its_own_ring_slot(n_fits, n_slots, env):
    """A flush of exactly ONE queued fit at a non-zero ring position (129 pipelined fits with a 128-fit solve batch; a solve
    batch of 1) takes the one-system TRSM, which writes K in place: it must apply the ring offset itself - three distinct
    snapshot sets in rotation, so a K written to slot 0 (or left stale) is seen."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-c", _LONE_FIT_SCRIPT, root, str(n_fits), str(n_slots)], capture_output=True, text=True, timeout=300,
                       env=dict(os.environ, **env))
    assert r.returncode == 0 and "LONE_OK" in r.stdout, r.stdout[-1500:] + r.stderr[-1500:]


_GRAM6_SCRIPT = r"""
import sys
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, sys.argv[1] + "/tests")
import numpy as np
import koopman_realizations_amd as kra
from conftest import synth_pairs
from oracle import koopman_oracle as ko
ctx = kra.Context(0)
for Ns in (20001, 37):
    p = synth_pairs(Ns, seed=3)
    dic = ko.build_dictionary("bilinear", 6, 3, ["poly"], [3])
    b = kra.Basis(ctx, "bilinear", 6, 3, [("poly", dic.basis.blocks[0][1][6:].astype(np.uint8))])
    s = kra.Snapshots(ctx, p["alpha"], p["beta"], p["u"])
    G, C = kra.fit_gram(ctx, b, s)
    Px, Py = ko.px_py(dic, p)
    assert np.abs(G - Px.T @ Px).max() <= 1e-12 * np.abs(G).max() and np.array_equal(G, G.T)
    assert np.abs(C - Px.T @ Py).max() <= 1e-12 * np.abs(G).max()
    G2, C2 = kra.fit_gram(ctx, b, s)
    assert np.array_equal(G, G2) and np.array_equal(C, C2)          # fixed summation order: bitwise reproducible
print("GRAM6_OK")
"""


def test_eight_wave_kronecker_kernel_experiment_is_exact():
    """KP_GRAM6=1 selects kp_gram6_kernel (kp_gram6.hip: one 8-wave workgroup per CU, the ten weighted copies of psi_x written
    to LDS by the lift, no multiply in the MFMA loop) for the W = 336 dictionary.  It is an opt-in experiment - the compiler
    spills 66 dwords of its 7-quad waves and it runs at 0.565 ms against kp_gram3_kernel's 0.395 - but it is exact: G, C
    against the oracle to 1e-12, symmetric, reproducible, with a ragged tail and with fewer pairs than one tile row."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-c", _GRAM6_SCRIPT, root], capture_output=True, text=True, timeout=300, env=dict(os.environ, KP_GRAM6="1"))
    assert r.returncode == 0 and "GRAM6_OK" in r.stdout, r.stdout[-1500:] + r.stderr[-1500:]


@pytest.mark.parametrize("mt,deg,steps,tol", [("bilinear", 2, 1, 1e-10), ("linear", 2, 1, 1e-10), ("bilinear", 3, 1, 1e-12), ("nonlinear", 2, 2, 1e-10)])
def test_fit_refine_reaches_qr_accuracy(ctx, arm, mt, deg, steps, tol):
    """kp_fit_refine: K += G^-1 Px'(Py - Px K) with the residual taken from the lifted rows.  On the arm data with
    dim_red the poly-2 dictionaries have cond(Px) ~ 1e5: the normal equations alone are good to ~5e-7, one step gives
    the QR / SVD least-squares solution (MATLAB's `\\`, Ksysid.m:1069) to 1e-10 or better."""
    p = arm["pairs"]
    dic = ko.build_dictionary(mt, 6, 3, ["poly"], [deg], p, True)
    basis = make_basis(ctx, dic)
    snaps = kra.Snapshots(ctx, p["alpha"], p["beta"], p["u"])
    K0 = kra.fit(ctx, basis, snaps)[0]
    K1 = kra.fit_refine(ctx, basis, snaps, K0, steps)
    Px, Py = ko.px_py(dic, p)
    Kq = np.linalg.lstsq(Px, Py, rcond=None)[0]
    scale = max(1.0, np.abs(Kq).max())
    assert np.abs(K1 - Kq).max() <= tol * scale
    assert np.abs(K1 - Kq).max() <= np.abs(K0 - Kq).max()          # never worse than the unrefined solution


@pytest.mark.parametrize("mt,Ns", [("bilinear", 1_000_000), ("linear", 1_000_000), ("nonlinear", 400_000)])
def test_large_snapshot_counts_by_size_independent_properties(ctx, mt, Ns):
    """Scaling sizes of SURVEY 8(d) (1e6 pairs) where the oracle is too slow to run: (1) Gram over a concatenation = sum
    of the Grams of the parts; (2) exact snapshot count in the constant observable; (3) the fit recovers the generating
    model: the data are produced by a random K_true in the lifted space restricted to the state rows, so the first nzeta
    columns of the fitted K reproduce it."""
    dic = ko.build_dictionary(mt, 6, 3, ["poly"], [3 if mt != "nonlinear" else 2])
    b = make_basis(ctx, dic)
    rng = np.random.default_rng(11)
    alpha = rng.uniform(-1, 1, (Ns, 6)); u = rng.uniform(-1, 1, (Ns, 3))
    # exact linear-in-the-dictionary dynamics for the state rows: beta = Px @ Kt (no noise), Kt small
    probe = ko.lift_rows(dic, alpha[:2000], u[:2000])
    Kt = 0.02 * rng.standard_normal((probe.shape[1], 6)) / np.sqrt(probe.shape[1])
    chunks = [slice(i, min(i + 200_000, Ns)) for i in range(0, Ns, 200_000)]
    beta = np.vstack([ko.lift_rows(dic, alpha[c], u[c]) @ Kt for c in chunks])
    sA = kra.Snapshots(ctx, alpha, beta, u)
    G, C = kra.fit_gram(ctx, b, sA)
    Gs = np.zeros_like(G); Cs = np.zeros_like(C)
    for c in chunks:
        sc = kra.Snapshots(ctx, alpha[c], beta[c], u[c])
        g1, c1 = kra.fit_gram(ctx, b, sc)
        Gs += g1; Cs += c1
        sc.close()
    assert np.abs(G - Gs).max() <= 1e-12 * np.abs(G).max()
    assert np.abs(C - Cs).max() <= 1e-12 * np.abs(G).max()
    assert G[dic.N - 1, dic.N - 1] == float(Ns)
    assert np.array_equal(G, G.T)
    K = kra.fit(ctx, b, sA)[0]
    assert np.abs(K[:, :6] - Kt).max() < 1e-9
    sA.close()


_PRELIFT_SCRIPT = """
import sys, numpy as np
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, sys.argv[1] + '/tests')
import koopman_realizations_amd as kra
from conftest import synth_pairs
ctx = kra.Context(0)
pcs = np.linalg.qr(np.random.default_rng(0).standard_normal((84, 27)))[0]
out = {}
for Ns in (100000, 4099, 5):
    p = synth_pairs(Ns, seed=2)
    b = kra.Basis(ctx, "bilinear", 6, 3, [("poly", kra.poly_exponent_table(6, 3)[6:])], pcs)
    s = kra.Snapshots(ctx, p["alpha"], p["beta"], p["u"])
    G, C = kra.fit_gram(ctx, b, s)
    out["G%d" % Ns] = G; out["C%d" % Ns] = C
    # fourier / gaussian blocks: their table entries once per snapshot (kp_gram3_prelift_ext_kernel) or inside the Gram kernel
    bg = kra.Basis(ctx, "bilinear", 6, 3, [("gaussian", np.random.default_rng(3).uniform(-1, 1, (6, 20)))])
    G, C = kra.fit_gram(ctx, bg, s)
    out["Gg%d" % Ns] = G; out["Cg%d" % Ns] = C
    p3 = synth_pairs(Ns, 3, 2, seed=4)
    s3 = kra.Snapshots(ctx, p3["alpha"], p3["beta"], p3["u"])
    bf = kra.Basis(ctx, "bilinear", 3, 2, [("fourier", 1), ("poly", kra.poly_exponent_table(3, 2)[3:])])
    G, C = kra.fit_gram(ctx, bf, s3)
    out["Gf%d" % Ns] = G; out["Cf%d" % Ns] = C
    # the widest rows the lifted-tile loader takes (N = 55: 14 column groups, 496 of its 512 pieces) and the first that it does not (N = 59)
    p4 = synth_pairs(Ns, 4, 2, seed=6)
    s4 = kra.Snapshots(ctx, p4["alpha"], p4["beta"], p4["u"])
    for ng in (20, 24):
        bw = kra.Basis(ctx, "bilinear", 4, 2, [("poly", kra.poly_exponent_table(4, 3)[4:]), ("gaussian", np.random.default_rng(ng).uniform(-1, 1, (4, ng)))])
        G, C = kra.fit_gram(ctx, bw, s4)
        out["Gw%d_%d" % (ng, Ns)] = G; out["Cw%d_%d" % (ng, Ns)] = C
np.savez(sys.argv[2], **out)
"""


def test_econ_lift_once_per_snapshot_equals_the_in_kernel_projection():
    """dim_red bilinear dictionary (N = 34, W = 136, the shape of example_sysid.m): by default the econ lift [zeta; pcs' psi; 1]
    (Ksysid.m:1594-1618) is formed once per snapshot by kp_gram3_prelift_kernel and the Kronecker kernel loads lifted tiles;
    KP_GRAM3_NO_PRELIFT=1 (read once per process) keeps the projection inside every workgroup of the Gram kernel.  Same G, C to
    rounding, at 1e5 pairs, with a ragged tail (4099) and with fewer pairs than one tile (5).  The same for a gaussian dictionary
    (20 centres on 6 states, m = 3) and a fourier + poly-2 one (3 states, m = 2), whose table entries - sincospi, exp - move out of
    the Gram kernel the same way (kp_gram3_prelift_ext_kernel)."""
    import os
    import subprocess
    import sys
    import tempfile
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    res = {}
    with tempfile.TemporaryDirectory() as td:
        for name, env in (("pre", {"KP_GRAM3_PRELIFT_MIN_NS": "0"}), ("proj", {"KP_GRAM3_NO_PRELIFT": "1"})):
            f = os.path.join(td, name + ".npz")
            r = subprocess.run([sys.executable, "-c", _PRELIFT_SCRIPT, root, f], capture_output=True, text=True, timeout=600, env=dict(os.environ, **env))
            assert r.returncode == 0, r.stderr[-1500:]
            res[name] = dict(np.load(f))
    for k in res["pre"]:
        a, b = res["pre"][k], res["proj"][k]
        assert np.isfinite(a).all()
        assert np.abs(a - b).max() <= 1e-12 * np.abs(b).max(), k


_CONG_SCRIPT = """
import sys, os, numpy as np
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, sys.argv[1] + '/tests')
import koopman_realizations_amd as kra
g = np.load(os.path.join(sys.argv[1], 'tests', 'golden', 'arm_data.npz'))
lens = g['train_len']; off = np.concatenate([[0], np.cumsum(lens)])
train = [{'t': g['train_t'][a:b], 'y': g['train_y'][a:b], 'u': g['train_u'][a:b]} for a, b in zip(off[:-1], off[1:])]
data = {'train': train, 'val': [{'t': g['val_t'], 'y': g['val_y'], 'u': g['val_u']}]}
ctx = kra.Context(0)
out = {}
for mt in ('linear', 'nonlinear'):
    ks = kra.Ksysid(data, ctx=ctx, model_type=mt, obs_type=['poly'], obs_degree=[3], snapshots=np.inf, lasso=[np.inf], delays=0, dim_red=True)
    sp = ks.snapshotPairs
    s = ks._resident_snapshots(sp['alpha'], sp['beta'], sp['u'])
    G, C = kra.fit_gram(ctx, ks.basis_dev, s)
    out['G_' + mt] = G; out['C_' + mt] = C
np.savez(sys.argv[2], **out)
"""


def test_congruence_grams_of_dim_red_dictionaries_equal_the_per_pair_projection_on_the_arm_data():
    """Linear and nonlinear dim_red dictionaries take their Grams as T'(Psi_full' Psi_full) T from the monomial kernels;
    KP_NO_GRAM_CONGRUENCE=1 (read per process) keeps the general kernel, which projects every lifted pair.  The two summation orders
    are compared where they differ most: the arm data, whose FULL poly-3 dictionaries are rank-deficient (rank 252 of 336 for the
    bilinear row) - G, C agree to 1e-11 max|G| (the congruence forms differences of O(max|G_full|) terms)."""
    import os
    import subprocess
    import sys
    import tempfile
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    res = {}
    with tempfile.TemporaryDirectory() as td:
        for name, env in (("cong", {}), ("general", {"KP_NO_GRAM_CONGRUENCE": "1"})):
            f = os.path.join(td, name + ".npz")
            r = subprocess.run([sys.executable, "-c", _CONG_SCRIPT, root, f], capture_output=True, text=True, timeout=600, env=dict(os.environ, **env))
            assert r.returncode == 0, r.stderr[-1500:]
            res[name] = dict(np.load(f))
    for mt in ("linear", "nonlinear"):
        scale = np.abs(res["general"]["G_" + mt]).max()
        for k in ("G_" + mt, "C_" + mt):
            assert np.abs(res["cong"][k] - res["general"][k]).max() <= 1e-11 * scale, k


_PRELIFT_RANDOM_SCRIPT = """
import sys, numpy as np
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, sys.argv[1] + '/tests')
import koopman_realizations_amd as kra
from conftest import synth_pairs
ctx = kra.Context(0)
rng = np.random.default_rng(77)
out = {}
for i in range(24):
    nz = int(rng.integers(2, 7)); m = int(rng.integers(1, 4)); deg = int(rng.integers(2, 4)); Ns = int(rng.choice([3, 8, 9, 257, 4099, 20011]))
    tab = kra.poly_exponent_table(nz, deg)
    nfull = len(tab) + 1
    if nfull > 96:
        continue
    p = synth_pairs(Ns, nz, m, seed=100 + i)
    s = kra.Snapshots(ctx, p["alpha"], p["beta"], p["u"])
    if i % 3 == 0:                                     # gaussian block beside the monomials
        ng = int(rng.integers(1, 12))
        if nz + (len(tab) - nz) + ng + 1 > 56:
            ng = max(1, 56 - len(tab) - 1)
        b = kra.Basis(ctx, "bilinear", nz, m, [("poly", tab[nz:]), ("gaussian", rng.uniform(-1, 1, (nz, ng)))])
    else:                                              # dim_red
        k = int(rng.integers(1, min(32, nfull - 1) + 1))
        pcs = np.linalg.qr(rng.standard_normal((nfull, nfull)))[0][:, :k].copy()
        b = kra.Basis(ctx, "bilinear", nz, m, [("poly", tab[nz:])], pcs)
    G, C = kra.fit_gram(ctx, b, s)
    out["G%d" % i] = G; out["C%d" % i] = C; out["d%d" % i] = np.array([nz, m, deg, Ns, b.N, b.W])
np.savez(sys.argv[2], **out)
"""


def test_lifted_tiles_on_random_dictionaries_equal_the_in_kernel_lift():
    """Two dozen random bilinear dictionaries - 2 ... 6 states, 1 ... 3 inputs, degree 2 / 3, dim_red with 1 ... 32 components or a
    gaussian block, 3 ... 20 011 pairs (fewer than a tile, ragged tails) - through the lifted-tile form at every size
    (KP_GRAM3_PRELIFT_MIN_NS=0) and through the in-kernel lift (KP_GRAM3_NO_PRELIFT=1): the same G, C to 1e-12 max|G|."""
    import os
    import subprocess
    import sys
    import tempfile
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    res = {}
    with tempfile.TemporaryDirectory() as td:
        for name, env in (("pre", {"KP_GRAM3_PRELIFT_MIN_NS": "0"}), ("proj", {"KP_GRAM3_NO_PRELIFT": "1"})):
            f = os.path.join(td, name + ".npz")
            r = subprocess.run([sys.executable, "-c", _PRELIFT_RANDOM_SCRIPT, root, f], capture_output=True, text=True, timeout=600, env=dict(os.environ, **env))
            assert r.returncode == 0, r.stderr[-1500:]
            res[name] = dict(np.load(f))
    n = 0
    for k in res["pre"]:
        if k[0] in "GC":
            a, b = res["pre"][k], res["proj"][k]
            assert np.isfinite(a).all()
            assert np.abs(a - b).max() <= 1e-12 * max(np.abs(res["proj"]["G" + k[1:]]).max(), 1e-300), (k, res["pre"]["d" + k[1:]])
            n += 1
    assert n >= 30


def test_fits_do_not_depend_on_what_the_dictionary_was_fitted_on_before(ctx):
    """A dictionary remembers the rank its last synchronous fit found (`rank_hint`: the next fit queues only the pivoting panels
    that reach that rank and holds back the copy of a K that will most likely be replaced).  Only the ORDER of the work may
    depend on that: every fit below gives, bit for bit, what a fresh dictionary object gives on the same snapshots -
    rank-deficient after full rank, full rank after rank-deficient, and a rank far ABOVE the remembered one (more than a
    panel of 32 pivots: the factorisation continues behind the first synchronisation)."""
    rng = np.random.default_rng(5)
    dic = ko.build_dictionary("bilinear", 6, 3, ["poly"], [3])         # W = 336
    Ns = 4000

    def pairs(nfree):                                                    # states confined to an nfree-dimensional subspace
        mix = np.linalg.qr(rng.standard_normal((6, 6)))[0][:, :nfree]
        a = rng.uniform(-1, 1, (Ns, nfree)) @ mix.T
        b_ = 0.9 * a + 0.05 * rng.uniform(-1, 1, (Ns, nfree)) @ mix.T
        return {"alpha": a, "beta": b_, "u": rng.uniform(-1, 1, (Ns, 3))}

    sets = [pairs(6), pairs(3), pairs(5), pairs(6), pairs(4), pairs(3)]
    shared = make_basis(ctx, dic)
    ranks = []
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        for p in sets:
            s = kra.Snapshots(ctx, p["alpha"], p["beta"], p["u"])
            K1 = kra.fit(ctx, shared, s)[0]
            r1 = ctx.last_rank()
            fresh = make_basis(ctx, dic)
            K2 = kra.fit(ctx, fresh, s)[0]
            assert ctx.last_rank() == r1 and np.array_equal(K1, K2)
            K3 = kra.fit(ctx, shared, s)[0]                             # and once more with the hint now matching
            assert ctx.last_rank() == r1 and np.array_equal(K1, K3)
            ranks.append(r1)
            s.close(); fresh.close()
    assert ranks[0] == 336 and ranks[3] == 336 and ranks[1] < ranks[2] - 32 and ranks[5] < ranks[4] - 32, ranks


@pytest.mark.parametrize("hint", [0, 100, 200, 240, 252, 300, 335])
def test_a_wrong_remembered_rank_changes_nothing(ctx, arm, hint):
    """The arm data's raw bilinear poly-3 dictionary has rank 252 of 336.  Whatever rank the dictionary 'remembers' (forced here
    through KP_RANK_HINT_TEST) - far too small (the factorisation continues behind the first synchronisation), a little too
    small (finished within the queued panels but beyond the width the substitution was queued at: repeated at full width), exact,
    too large - the fit returns the same K bit for bit, and the rank."""
    import os, warnings
    p = arm["pairs"]
    dic = ko.build_dictionary("bilinear", 6, 3, ["poly"], [3])
    # the hook is honoured only by a context created while KP_TEST_HOOKS is set (the shipped library does not read it per fit)
    os.environ["KP_TEST_HOOKS"] = "1"
    try:
        ctx = kra.Context(0)
    finally:
        del os.environ["KP_TEST_HOOKS"]
    b = make_basis(ctx, dic)
    snaps = kra.Snapshots(ctx, p["alpha"], p["beta"], p["u"])
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        Kref = kra.fit(ctx, make_basis(ctx, dic), snaps)[0]
        assert ctx.last_rank() == 252
        os.environ["KP_RANK_HINT_TEST"] = str(hint)
        try:
            K = kra.fit(ctx, b, snaps)[0]
        finally:
            del os.environ["KP_RANK_HINT_TEST"]
    assert ctx.last_rank() == 252 and np.array_equal(K, Kref)
    snaps.close()
    ctx.close()
