"""CPU tests of the oracle's regularisation-path solver of solve_KoopmanQP (oracle/koopman_oracle.py: koopman_lasso_path), the checker of
the device's homotopy: pinned to the oracle's projected-gradient solver (itself the checker of the round 1-3 lasso tests) where that one
converges, and to the optimality conditions of the QP of Ksysid.m:1126-1137 on an ill-conditioned monomial Gram where it does not."""
import numpy as np

from oracle import koopman_oracle as ko
from conftest import synth_pairs


def test_path_solver_equals_projected_gradient_on_a_well_conditioned_problem():
    p = synth_pairs(3000, 3, 2, seed=5)
    dic = ko.build_dictionary("bilinear", 3, 2, ["poly"], [2])
    Px, Py = ko.px_py(dic, p)
    G, C = ko.gram(Px, Py)
    l1 = np.abs(np.linalg.solve(G, C)).sum()
    for f in (0.9, 0.4, 0.03):
        K1 = ko.koopman_lasso(G, C, f * l1)
        K2, theta = ko.koopman_lasso_path(G, C, f * l1)
        assert np.abs(K1 - K2).max() <= 1e-10 * np.abs(K1).max()
        th, res_on, off, feas = ko.lasso_kkt(G, C, K2, f * l1)
        assert abs(th - theta) <= 1e-9 * theta and res_on <= 1e-9 * theta and off <= 1 + 1e-9 and abs(feas - 1) <= 1e-12
    K3, theta = ko.koopman_lasso_path(G, C, 1.5 * l1)                            # inactive L1 row: the least-squares solution
    assert theta == 0.0 and np.abs(K3 - np.linalg.solve(G, C)).max() <= 1e-9 * np.abs(K3).max()


def test_path_solver_is_optimal_on_an_ill_conditioned_monomial_gram():
    """Degree-5 monomials of two strongly correlated smooth signals: cond(G) ~ 1e12.  The first-order oracle does not get there in
    2e5 iterations (its answer keeps a larger objective); the path solver satisfies the optimality conditions."""
    rng = np.random.default_rng(0)
    ts = np.linspace(0, 6, 1500)
    x = np.stack([np.sin(ts) + 0.01 * rng.standard_normal(ts.size), np.sin(ts + 0.05) + 0.01 * rng.standard_normal(ts.size)], 1)
    cols = [x[:, 0] ** a * x[:, 1] ** b for a in range(6) for b in range(6 - a)]
    Px = np.stack(cols, 1)[:-1]; Py = np.stack(cols, 1)[1:]
    G, C = Px.T @ Px, Px.T @ Py
    assert np.linalg.cond(G) > 1e9
    Gq = (G + G.T) / 2
    if np.linalg.eigvalsh(Gq).min() < 0:                                         # the PSD guard of Ksysid.m:1117-1120, as the solver applies it
        Gq = Gq + 1e-6 * np.eye(len(Gq))
    t = 0.5 * np.abs(np.linalg.solve(Gq, C)).sum()
    K, theta = ko.koopman_lasso_path(G, C, t)
    th, res_on, off, feas = ko.lasso_kkt(G, C, K, t)
    noise = 64 * np.finfo(float).eps * (np.abs(G) @ np.abs(K)).max()
    assert abs(feas - 1) <= 1e-10
    assert res_on <= 1e-9 * theta + noise
    assert off <= 1 + 1e-6 + noise / theta
    f = lambda X: 0.5 * (X * (G @ X)).sum() - (C * X).sum()
    Kpg = ko.koopman_lasso(G, C, t, iters=20000)
    assert f(K) <= f(Kpg) + 1e-9 * abs(f(Kpg))


def test_path_solver_takes_tied_entries_together():
    """Two identical dictionary columns under the 1e-6 PSD guard (Ksysid.m:1117-1120): their correlations tie at every event, the second
    one sits ON the boundary and moves out - it has to enter in the same breath, and the strictly convex guarded problem splits the
    weight equally between the twins (what the projected-gradient oracle finds)."""
    rng = np.random.default_rng(1)
    P = rng.standard_normal((300, 6)); P[:, 4] = P[:, 1]
    Y = P @ rng.standard_normal((6, 6))
    G, C = P.T @ P, P.T @ Y
    Gg = (G + G.T) / 2
    if np.linalg.eigvalsh(Gg).min() >= 0:
        G = G - 1e-9 * np.eye(6)                                                  # make sure the guard applies, as on the device after a failed factorisation
        Gg = (G + G.T) / 2
    Gg = Gg + 1e-6 * np.eye(6)
    Kg = np.linalg.solve(Gg, C)
    for f in (0.7, 0.2):
        t = f * np.abs(Kg).sum()
        K, theta = ko.koopman_lasso_path(G, C, t)
        assert np.abs(K[1] - K[4]).max() <= 1e-6 * np.abs(K).max()
        K1 = ko.koopman_lasso(G, C, t)
        assert np.abs(K - K1).max() <= 1e-6 * np.abs(K1).max()
    K0, th0 = ko.koopman_lasso_path(G, C, 2.0 * np.abs(Kg).sum())
    assert th0 == 0.0 and np.abs(K0 - Kg).max() <= 1e-6 * np.abs(Kg).max()


def test_both_oracle_solvers_solve_the_literal_qp_of_the_reference():
    """The QP exactly as solve_KoopmanQP writes it (Ksysid.m:1112-1137): x = [K+; K-] >= 0, M = [I, -I], H = M'(I (x) Px'Px)M,
    f = -M'vec(Px'Py), rows Aq = [-I; 1'], bq = [0; t] - solved here by a general-purpose solver (scipy, exact gradient and Hessian) on a
    5 x 5 problem (trust-constr, an interior-point method like quadprog's default).  vec(K) = M x from that solver, the oracle's projected-gradient solver and its path solver agree (1e-5: the general solver's
    stopping tolerance), and all three reach the same objective - the restatement `min 1/2|Px K - Py|^2 s.t. |vec K|_1 <= t` that the
    oracle and the device work on IS the reference's QP."""
    from scipy.optimize import minimize, LinearConstraint
    rng = np.random.default_rng(12)
    Nm = 5
    Px = rng.standard_normal((60, Nm)); Py = Px @ (rng.standard_normal((Nm, Nm)) * (rng.random((Nm, Nm)) < 0.6)) + 0.1 * rng.standard_normal((60, Nm))
    PxTPx = Px.T @ Px                                                       # :1114
    PxTPy = Px.T @ Py                                                       # :1125
    M = np.hstack([np.eye(Nm * Nm), -np.eye(Nm * Nm)])                      # :1112
    ATA = np.kron(np.eye(Nm), PxTPx)                                        # :1126
    ATb = PxTPy.reshape(Nm * Nm, 1, order="F")                              # :1127 (MATLAB reshape: column-major)
    H = M.T @ (ATA @ M)                                                     # :1130-1131
    f = (-M.T @ ATb).ravel()                                                # :1132
    Kls = np.linalg.solve(PxTPx, PxTPy)
    for frac in (0.7, 0.25):
        t = frac * np.abs(Kls).sum()
        Aq = np.vstack([-np.eye(2 * Nm * Nm), np.ones((1, 2 * Nm * Nm))])   # :1136
        bq = np.concatenate([np.zeros(2 * Nm * Nm), [t]])                   # :1137
        res = minimize(lambda x: 0.5 * x @ H @ x + f @ x, np.full(2 * Nm * Nm, t / (4 * Nm * Nm)), jac=lambda x: H @ x + f, hess=lambda x: H,
                       method="trust-constr", constraints=[LinearConstraint(Aq, -np.inf, bq)],
                       options={"gtol": 1e-11, "xtol": 1e-13, "barrier_tol": 1e-12, "maxiter": 5000})
        assert res.constr_violation <= 1e-9, res.message
        Kq = (M @ res.x).reshape(Nm, Nm, order="F")                         # :1172 xout = M x; reshaped by the caller (:1076)
        K1 = ko.koopman_lasso(PxTPx, PxTPy, t)
        K2, _ = ko.koopman_lasso_path(PxTPx, PxTPy, t)
        obj = lambda K: 0.5 * np.linalg.norm(Px @ K - Py) ** 2
        assert np.abs(K1 - K2).max() <= 1e-10 * np.abs(K1).max()
        assert np.abs(Kq - K2).max() <= 1e-5 * np.abs(K2).max(), np.abs(Kq - K2).max()
        assert abs(obj(Kq) - obj(K2)) <= 2e-7 * obj(K2)                          # (the interior-point answer stops just inside: no exact zeros, as quadprog's)
        assert np.abs(Kq).sum() <= t * (1 + 1e-9)
