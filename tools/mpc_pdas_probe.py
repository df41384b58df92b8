"""Would a primal-dual active-set iteration (all violated rows in, all rows with negative multipliers out, per step) solve the cold MPC
QPs of bench.py's independent states in a handful of linear solves?  Per QP: iterations until the set repeats itself (converged) or
cycles, size of the sets, rank trouble (dependent active rows)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import koopman_realizations_amd as kra, bench
ctx = kra.Context(0)
a, b, u = bench.synth_pairs(100000)
basis = kra.Basis(ctx, "bilinear", 6, 3, [("poly", kra.poly_exponent_table(6, 3)[6:])])
snaps = kra.Snapshots(ctx, a, b, u)
mpc, setup = bench.mpc_problem(kra, ctx, basis, snaps)
nq = 48
zeta, u_prev, Yr = bench.mpc_inputs(nq)
conv = []; its = []; dep = 0
for i in range(nq):
    U, z, st = mpc.step_zeta(basis, zeta[i], u_prev[i], Yr[i])
    H, f, A, bq = mpc.last_qp()
    x_opt = U.reshape(-1)
    nrm = np.linalg.norm(A, axis=1); ok = nrm > 0
    A = A[ok] / nrm[ok, None]; bq = bq[ok] / nrm[ok]
    n = len(f); m = len(bq)
    Hi = np.linalg.inv(H)
    act = np.zeros(m, bool)
    seen = []
    status = "cap"
    for it in range(1, 31):
        idx = np.flatnonzero(act)
        if len(idx):
            Aa = A[idx]
            S = Aa @ Hi @ Aa.T
            rhs = -(bq[idx] + Aa @ Hi @ f)
            lam_a, *_ = np.linalg.lstsq(S, rhs, rcond=1e-12)
            if np.linalg.matrix_rank(S, tol=1e-9 * np.abs(S).max()) < len(idx): dep += 1
            x = -Hi @ (f + Aa.T @ lam_a)
        else:
            lam_a = np.zeros(0); x = -Hi @ f
        lam = np.zeros(m); lam[idx] = lam_a
        r = A @ x - bq
        new = (lam + 1.0 * r) > 1e-12
        key = new.tobytes()
        if (new == act).all():
            status = "ok" if np.abs(x - x_opt).max() < 1e-6 * max(1, np.abs(x_opt).max()) else "wrong"
            break
        if key in seen:
            status = "cycle"; break
        seen.append(key); act = new
    conv.append(status); its.append(it)
    print(i, status, it, int(act.sum()), "opt active", int((np.abs(A @ x_opt - bq) < 1e-9).sum()))
print("summary", {s: conv.count(s) for s in set(conv)}, "mean its of ok", np.mean([t for t, s in zip(its, conv) if s == "ok"]) if "ok" in conv else None, "dependent solves", dep)
