"""Would a few ADMM (OSQP-style) iterations predict the active set of a cold MPC QP well enough to warm-start the dual
active-set solver?  For the independent random states of bench.py: rows to release / to add from the ADMM guess after k
iterations, against the optimal active set."""
import os, sys, ctypes as C
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import koopman_realizations_amd as kra, bench
from koopman_realizations_amd import _ffi as F
ctx = kra.Context(0)
a, b, u = bench.synth_pairs(100000)
basis = kra.Basis(ctx, "bilinear", 6, 3, [("poly", kra.poly_exponent_table(6, 3)[6:])])
snaps = kra.Snapshots(ctx, a, b, u)
mpc, setup = bench.mpc_problem(kra, ctx, basis, snaps)
zeta, u_prev, Yr = bench.mpc_inputs(24)
res = {k: [] for k in (5, 10, 20, 40)}
for i in range(24):
    U, z, st = mpc.step_zeta(basis, zeta[i], u_prev[i], Yr[i])
    H, f, A, bq = mpc.last_qp()
    x_opt = U.reshape(-1)
    nrm = np.linalg.norm(A, axis=1); ok = nrm > 0
    A = A[ok] / nrm[ok, None]; bq = bq[ok] / nrm[ok]
    r = A @ x_opt - bq
    tight = np.abs(r) < 1e-9
    # multipliers of the optimum (least squares on the tight rows) -> strongly active rows
    lam = np.linalg.lstsq(A[tight].T, -(H @ x_opt + f), rcond=None)[0]
    Fset = set(np.nonzero(tight)[0][lam > 1e-9].tolist())
    ev = np.linalg.eigvalsh(H)
    for rho in (np.sqrt(ev[0] * ev[-1]),):
        sigma = 1e-6 * ev[-1]
        M = np.linalg.inv(H + sigma * np.eye(len(f)) + rho * A.T @ A)
        x = np.zeros(len(f)); zc = np.minimum(A @ x, bq); y = np.zeros(len(bq)); alpha = 1.6
        for k in range(1, 41):
            xt = M @ (sigma * x - f + A.T @ (rho * zc - y))
            zt = A @ xt
            x = alpha * xt + (1 - alpha) * x
            zn = np.minimum(alpha * zt + (1 - alpha) * zc + y / rho, bq)
            y = y + rho * (alpha * zt + (1 - alpha) * zc - zn)
            zc = zn
            if k in res:
                G0 = [j for j in np.argsort(-y) if y[j] > 1e-9 * max(1.0, np.abs(y).max())]
                sel = []
                for j in G0:                              # independent subset, largest multipliers first
                    if len(sel) < len(f) and np.linalg.matrix_rank(A[sel + [j]]) == len(sel) + 1: sel.append(j)
                S = set(sel)
                res[k].append((len(Fset), len(S), len(S - Fset), len(Fset - S)))
for k, v in res.items():
    v = np.array(v)
    print(f"after {k:2d} ADMM iterations: |F| mean {v[:,0].mean():.1f}, guess {v[:,1].mean():.1f}, to release mean {v[:,2].mean():.1f} max {v[:,2].max()}, to add mean {v[:,3].mean():.1f} max {v[:,3].max()}")
