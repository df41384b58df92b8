"""Host probe 3: ADMM (K = Z split, one Cholesky of G + rho I for all columns and iterations) on the ill-conditioned arm Grams:
objective of the feasible iterate Z, Frank-Wolfe gap <g,Z> + t |g|_inf (a bound on f(Z) - f*), per iteration count and rho."""
import sys, os
import numpy as np
import scipy.linalg as sl
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from lasso_pdas_probe import project_l1_ball

def fobj(G, C, K): return 0.5 * (K * (G @ K)).sum() - (C * K).sum()
def gap(G, C, K, t):
    g = G @ K - C
    return (g * K).sum() + t * np.abs(g).max()

def admm(G, C, t, rho, iters, alpha=1.6, Z0=None, report=()):
    W = G.shape[0]
    cf = sl.cho_factor(G + rho * np.eye(W))
    Z = np.zeros_like(C) if Z0 is None else Z0.copy(); U = np.zeros_like(C)
    for it in range(1, iters + 1):
        K = sl.cho_solve(cf, C + rho * (Z - U))
        Kr = alpha * K + (1 - alpha) * Z
        Zn = project_l1_ball((Kr + U).ravel(), t).reshape(C.shape)
        U = U + Kr - Zn
        dz = np.abs(Zn - Z).max(); Z = Zn
        if it in report:
            print("   rho %.1e it %5d  f(Z) %.10e gap %.3e  |K-Z| %.2e dZ %.2e nnz %d" % (rho, it, fobj(G, C, Z), gap(G, C, Z, t), np.abs(K - Z).max(), dz, (Z != 0).sum()))
    return Z

d = np.load(sys.argv[1]); G, C, Kls = d["G"], d["C"], d["Kls"]
G = (G + G.T) / 2
ev = np.linalg.eigvalsh(G)
print("eig min %.3e max %.3e  f_LS %.10e" % (ev[0], ev[-1], fobj(G, C, Kls)))
for f in [float(x) for x in sys.argv[2:]] or [0.5]:
    t = f * np.abs(Kls).sum()
    print("== factor", f)
    for rho in (ev[-1] * 1e-2, ev[-1] * 1e-4, ev[-1] * 1e-6, ev[-1] * 1e-8):
        admm(G, C, t, rho, 3200, report=(50, 200, 800, 3200))
