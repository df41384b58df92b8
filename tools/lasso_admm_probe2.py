"""Host probe 4: ADMM with small rho, then the uncapped active-set rounds from its feasible iterate."""
import sys, os
import numpy as np
import scipy.linalg as sl
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from lasso_pdas_probe import project_l1_ball, pdas
from lasso_admm_probe import fobj

def admm(G, C, t, rho, iters, alpha=1.6, report=()):
    W = G.shape[0]
    cf = sl.cho_factor(G + rho * np.eye(W))
    Z = np.zeros_like(C); U = np.zeros_like(C)
    snaps = {}
    for it in range(1, iters + 1):
        K = sl.cho_solve(cf, C + rho * (Z - U))
        Kr = alpha * K + (1 - alpha) * Z
        Zn = project_l1_ball((Kr + U).ravel(), t).reshape(C.shape)
        U = U + Kr - Zn
        dz = np.abs(Zn - Z).max(); Z = Zn
        if it in report:
            snaps[it] = (Z.copy(), np.abs(K - Z).max(), dz, rho * np.abs(U).max())
    return snaps

d = np.load(sys.argv[1]); G, C, Kls = d["G"], d["C"], d["Kls"]
G = (G + G.T) / 2
ev = np.linalg.eigvalsh(G)
f = float(sys.argv[2]); t = f * np.abs(Kls).sum()
rep = (100, 200, 400, 800, 1600)
for rho in [float(x) for x in sys.argv[3:]]:
    snaps = admm(G, C, t, rho, rep[-1], report=rep)
    for it in rep:
        Z, r, dz, th = snaps[it]
        Kh, thp, hist = pdas(G, C, t, Z, rounds=10, verbose=False)
        print("rho %.1e it %4d f(Z) %.10e |K-Z| %.1e dZ %.1e theta~%.2e nnz %d | rounds " % (rho, it, fobj(G, C, Z), r, dz, th, (Z != 0).sum()) + " ".join("%d(%.1e)" % (h[1], h[0]) for h in hist) + "  f %.10e" % hist[-1][3])
