"""Host probe 5: ADMM with residual balancing of rho (refactorisation on every change), over-relaxation 1.6, from Z0 = P(K_LS), on the
dumped arm Grams; iterations until |K - Z| and rho |Z - Zprev| are small; then the active-set rounds."""
import sys, os
import numpy as np
import scipy.linalg as sl
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from lasso_pdas_probe import project_l1_ball, pdas

def fobj(G, C, K): return 0.5 * (K * (G @ K)).sum() - (C * K).sum()

def admm_rb(G, C, t, Z0, rho0, iters, alpha=1.6, every=10, mu=10.0, tau=2.0, tol=1e-9, verbose=False):
    W = G.shape[0]
    rho = rho0
    cf = sl.cho_factor(G + rho * np.eye(W))
    Z = Z0.copy(); U = np.zeros_like(C)
    nfac = 1
    for it in range(1, iters + 1):
        K = sl.cho_solve(cf, C + rho * (Z - U))
        Kr = alpha * K + (1 - alpha) * Z
        Zn = project_l1_ball((Kr + U).ravel(), t).reshape(C.shape)
        U = U + Kr - Zn
        r = np.abs(K - Zn).max(); s = rho * np.abs(Zn - Z).max()
        Z = Zn
        kmax = max(1.0, np.abs(Z).max())
        if verbose and it % 50 == 0:
            print("     it %4d rho %.2e r %.2e s %.2e theta~%.3e f %.10e nnz %d" % (it, rho, r, s, rho * np.abs(U).max(), fobj(G, C, Z), (Z != 0).sum()))
        if r <= tol * kmax and s <= tol * max(rho * np.abs(U).max(), 1e-300):
            return Z, it, rho, nfac
        if it % every == 0:
            rn = np.linalg.norm(K - Z); sn = rho * np.linalg.norm(Zn - Z if False else 0) if False else None
        if it % every == 0:
            # Frobenius residuals
            rF = np.linalg.norm(K - Z); sF = s_last
            if rF > mu * sF: rho *= tau; U /= tau; cf = sl.cho_factor(G + rho * np.eye(W)); nfac += 1
            elif sF > mu * rF: rho /= tau; U *= tau; cf = sl.cho_factor(G + rho * np.eye(W)); nfac += 1
        s_last = rho * np.linalg.norm(Zn - (Zn if False else Z)) if False else rho * np.linalg.norm(U * 0 + (Z - Zprev)) if 'Zprev' in dir() else 0.0
        Zprev = Z.copy()
    return Z, iters, rho, nfac

if __name__ == "__main__":
    for path in sys.argv[1].split(","):
        d = np.load(path); G, C, Kls = d["G"], d["C"], d["Kls"]
        G = (G + G.T) / 2
        ev = np.linalg.eigvalsh(G)
        W = G.shape[0]
        print("#### %s W %d eig %.2e .. %.2e  |Kls|_1 %.4e f_LS %.10e" % (os.path.basename(path), W, ev[0], ev[-1], np.abs(Kls).sum(), fobj(G, C, Kls)))
        for f in [float(x) for x in sys.argv[2].split(",")]:
            t = f * np.abs(Kls).sum()
            Z0 = project_l1_ball(Kls.ravel(), t).reshape(Kls.shape)
            for rho0 in [float(x) for x in sys.argv[3].split(",")]:
                Z, it, rho, nfac = admm_rb(G, C, t, Z0, rho0 * ev[-1], 4000, verbose=len(sys.argv) > 4)
                Kh, thp, hist = pdas(G, C, t, Z, rounds=6, verbose=False)
                print("  factor %g rho0 %.0e*lmax: its %4d final rho %.2e (%d factorisations) f(Z) %.10e nnz %d | rounds " % (f, rho0, it, rho, nfac, fobj(G, C, Z), (Z != 0).sum())
                      + " ".join("%d(%.1e)" % (h[1], h[0]) for h in hist) + "  f %.10e" % hist[-1][3])
