"""Host probe: the primal-dual active-set rounds of kp_lasso (polish_cols / combine / check) in numpy with NO cap on the support
size, on the ill-conditioned arm Grams dumped by tools/lasso_illcond_probe.py.  Does the iteration settle, and at what tolerance
can its answer be certified?"""
import sys, os
import numpy as np
import scipy.linalg as sl
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "oracle"))
from koopman_oracle import project_l1_ball

def pdas(G, C, t, K0, rounds=60, verbose=True):
    W, nc = C.shape
    pat = np.sign(K0)
    hist = []
    for r in range(rounds):
        A = np.zeros_like(C); B = np.zeros_like(C)
        for j in range(nc):
            S = np.flatnonzero(pat[:, j])
            if S.size == 0: continue
            cf = sl.cho_factor(G[np.ix_(S, S)])
            A[S, j] = sl.cho_solve(cf, C[S, j]); B[S, j] = sl.cho_solve(cf, pat[S, j])
        sa = (pat * A).sum(); sb = (pat * B).sum()
        th = (sa - t) / sb
        Kh = A - th * B
        g = G @ Kh - C
        on = pat != 0
        leave = on & ~(Kh * pat > 0)
        enter = ~on & (np.abs(g) > th * (1 + 1e-9))
        res = np.abs(g + th * pat)[on].max() if on.any() else 0.0
        f = 0.5 * (Kh * (G @ Kh)).sum() - (C * Kh).sum()
        if verbose:
            print("round %2d theta %.6e  nnz %5d leave %4d enter %4d  res_on %.3e  |Kh|_1-t %.2e f %.10e" % (r, th, on.sum(), leave.sum(), enter.sum(), res, np.abs(Kh).sum() - t, f))
        hist.append((th, leave.sum() + enter.sum(), res, f))
        if leave.sum() + enter.sum() == 0:
            return Kh, th, hist
        pat = np.where(leave, 0.0, pat)
        pat = np.where(enter, -np.sign(g), pat)
    return Kh, th, hist

if __name__ == "__main__":
    d = np.load(sys.argv[1]); G, C, Kls, N = d["G"], d["C"], d["Kls"], int(d["N"])
    G = (G + G.T) / 2
    print("W", G.shape[0], "cond %.2e" % np.linalg.cond(G), "max|G| %.3e max|C| %.3e max|Kls| %.3e" % (np.abs(G).max(), np.abs(C).max(), np.abs(Kls).max()))
    Kls2 = np.linalg.solve(G, C)
    print("Kls check", np.abs(Kls2 - Kls).max())
    for f in [float(x) for x in sys.argv[2:]] or [0.5, 0.1, 0.01]:
        t = f * np.abs(Kls).sum()
        K0 = project_l1_ball(Kls.ravel(), t).reshape(Kls.shape)
        print("== factor", f, "t", t, "start nnz", (K0 != 0).sum())
        Kh, th, hist = pdas(G, C, t, K0)
