"""Host probe 6: the regularisation path of one column's penalised lasso  min 1/2 k'Gk - c'k + theta |k|_1  by the homotopy
(LARS with drops), theta from max|c| downwards; counts breakpoints until theta_stop and checks the joint optimum read off the
paths (sum_j |k_j(theta)|_1 = t) against the ADMM optimum of probe 4."""
import sys, os
import numpy as np
import scipy.linalg as sl

def column_path(G, c, theta_stop, max_steps=100000):
    """returns breakpoints [(theta, l1)], events, and k at theta_stop"""
    W = G.shape[0]
    k = np.zeros(W); sgn = np.zeros(W)
    r = c.copy()                                   # r = c - G k
    j0 = int(np.argmax(np.abs(r))); theta = abs(r[j0])
    if theta <= theta_stop: return [(theta_stop, 0.0)], 0, k
    S = [j0]; sgn[j0] = np.sign(r[j0])
    bps = [(theta, 0.0)]; steps = 0
    while steps < max_steps:
        steps += 1
        Sa = np.array(S)
        d = np.zeros(W)
        d[Sa] = sl.solve(G[np.ix_(Sa, Sa)], sgn[Sa], assume_a="pos")      # dk/d(-theta)
        a = G[:, Sa] @ d[Sa]                                              # dr/d(theta) ... r(theta - delta) = r - delta*(-a)?  r = c - Gk, k += delta d => r -= delta a
        # on S: r_i = theta s_i and a_i = s_i (consistent).  off S: |r_i - delta a_i| = theta - delta
        best = theta - theta_stop; ev = None
        off = np.ones(W, bool); off[Sa] = False
        for i in np.flatnonzero(off):
            for s in (1.0, -1.0):
                den = s * a[i] - 1.0                                      # s (r_i - delta a_i) = theta - delta  ->  delta (1 - s a_i) = theta - s r_i
                num = s * r[i] - theta
                if abs(den) > 0:
                    dl = num / den
                    if 1e-15 * theta < dl < best: best = dl; ev = ("add", i, s)
        for i in Sa:
            if d[i] * sgn[i] < 0:
                dl = -k[i] / d[i]
                if 0 <= dl < best: best = dl; ev = ("del", i, 0.0)
        k = k + best * d; r = r - best * a; theta -= best
        bps.append((theta, np.abs(k).sum()))
        if ev is None: break
        if ev[0] == "add": S.append(ev[1]); sgn[ev[1]] = ev[2]
        else: S.remove(ev[1]); sgn[ev[1]] = 0.0; k[ev[1]] = 0.0
    return bps, steps, k

if __name__ == "__main__":
    d = np.load(sys.argv[1]); G, C, Kls = d["G"], d["C"], d["Kls"]
    G = (G + G.T) / 2
    th_stop = float(sys.argv[2])
    W, nc = C.shape
    K = np.zeros_like(C); tot = 0; mx = 0
    import time; t0 = time.time()
    for j in range(nc):
        bps, steps, K[:, j] = column_path(G, C[:, j], th_stop)
        tot += steps; mx = max(mx, steps)
    g = G @ K - C
    on = K != 0
    print("theta_stop %.3e: steps total %d max/column %d (W %d); |K|_1 %.6e  nnz %d;  KKT on-support %.2e off-support max|g|/theta %.6f   f %.10e  (%.1f s)"
          % (th_stop, tot, mx, W, np.abs(K).sum(), on.sum(), np.abs(g + th_stop * np.sign(K))[on].max(), np.abs(g[~on]).max() / th_stop if (~on).any() else 0, 0.5 * (K * (G @ K)).sum() - (C * K).sum(), time.time() - t0))
    print("|Kls|_1", np.abs(Kls).sum())
