"""Host probe 7: the column homotopy as the device kernel will run it - explicit inverse M = G_SS^-1 kept by bordering / deletion
(rank-1 updates, no triangular solves), one refinement of the direction from its own residual, periodic re-synchronisation of
r = c - G k and of the on-support identity r_S = theta s_S.  Numerics against probe 6 (dense solves)."""
import sys, os, time
import numpy as np

class Path:
    def __init__(self, G, c, resync=16, refine_tol=1e-11, refine_u=True):
        self.refine_u = refine_u
        self.G, self.c = G, c
        W = G.shape[0]; self.W = W
        self.k = np.zeros(W); self.r = c.copy(); self.sgn = np.zeros(W)
        self.idx = []; self.M = np.zeros((W, W))
        self.resync, self.refine_tol = resync, refine_tol
        j0 = int(np.argmax(np.abs(self.r))); self.theta = abs(self.r[j0])
        self.steps = 0; self.nref = 0; self.bad = 0
        self.last_add = -1; self.last_del = -1; self.last_del_sgn = 0.0
        self.bps = [(self.theta, 0.0)]
        self._add(j0, np.sign(self.r[j0]))

    def _add(self, p, s):
        n = len(self.idx); G = self.G; M = self.M
        if n == 0:
            M[0, 0] = 1.0 / G[p, p]
        else:
            S = np.array(self.idx)
            g = G[S, p]
            u = M[:n, :n] @ g
            if self.refine_u: u = u + M[:n, :n] @ (g - G[np.ix_(S, S)] @ u)          # one refinement of u
            alpha = G[p, p] - g @ u
            if not alpha > 0: self.bad += 1; return False
            M[:n, :n] += np.outer(u, u) / alpha
            M[:n, n] = -u / alpha; M[n, :n] = -u / alpha; M[n, n] = 1.0 / alpha
        self.idx.append(p); self.sgn[p] = s
        return True

    def _del(self, p):
        n = len(self.idx); q = self.idx.index(p); M = self.M
        m = M[:n, q].copy(); mq = m[q]
        M[:n, :n] -= np.outer(m, m) / mq
        last = n - 1
        if q != last:
            M[q, :n] = M[last, :n]; M[:n, q] = M[:n, last]; M[q, q] = M[last, last]
            self.idx[q] = self.idx[last]
        self.idx.pop(); self.sgn[p] = 0.0; self.k[p] = 0.0

    def advance(self, theta_stop, max_steps):
        G = self.G; W = self.W
        while self.theta > theta_stop and self.steps < max_steps:
            self.steps += 1
            n = len(self.idx); S = np.array(self.idx); M = self.M[:n, :n]
            if self.steps % self.resync == 0:
                self.r = self.c - G[:, S] @ self.k[S]
                e = self.r[S] - self.theta * self.sgn[S]
                dk = M @ e
                self.k[S] += dk; self.r -= G[:, S] @ dk
            sS = self.sgn[S]
            d = M @ sS
            a = G[:, S] @ d
            res = sS - a[S]
            if np.abs(res).max() > self.refine_tol:
                self.nref += 1
                d = d + M @ res
                a = G[:, S] @ d
            th = self.theta
            best = th - theta_stop; ev = None
            off = np.ones(W, bool); off[S] = False
            io = np.flatnonzero(off)
            for s in (1.0, -1.0):
                den = s * a[io] - 1.0; num = s * self.r[io] - th
                with np.errstate(divide="ignore", invalid="ignore"):
                    dl = num / den
                allowed = (den != 0) & ~((io == self.last_del) & (s == self.last_del_sgn))
                ok = allowed & (dl > 1e-14 * th)
                now = allowed & ~ok & (num >= 0) & (den < 0)
                dl = np.where(now, 0.0, dl); ok = ok | now
                if ok.any():
                    j = np.argmin(np.where(ok, dl, np.inf))
                    if dl[j] < best: best = dl[j]; ev = ("add", int(io[j]), s)
            dS = d; kS = self.k[S]
            mov = (dS * sS < 0) & (S != self.last_add)
            if mov.any():
                dl = np.where(mov, -kS / np.where(mov, dS, 1.0), np.inf)
                j = np.argmin(dl)
                if max(dl[j], 0.0) < best: best = max(dl[j], 0.0); ev = ("del", int(S[j]), 0.0)
            self.k[S] += best * d; self.r -= best * a; self.theta = th - best
            self.last_add = self.last_del = -1
            if ev is not None:
                if ev[0] == "add":
                    if self._add(ev[1], ev[2]): self.last_add = ev[1]
                else:
                    self.last_del_sgn = self.sgn[ev[1]]; self._del(ev[1]); self.last_del = ev[1]
            self.bps.append((self.theta, np.abs(self.k).sum()))
        return self.k.copy()

if __name__ == "__main__":
    d = np.load(sys.argv[1]); G, C, Kls = d["G"], d["C"], d["Kls"]
    G = (G + G.T) / 2
    if np.linalg.eigvalsh(G)[0] <= 0: G = G + 1e-6 * np.eye(G.shape[0]); print("PSD guard applied")
    stops = [float(x) for x in sys.argv[2].split(",")]
    cols = range(C.shape[1]) if len(sys.argv) < 4 else range(0, C.shape[1], int(sys.argv[3]))
    W = G.shape[0]
    Ks = [np.zeros_like(C) for _ in stops]
    tot = 0; mx = 0; nref = 0; bad = 0; t0 = time.time()
    for j in cols:
        p = Path(G, C[:, j], resync=int(os.environ.get('RESYNC', 16)), refine_tol=float(os.environ.get('RTOL', 1e-11)), refine_u=os.environ.get('REFU', '1') == '1')
        for si, st in enumerate(stops):
            Ks[si][:, j] = p.advance(st, 40 * W)
        tot += p.steps; mx = max(mx, p.steps); nref += p.nref; bad += p.bad
    print("steps total %d max/column %d (W %d) refinements %d bad adds %d  (%.1f s)" % (tot, mx, W, nref, bad, time.time() - t0))
    for st, K in zip(stops, Ks):
        K = K[:, list(cols)]; Cc = C[:, list(cols)]
        g = G @ K - Cc; on = K != 0
        print("theta %.3e: |K|_1 %.8e nnz %d; KKT on-support %.2e off-support max|g|/theta %.6f f %.10e"
              % (st, np.abs(K).sum(), on.sum(), np.abs(g + st * np.sign(K))[on].max() if on.any() else 0, np.abs(g[~on]).max() / st if (~on).any() else 0, 0.5 * (K * (G @ K)).sum() - (Cc * K).sum()))
