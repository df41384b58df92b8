"""Host probe 2: FISTA (as kp_lasso runs it: gradient restart, exact projection) for k iterations, then the active-set rounds with
no cap on the support; at which k do the rounds settle?"""
import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from lasso_pdas_probe import pdas, project_l1_ball

def fista(G, C, t, K0, iters, L):
    K = K0.copy(); Ko = K0.copy(); tk = 1.0; mom = 0.0
    out = {}
    for it in range(1, iters + 1):
        Y = K + mom * (K - Ko)
        V = Y - (G @ Y - C) / L
        Kn = project_l1_ball(V.ravel(), t).reshape(K.shape)
        restart = ((Y - Kn) * (Kn - K)).sum() > 0
        tn = 1.0 if restart else 0.5 * (1 + np.sqrt(1 + 4 * tk * tk))
        mom = 0.0 if restart else (tk - 1) / tn
        tk = tn
        Ko, K = K, Kn
        if it in CHECK: out[it] = K.copy()
    return out

CHECK = [20, 100, 400, 1600, 6400]
d = np.load(sys.argv[1]); G, C, Kls = d["G"], d["C"], d["Kls"]
G = (G + G.T) / 2
L = np.linalg.eigvalsh(G)[-1]
for f in [float(x) for x in sys.argv[2:]] or [0.5]:
    t = f * np.abs(Kls).sum()
    K0 = project_l1_ball(Kls.ravel(), t).reshape(Kls.shape)
    snaps = fista(G, C, t, K0, CHECK[-1], L)
    for it in CHECK:
        K = snaps[it]
        fv = 0.5 * (K * (G @ K)).sum() - (C * K).sum()
        print("=== factor %g  FISTA it %d  f %.10e nnz %d" % (f, it, fv, (K != 0).sum()))
        Kh, th, hist = pdas(G, C, t, K, rounds=12, verbose=False)
        print("    rounds: " + " ".join("%d(%.1e)" % (h[1], h[0]) for h in hist), " final f %.10e" % hist[-1][3])
