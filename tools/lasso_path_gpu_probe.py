"""GPU probe: kp_fit_lasso_batch on the dumped arm Grams (tools/data/*.npz) - the homotopy (KP_LASSO_PATH_AFTER=0: at once; default:
after 100 FISTA iterations) - with the KKT conditions of every answer checked on the host."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import koopman_realizations_amd as kra

ctx = kra.Context(0)
here = os.path.dirname(os.path.abspath(__file__))
names = sys.argv[1].split(",") if len(sys.argv) > 1 else ["ill_l2d", "ill2", "ill_l3", "ill_b2", "ill3"]
factors = [float(x) for x in sys.argv[2].split(",")] if len(sys.argv) > 2 else [0.9, 0.5, 0.1, 0.01, 0.001]
for nm in names:
    d = np.load(os.path.join(here, "data", nm + ".npz")); G, C, Kls = d["G"], d["C"], d["Kls"]
    W = G.shape[0]
    ev = np.linalg.eigvalsh((G + G.T) / 2)
    Gq = G + 1e-6 * np.eye(W) if ev[0] <= 0 else G           # the PSD guard of Ksysid.m:1117-1120 as the device applies it (on a failed factorisation)
    tv = np.array(factors) * np.abs(Kls).sum()
    t0 = time.time()
    try:
        Ks, its = ctx.fit_lasso_batch(G, C, tv)
    except kra.KoopmanHipError as e:
        print("%-8s W %3d FAILED %.1f ms: %s" % (nm, W, (time.time() - t0) * 1e3, e)); continue
    ms = (time.time() - t0) * 1e3
    print("%-8s W %3d eig %.1e..%.1e  %.1f ms wall, homotopy %.2f ms, iters %s" % (nm, W, ev[0], ev[-1], ms, ctx.timer(11), list(its)))
    for f, t, K in zip(factors, tv, Ks):
        g = Gq @ K - C
        on = K != 0
        # multiplier from the support: g_i = -theta s_i
        th = np.median(-(g * np.sign(K))[on]) if on.any() else np.abs(g).max()
        res_on = np.abs(g + th * np.sign(K))[on].max() if on.any() else 0.0
        off = np.abs(g[~on]).max() / th if (~on).any() and th > 0 else 0.0
        fobj = 0.5 * (K * (Gq @ K)).sum() - (C * K).sum()
        print("    factor %-6g |K|_1/t %.12f nnz %5d theta %.4e on-support residual %.2e (rel %.1e) off-support max|g|/theta %.6f  f %.10e"
              % (f, np.abs(K).sum() / t, on.sum(), th, res_on, res_on / max(th, 1e-300), off, fobj))
