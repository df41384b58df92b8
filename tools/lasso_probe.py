import sys, time, numpy as np
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import koopman_realizations_amd as kra, bench
from oracle import koopman_oracle as ko
ctx = kra.Context(0); a, b, u = bench.synth_pairs(100000)
basis = kra.Basis(ctx, "bilinear", 6, 3, [("poly", kra.poly_exponent_table(6, 3)[6:])]); snaps = kra.Snapshots(ctx, a, b, u)
G, C = kra.fit_gram(ctx, basis, snaps)
Kls = ctx.fit_solve(G, C)
l1 = np.abs(Kls).sum(); print("W", basis.W, "||K_ls||_1", l1, "N", basis.N, "cond(G)", np.linalg.cond(G))
for frac in (0.9, 0.5, 0.1):
    t0 = time.perf_counter()
    try:
        K, it = ctx.fit_lasso(G, C, frac * l1, max_iter=4000, tol=1e-9)
        st = "ok"
    except Exception as e:
        K, it, st = None, -1, str(e)[:80]
    dt = time.perf_counter() - t0
    if K is not None:
        print("frac", frac, "iters", it, "time s %.3f" % dt, "l1", np.abs(K).sum() / l1, "kkt", ko.lasso_kkt_residual(G, C, K, frac * l1), st)
    else:
        print("frac", frac, "time s %.3f" % dt, st)
