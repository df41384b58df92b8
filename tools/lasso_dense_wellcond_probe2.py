"""As lasso_dense_wellcond_probe.py at W = 112 (bilinear poly-2 on 6 states, well conditioned): is the early handover for W <= 136 a loss?"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, koopman_realizations_amd as kra
from conftest import synth_pairs
ctx = kra.Context(0)
p = synth_pairs(100000)
b = kra.Basis(ctx, "bilinear", 6, 3, [("poly", kra.poly_exponent_table(6, 2)[6:])])
s = kra.Snapshots(ctx, p["alpha"], p["beta"], p["u"])
G, C = kra.fit_gram(ctx, b, s)
l1 = np.abs(ctx.fit_solve(G, C)).sum()
for f in (0.99, 0.95, 0.9, 0.7, 0.3):
    ctx.fit_lasso_batch(G, C, [f * l1])
    t0 = time.perf_counter(); K, it = ctx.fit_lasso_batch(G, C, [f * l1]); dt = time.perf_counter() - t0
    print("factor %.2f: %.1f ms, iterations %d, homotopy %.1f ms, nnz %d" % (f, dt * 1e3, it[0], ctx.timer(11), (K[0] != 0).sum()))
