"""Lasso grid at BASELINE configs[3]'s shape (W = 336, 1e5 pairs): iterations, time and KKT residual per value."""
import sys, time, numpy as np
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import koopman_realizations_amd as kra, bench
from oracle import koopman_oracle as ko
ctx = kra.Context(0); a, b, u = bench.synth_pairs(100000)
basis = kra.Basis(ctx, "bilinear", 6, 3, [("poly", kra.poly_exponent_table(6, 3)[6:])]); snaps = kra.Snapshots(ctx, a, b, u)
G, C = kra.fit_gram(ctx, basis, snaps)
Kls = ctx.fit_solve(G, C)
l1 = np.abs(Kls).sum(); print("W", basis.W, "||K_ls||_1", l1, "N", basis.N, "max|C|", np.abs(C).max())
nv = int(sys.argv[1]) if len(sys.argv) > 1 else 8
fr = np.geomspace(0.99, 0.01, nv)
ctx.fit_lasso_batch(G, C, fr[:1] * l1)
for reps in range(2):
    t0 = time.perf_counter(); Ks, it = ctx.fit_lasso_batch(G, C, fr * l1); dt = time.perf_counter() - t0
    print(f"batch of {nv}: {dt*1e3:.1f} ms wall, device {ctx.timer(3):.1f} ms, {dt*1e3/nv:.2f} ms/value; iters", it.tolist())
for f, K in list(zip(fr, Ks))[:: max(1, nv // 8)]:
    print("frac %.3f" % f, "nnz", int((K != 0).sum()), "l1", np.abs(K).sum() / (f * l1), "kkt/max|C|", ko.lasso_kkt_residual(G, C, K, f * l1) / np.abs(C).max())
