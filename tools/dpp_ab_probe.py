"""A/B numbers for kernels that use DPP moves (KP_LIB_PATH selects the build): Cholesky phase cycles, linear / nonlinear monomial Gram
kernels (kp_gram5), MPC closed-loop kernel time, rank-deficient fit."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, koopman_realizations_amd as kra, bench
ctx = kra.Context(0)
a, b, u = bench.synth_pairs(100000, seed=5)
snaps = kra.Snapshots(ctx, a, b, u)
for mt, deg in (("linear", 3), ("nonlinear", 3), ("linear", 4)):
    nv = 6 + (3 if mt == "nonlinear" else 0)
    basis = kra.Basis(ctx, mt, 6, 3, [("poly", kra.poly_exponent_table(nv, deg)[nv:])])
    for _ in range(40): kra.fit_gram(ctx, basis, snaps, fetch=False)
    ts = []
    for _ in range(40):
        kra.fit_gram(ctx, basis, snaps, fetch=False); ts.append(ctx.timer(0))
    print(f"{mt} poly-{deg}: W {basis.W} gram_ms {np.mean(ts):.4f}")
    basis.close()
W = 336
rng = np.random.default_rng(0)
P = rng.standard_normal((4 * W, W)); G = P.T @ P; C = P.T @ rng.standard_normal((4 * W, W))
for _ in range(5): K = ctx.fit_solve(G, C)
print("W 336 solve timer ms", ctx.timer(1))
