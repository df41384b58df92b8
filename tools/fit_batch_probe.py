"""Kernel time of kp_fit_batch (kp_small_fit_kernel) for 1024 systems x 9000 pairs, by dictionary."""
import os, sys, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import koopman_realizations_amd as kra
ctx = kra.Context(0)
rng = np.random.default_rng(0)
nb, Ns = 1024, 9000
a = rng.uniform(-1, 1, (nb * Ns, 1)); b = np.clip(a + 0.05 * rng.standard_normal(a.shape), -1, 1); u = rng.uniform(-1, 1, (nb * Ns, 1))
for mt, deg in (("linear", 13), ("linear", 4), ("bilinear", 6), ("nonlinear", 4)):
    nv = 2 if mt == "nonlinear" else 1
    basis = kra.Basis(ctx, mt, 1, 1, [("poly", kra.poly_exponent_table(nv, deg)[nv:])])
    snaps = kra.Snapshots(ctx, a, b, u)
    ts = []
    for _ in range(4):
        ctx.fit_batch(basis, snaps, nb); ts.append(ctx.timer(0))
    print("%-9s deg %2d  W %2d  kernel %.3f ms" % (mt, deg, basis.W, min(ts)))
    snaps.close(); basis.close()
