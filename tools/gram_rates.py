"""Gram kernel time and algorithmic rate for the three model types (synthetic pairs, poly dictionaries)."""
import os, sys, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import koopman_realizations_amd as kra, bench
ctx = kra.Context(0)
a, b, u = bench.synth_pairs(100000)
for mt, deg in (("linear", 3), ("nonlinear", 3), ("bilinear", 3), ("linear", 4), ("nonlinear", 2), ("bilinear", 2)):
    nv = 9 if mt == "nonlinear" else 6
    basis = kra.Basis(ctx, mt, 6, 3, [("poly", kra.poly_exponent_table(nv, deg)[nv:])])
    snaps = kra.Snapshots(ctx, a, b, u)
    t = []
    for i in range(8):
        kra.fit_gram(ctx, basis, snaps, fetch=False); t.append((ctx.timer(0), ctx.timer(6)))
    t = np.array(t[3:]).mean(axis=0)
    W = basis.W
    fl = (W * (W + 1) + 2.0 * W * W) * 1e5
    print("%-9s deg %d N %3d W %3d gram %.3f ms reduce %.3f ms  -> %.1f TFLOP/s algorithmic" % (mt, deg, basis.N, W, t[0], t[1], fl / (t[0] * 1e-3) / 1e12))
