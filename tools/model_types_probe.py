"""Fused lift+Gram kernel time of the three model types at 1e5 pairs (poly-3 on 6 states + 3 inputs): which kernel serves
them, executed / dense-equivalent fraction of the f64 matrix peak."""
import os, sys, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import koopman_realizations_amd as kra, bench
ctx = kra.Context(0)
Ns = 100000
a, b, u = bench.synth_pairs(Ns, seed=5)
snaps = kra.Snapshots(ctx, a, b, u)
for mt, deg in (("linear", 3), ("linear", 4), ("nonlinear", 2), ("nonlinear", 3), ("bilinear", 2), ("bilinear", 3)):
    nv = 9 if mt == "nonlinear" else 6
    basis = kra.Basis(ctx, mt, 6, 3, [("poly", kra.poly_exponent_table(nv, deg)[nv:])])
    for _ in range(48):
        kra.fit(ctx, basis, snaps, fetch=False)
    ctx.synchronize()
    for _ in range(64):
        kra.fit(ctx, basis, snaps, fetch=False)
    ctx.synchronize()
    ms, ex = ctx.timer(0), ctx.timer(10)
    W = basis.W
    F = W * (W + 1) + 2.0 * W * W
    print(f"{mt:9s} poly-{deg}: N {basis.N:3d} W {W:3d}  gram {ms:.4f} ms  executed {ex * Ns / (ms * 1e-3) / 1e12 / 78.6:.3f} of peak, dense-equivalent {F * Ns / (ms * 1e-3) / 1e12 / 78.6:.3f}")
    basis.close()
