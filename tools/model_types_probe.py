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
# the reference's own example_sysid.m settings: every model type with dim_red (econ lift [zeta; pcs' psi; 1])
rng = np.random.default_rng(3)
for mt, nfull_vars, k in (("linear", 6, 27), ("bilinear", 6, 27), ("nonlinear", 9, 60)):
    tab = kra.poly_exponent_table(nfull_vars, 3)[nfull_vars:]
    nfull = nfull_vars + len(tab) + 1
    pcs = np.linalg.qr(rng.standard_normal((nfull, k)))[0]
    basis = kra.Basis(ctx, mt, 6, 3, [("poly", tab)], pcs)
    for Nsx, sn in ((Ns, snaps),):
        for _ in range(24):
            kra.fit(ctx, basis, sn, fetch=False)
        ctx.synchronize()
        for _ in range(32):
            kra.fit(ctx, basis, sn, fetch=False)
        ctx.synchronize()
        ms = ctx.timer(0)
        W = basis.W
        print(f"{mt:9s} poly-3 dim_red: nfull {nfull} k {k} N {basis.N} W {W}  gram {ms:.4f} ms = {Nsx / (ms * 1e-3):.3e} pairs/s")
    basis.close()
