"""Fused lift+Gram kernel time at 1e5 pairs of dictionaries that are not monomial (which still go through the general kernel?)."""
import os, sys, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import koopman_realizations_amd as kra, bench
ctx = kra.Context(0)
Ns = 100000
a, b, u = bench.synth_pairs(Ns, seed=5)
snaps = kra.Snapshots(ctx, a, b, u)
a3, b3, u3 = bench.synth_pairs(Ns, 3, 3, seed=6)
snaps3 = kra.Snapshots(ctx, a3, b3, u3)
rng = np.random.default_rng(3)
cases = [("linear", 6, [("gaussian", rng.uniform(-1, 1, (6, 10)))], snaps, "Ksysid_setup.m default: linear, 10 gaussians"),
         ("linear", 6, [("gaussian", rng.uniform(-1, 1, (6, 20)))], snaps, "linear, 20 gaussians"),
         ("linear", 3, [("fourier", 1)], snaps3, "linear, fourier degree 1 on 3 states"),
         ("nonlinear", 3, [("fourier", 1)], snaps3, "nonlinear, fourier degree 1 on [zeta; u] (6 variables)"),
         ("bilinear", 6, [("hermite", kra.poly_exponent_table(6, 3)[6:])], snaps, "bilinear, hermite degree 3"),
         ("linear", 6, [("hermite", kra.poly_exponent_table(6, 3)[6:])], snaps, "linear, hermite degree 3")]
for mt, nz, blocks, sn, what in cases:
    try:
        basis = kra.Basis(ctx, mt, nz, 3, blocks)
    except Exception as e:
        print(what, "->", repr(e)[:120]); continue
    try:
        for _ in range(16):
            kra.fit(ctx, basis, sn, fetch=False)
        ctx.synchronize()
        for _ in range(24):
            kra.fit(ctx, basis, sn, fetch=False)
        ctx.synchronize()
        print(f"{what}: N {basis.N} W {basis.W}  gram {ctx.timer(0):.4f} ms")
    except Exception as e:
        print(f"{what}: N {basis.N} W {basis.W} ->", repr(e)[:140])
    basis.close()
