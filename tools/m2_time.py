"""Gram kernel time of a bilinear dictionary with TWO inputs (6 Kronecker weights): poly-3 on 6 states, 1e5 pairs.  KP_GRAM3_NOTUP=1 for the A/B."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, koopman_realizations_amd as kra, bench
ctx = kra.Context(0)
a, b, u = bench.synth_pairs(100000, 6, 2, seed=4)
basis = kra.Basis(ctx, "bilinear", 6, 2, [("poly", kra.poly_exponent_table(6, 3)[6:])])
snaps = kra.Snapshots(ctx, a, b, u)
for _ in range(200): kra.fit_gram(ctx, basis, snaps, fetch=False)
ts = []
for _ in range(100):
    kra.fit_gram(ctx, basis, snaps, fetch=False); ts.append(ctx.timer(0))
print("m=2 W", basis.W, "notup" if os.environ.get("KP_GRAM3_NOTUP") else "tup", "gram_ms %.4f" % np.mean(ts))
