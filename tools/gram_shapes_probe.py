import os, sys; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, koopman_realizations_amd as kra, bench
ctx = kra.Context(0)
a,b,u = bench.synth_pairs(100000)
basis = kra.Basis(ctx, "bilinear", 6, 3, [("poly", kra.poly_exponent_table(6, 3)[6:])]); snaps = kra.Snapshots(ctx, a, b, u)
for rep in range(3):
    for _ in range(128): kra.fit(ctx, basis, snaps, fetch=False)
    ctx.synchronize(); print("W336 gram ms", ctx.timer(0))
print({k: (round(v["gram_ms"],4), round(v["roofline"]["frac"],3), round(v["roofline"]["dense_equivalent_frac"],3)) for k, v in bench.bench_width_points(ctx, kra, 100000).items()})
