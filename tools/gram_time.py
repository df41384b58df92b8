"""Kernel time of the fused lift+Gram launch alone (no solve): mean of the last launches of a queue of kp_fit_gram calls.
Used with KP_LIB_PATH=<experimental build> for timing-only ablations (wrong Grams do not matter here)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import koopman_realizations_amd as kra
from bench import synth_pairs
Ns = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
deg = int(sys.argv[2]) if len(sys.argv) > 2 else 3
ctx = kra.Context(0)
a, b, u = synth_pairs(Ns)
basis = kra.Basis(ctx, "bilinear", 6, 3, [("poly", kra.poly_exponent_table(6, deg)[6:])])
snaps = kra.Snapshots(ctx, a, b, u)
for _ in range(300):
    kra.fit_gram(ctx, basis, snaps, fetch=False)
ts = []
for _ in range(100):
    kra.fit_gram(ctx, basis, snaps, fetch=False)
    ts.append(ctx.timer(0))
t0 = time.perf_counter()
for _ in range(200):
    kra.fit_gram(ctx, basis, snaps, fetch=False)
ctx.synchronize()
wall = (time.perf_counter() - t0) / 200
print(f"lib={os.environ.get('KP_LIB_PATH','default')} Ns={Ns} W={basis.W} gram_ms mean={np.mean(ts):.4f} min={np.min(ts):.4f} wall_ms={wall*1e3:.4f} flop/pair={ctx.timer(10):.0f}")
