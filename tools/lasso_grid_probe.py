"""The 64-value lasso grid of bench.py (BASELINE configs[3]: lasso = t/N log-spaced in [1e-2, 1e2] on the W = 336 fit) on one GPU:
wall / device time of the grid, iterations per value; KP_LASSO_TRACE=1 prints the running-value count per check block."""
import sys, time, numpy as np
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import koopman_realizations_amd as kra, bench
ctx = kra.Context(0); a, b, u = bench.synth_pairs(100000)
basis = kra.Basis(ctx, "bilinear", 6, 3, [("poly", kra.poly_exponent_table(6, 3)[6:])]); snaps = kra.Snapshots(ctx, a, b, u)
vals = list(np.geomspace(1e-2, 1e2, int(sys.argv[1]) if len(sys.argv) > 1 else 64))
kra.fit(ctx, basis, snaps, vals[:1])
for rep in range(3):
    t0 = time.perf_counter(); Ks = kra.fit(ctx, basis, snaps, vals); dt = time.perf_counter() - t0
    print(f"grid of {len(vals)}: {dt*1e3:.2f} ms wall, lasso device {ctx.timer(3):.2f} ms", flush=True)
l1 = np.array([np.abs(K).sum() for K in Ks]); t = np.array(vals) * basis.N
print("active", int((l1 < l1.max() * (1 - 1e-9)).sum()), "budget met", bool(np.all(l1 <= t * (1 + 1e-9) + 1e-12)))
print("nnz per value", [int((K != 0).sum()) for K in Ks])
