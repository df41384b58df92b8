"""One synchronous kp_fit (K back on the host) at the arm data's shape with dim_red (11 999 pairs, N = 34, W = 136), averaged over 300
calls: the prelift form (default) against the in-kernel projection (KP_GRAM3_NO_PRELIFT=1)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, koopman_realizations_amd as kra, bench
ctx = kra.Context(0)
NS = int(sys.argv[1]) if len(sys.argv) > 1 else 11999
a, b, u = bench.synth_pairs(NS, seed=3)
pcs = np.linalg.qr(np.random.default_rng(0).standard_normal((84, 27)))[0]
basis = (kra.Basis(ctx, "bilinear", 6, 3, [("gaussian", np.random.default_rng(3).uniform(-1, 1, (6, 20)))]) if len(sys.argv) > 2 and sys.argv[2] == "gauss"
         else kra.Basis(ctx, "bilinear", 6, 3, [("poly", kra.poly_exponent_table(6, 3)[6:])], pcs))
snaps = kra.Snapshots(ctx, a, b, u)
for _ in range(50): kra.fit(ctx, basis, snaps)
t0 = time.perf_counter()
for _ in range(300): kra.fit(ctx, basis, snaps)
dt = (time.perf_counter() - t0) / 300
print("ms per synchronous fit %.4f  gram kernel(s) %.4f ms  solve %.4f ms" % (dt * 1e3, ctx.timer(0), ctx.timer(1)))
