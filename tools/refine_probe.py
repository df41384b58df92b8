"""Cost of the public Ksysid path at the config-2 shape: kp_fit + kp_fit_refine (one step) + Px/Py lift and fetch."""
import sys, os, time, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import koopman_realizations_amd as kra, bench
from koopman_realizations_amd import _ffi as F
ctx = kra.Context(0)
Ns = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
a, b, u = bench.synth_pairs(Ns)
basis = kra.Basis(ctx, "bilinear", 6, 3, [("poly", kra.poly_exponent_table(6, 3)[6:])]); snaps = kra.Snapshots(ctx, a, b, u)
K = kra.fit(ctx, basis, snaps)[0]
for i in range(3):
    t0 = time.perf_counter(); K = kra.fit(ctx, basis, snaps)[0]; t1 = time.perf_counter()
    K1 = kra.fit_refine(ctx, basis, snaps, K, 1); t2 = time.perf_counter()
    Px = basis.lift(F.LIFT_ROW, a, u); t3 = time.perf_counter()
    print("fit %.2f ms  refine %.2f ms  lift+fetch Px (Ns x W) %.1f ms   |K1-K| %.2e" % ((t1 - t0) * 1e3, (t2 - t1) * 1e3, (t3 - t2) * 1e3, np.abs(K1 - K).max()))
print("pivot ratio of the config-2 Gram:", ctx.last_pivot_ratio() if (kra.fit(ctx, basis, snaps) is not None) else None)
