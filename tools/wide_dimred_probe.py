import sys, time, numpy as np
sys.path.insert(0, '/root/repo')
import koopman_realizations_amd as kra, bench
ctx = kra.Context(0)
Ns = 12000
rng = np.random.default_rng(0)
a = rng.uniform(-1, 1, (Ns, 15)); b = rng.uniform(-1, 1, (Ns, 15)); u = rng.uniform(-1, 1, (Ns, 3))
snaps = kra.Snapshots(ctx, a, b, u)
for mt in ("linear", "bilinear"):
    for deg, k in ((2, 30), (3, 30)):
        tab = kra.poly_exponent_table(15, deg)[15:]
        nfull = 15 + len(tab) + 1
        pcs = np.linalg.qr(rng.standard_normal((nfull, k)))[0]
        try:
            basis = kra.Basis(ctx, mt, 15, 3, [("poly", tab)], pcs)
            t0 = time.perf_counter(); K = kra.fit(ctx, basis, snaps)[0]; dt = time.perf_counter() - t0
            t0 = time.perf_counter(); K = kra.fit(ctx, basis, snaps)[0]; dt = time.perf_counter() - t0
            print(mt, "delays=1 poly-%d dim_red: nfull %d N %d W %d fit %.2f ms (gram %.3f)" % (deg, nfull, basis.N, basis.W, dt * 1e3, ctx.timer(0)))
        except Exception as e:
            print(mt, "poly-%d nfull %d ->" % (deg, nfull), repr(e)[:150])
