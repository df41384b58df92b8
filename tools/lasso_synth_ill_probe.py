"""GPU probe: the ill-conditioned synthetic lasso point of bench.py (bench_lasso_ill_conditioned)."""
import os, sys, time, json
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import koopman_realizations_amd as kra
import bench
ctx = kra.Context(0)
noise = float(sys.argv[1]) if len(sys.argv) > 1 else 1e-3
rng = np.random.default_rng(11)
Ns = 12000
ts = np.linspace(0.0, 60.0, Ns + 1)
lat = np.stack([np.sin(0.9 * ts), np.cos(0.37 * ts + 0.4)], 1)
Y = 0.8 * lat @ rng.uniform(-1, 1, (2, 6)) + noise * rng.standard_normal((Ns + 1, 6))
u = rng.uniform(-1, 1, (Ns, 3))
exps = kra.poly_exponent_table(6, 2)[6:]
b = kra.Basis(ctx, "bilinear", 6, 3, [("poly", exps)])
s_ = kra.Snapshots(ctx, np.ascontiguousarray(Y[:-1]), np.ascontiguousarray(Y[1:]), u)
G, C = kra.fit_gram(ctx, b, s_)
ev = np.linalg.eigvalsh((G + G.T) / 2)
print("W", b.W, "eig %.3e .. %.3e" % (ev[0], ev[-1]), "rank", ctx.last_rank() if hasattr(ctx, "last_rank") else "")
Kls = kra.fit(ctx, b, s_)[0]
l1 = float(np.abs(Kls).sum())
print("|Kls|_1", l1, "rank", ctx.last_rank())
if os.environ.get("KP_DUMP"):
    np.savez(os.environ["KP_DUMP"], G=G, C=C, Kls=Kls)
for f in (0.5, 0.1):
    t0 = time.time()
    try:
        K = kra.fit(ctx, b, s_, [f * l1 / b.N])[0]
        print("factor", f, "ok %.1f ms" % ((time.time() - t0) * 1e3), "|K|_1/t", np.abs(K).sum() / (f * l1), "nnz", (K != 0).sum(), "homotopy ms", ctx.timer(11))
    except kra.KoopmanHipError as e:
        print("factor", f, "FAILED %.1f ms" % ((time.time() - t0) * 1e3), e)
