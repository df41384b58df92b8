import os, sys; sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo")); sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "tests"))
import numpy as np, time, koopman_realizations_amd as kra
from oracle import koopman_oracle as ko
from conftest import synth_pairs
from test_gpu_fit import make_basis
ctx = kra.Context(0)
p = synth_pairs(4000, 6, 3, seed=5)
dic = ko.build_dictionary("linear", 6, 3, ["fourier"], [1])
b = make_basis(ctx, dic)
s = kra.Snapshots(ctx, p["alpha"], p["beta"], p["u"])
Kls = kra.fit(ctx, b, s)[0]
l1 = np.abs(Kls).sum()
print("W", b.W, "N", b.N, "|K_LS|_1", l1, "per N", l1 / b.N)
for frac in (0.9, 0.5, 0.1):
    t = frac * l1 / b.N
    try:
        t0 = time.perf_counter()
        K = kra.fit(ctx, b, s, lasso=[t])[0]
        dt = time.perf_counter() - t0
        G, C = kra.fit_gram(ctx, b, s)
        g = G @ K - C
        supp = K != 0
        theta = np.abs(g[supp]).mean() if supp.any() else 0
        print("t/N %.3g: %.1f ms, |K|_1/budget %.6f, nnz %d, on-support |g| spread %.2e (theta %.3e), off-support max|g|/theta %.4f" % (
            t, dt * 1e3, np.abs(K).sum() / (t * b.N), supp.sum(), np.abs(np.abs(g[supp]) - theta).max(), theta, np.abs(g[~supp]).max() / theta))
    except Exception as e:
        print("t/N %.3g failed:" % t, repr(e)[:300])
