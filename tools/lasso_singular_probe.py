"""Lasso on a rank-deficient Gram (two identical state columns): PSD guard of Ksysid.m:1117-1120 on the device."""
import os, sys, time, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import koopman_realizations_amd as kra
from koopman_realizations_amd import _ffi as F
from oracle import koopman_oracle as ko
ctx = kra.Context(0)
rng = np.random.default_rng(4)
Ns = 4000
a = rng.uniform(-1, 1, (Ns, 3)); a[:, 2] = a[:, 0]
u = rng.uniform(-1, 1, (Ns, 2))
b = np.clip(a + 0.05 * np.tanh(np.hstack([a, u]) @ rng.standard_normal((5, 3))), -1, 1); b[:, 2] = b[:, 0]
basis = kra.Basis(ctx, "bilinear", 3, 2, [("poly", kra.poly_exponent_table(3, 2)[3:])])
snaps = kra.Snapshots(ctx, a, b, u)
G, C = kra.fit_gram(ctx, basis, snaps)
ev = np.linalg.eigvalsh(G); print("W", basis.W, "eig min/max", ev[0], ev[-1], "rank", np.linalg.matrix_rank(G))
Gg = G + 1e-6 * np.eye(basis.W)
Kg = np.linalg.solve(Gg, C); l1 = np.abs(Kg).sum(); print("l1 of guarded LS", l1)
for frac in (2.0, 0.8, 0.3, 0.05):
    t0 = time.perf_counter()
    try:
        K, it = ctx.fit_lasso_batch(G, C, [frac * l1])
        K = K[0]
        print("frac", frac, "iters", it, "ms %.1f" % ((time.perf_counter() - t0) * 1e3), "l1/t", np.abs(K).sum() / (frac * l1),
              "kkt(guarded)/max|C|", ko.lasso_kkt_residual(Gg, C, K, frac * l1) / np.abs(C).max(), "finite", np.isfinite(K).all())
    except F.KoopmanHipError as e:
        print("frac", frac, "error", e, "ms %.1f" % ((time.perf_counter() - t0) * 1e3))
