"""GPU probe: the rank-deficient Gram of tests/test_gpu_edges.py (two identical state columns) through the homotopy."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import koopman_realizations_amd as kra
ctx = kra.Context(0)
rng = np.random.default_rng(4)
Ns = 4000
a = rng.uniform(-1, 1, (Ns, 3)); a[:, 2] = a[:, 0]
u = rng.uniform(-1, 1, (Ns, 2))
b_ = np.clip(a + 0.05 * np.tanh(np.hstack([a, u]) @ rng.standard_normal((5, 3))), -1, 1); b_[:, 2] = b_[:, 0]
basis = kra.Basis(ctx, "bilinear", 3, 2, [("poly", kra.poly_exponent_table(3, 2)[3:])])
snaps = kra.Snapshots(ctx, a, b_, u)
G, C = kra.fit_gram(ctx, basis, snaps)
W = basis.W
Gg = G + 1e-6 * np.eye(W)
Kg = np.linalg.solve(Gg, C)
l1 = np.abs(Kg).sum()
Ks, its = ctx.fit_lasso_batch(G, C, [2.0 * l1, 0.8 * l1])
print("iters", its, "homotopy ms", ctx.timer(11))
K = Ks[0]
print("max diff", np.abs(K - Kg).max(), "nnz", (K != 0).sum(), "of", K.size, "|K|_1", np.abs(K).sum(), l1)
g = Gg @ K - C
z = np.argwhere(K == 0)
print("zeros", len(z), z[:12].tolist())
print("|g| at zeros", np.abs(g[K == 0])[:12], "max |g| on support", np.abs(g[K != 0]).max())
for j in sorted(set(z[:, 1].tolist()))[:3]:
    print("column", j, "zero rows", np.flatnonzero(K[:, j] == 0), "Kg there", Kg[K[:, j] == 0, j])
