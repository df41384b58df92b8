import os, sys
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import numpy as np
import koopman_realizations_amd as kra
from oracle import koopman_oracle as ko
rng = np.random.default_rng(7)
P = rng.standard_normal((2000, 12))
P[:, 7] = P[:, 2] * (1 + 1e-7 * rng.standard_normal(2000))
Y = P @ (rng.standard_normal((12, 12)) * (rng.random((12, 12)) < 0.5)) + 0.01 * rng.standard_normal((2000, 12))
G, C = P.T @ P, P.T @ Y
print("eig", np.linalg.eigvalsh(G)[:2], "cond %.2e" % np.linalg.cond(G))
c = kra.Context(0)
l1 = np.abs(c.fit_solve(G, C)).sum()
print("l1", l1)
c = kra.Context(0)
K, it = c.fit_lasso_batch(G, C, [0.6 * l1, 0.2 * l1])
print("iters", it, "ms", c.timer(11), [np.abs(k).sum() / t for k, t in zip(K, [0.6 * l1, 0.2 * l1])])
