"""[KP_CHOL_PROF=1] python tools/chol_prof.py [W]: the event-timed solve of one W x W system; with KP_CHOL_PROF=1 the
Cholesky kernel prints its cycles per phase."""
import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, koopman_realizations_amd as kra
W = int(sys.argv[1]) if len(sys.argv) > 1 else 336
rng = np.random.default_rng(0)
P = rng.standard_normal((4 * W, W)); G = P.T @ P; C = P.T @ rng.standard_normal((4 * W, W))
ctx = kra.Context(0)
for _ in range(3):
    K = ctx.fit_solve(G, C)
t0 = time.perf_counter()
for _ in range(20):
    K = ctx.fit_solve(G, C)
print("W", W, "fit_solve wall ms (incl. H2D/D2H of 3 W^2)", (time.perf_counter() - t0) / 20 * 1e3, "solve timer ms", ctx.timer(1),
      "err", np.abs(K - np.linalg.solve(G, C)).max())
