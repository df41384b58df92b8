import os, sys, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import koopman_realizations_amd as kra
ctx = kra.Context(0)
rng = np.random.default_rng(0); P = rng.standard_normal((2000, 336)); G = P.T @ P; C = rng.standard_normal((336, 336))
for i in range(3):
    ctx.fit_solve(G, C); print("solve ms", ctx.timer(1))
