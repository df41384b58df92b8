import os, sys, time, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import koopman_realizations_amd as kra
ctx = kra.Context(0)
rng = np.random.default_rng(0)
for N, m, T in ((4, 1, 1001), (14, 1, 1001), (34, 3, 400), (84, 3, 400)):
    for mt in ("linear", "bilinear"):
        A = 0.5 * np.eye(N) + 0.01 * rng.standard_normal((N, N))
        B = 0.01 * rng.standard_normal((N, N * m if mt == "bilinear" else m))
        z0 = rng.standard_normal(N); U = rng.uniform(-1, 1, (T, m))
        for _ in range(3):
            t0 = time.perf_counter(); Y = ctx.rollout(mt, A[None], B[None], z0[None], U[None], min(N, 6)); dt = time.perf_counter() - t0
        print(mt, "N", N, "T", T, "wall ms %.3f kernel ms %.3f" % (dt * 1e3, ctx.timer(5)))
