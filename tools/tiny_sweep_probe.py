import sys, numpy as np
sys.path.insert(0, '/root/repo')
import koopman_realizations_amd as kra
from koopman_realizations_amd.device import Traj, Basis
ctx = kra.Context(0)
rng = np.random.default_rng(0)
for (nb, k, T) in ((2, 1, 3), (1, 1, 130), (2, 3, 45), (2, 2, 129)):
    Y = rng.uniform(-1, 1, (nb, k * T, 1)); U = rng.uniform(-1, 1, (nb, k * T, 1))
    Yv = rng.uniform(-1, 1, (nb, 20, 1)); Uv = rng.uniform(-1, 1, (nb, 20, 1))
    traj = Traj(ctx, Y, U, k, Yv, Uv)
    for mt, D in (("linear", 3), ("bilinear", 2), ("nonlinear", 2)):
        nv = 1 + (1 if mt == "nonlinear" else 0)
        b = Basis(ctx, mt, 1, 1, [("poly", kra.poly_exponent_table(nv, D)[nv:])], None)
        err, st = traj.sweep_eval_nested(b, D, np.inf)
        print((nb, k, T), mt, "status", st.ravel().tolist(), "finite", np.isfinite(err).ravel().tolist())
        b.close()
    traj.close()
print("done")
