"""Timing of the batched random-system sweep (evaluate_rand_models.m shape) for nb systems on one GPU."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import koopman_realizations_amd as kra
from koopman_realizations_amd import sweep
g = np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "rand_systems.npz"))
def system(i):
    t, y, u = g[f"s{i}_train_t"], g[f"s{i}_train_y"], g[f"s{i}_train_u"]
    n = t.shape[0] // 1001
    train = [{"t": t[k*1001:(k+1)*1001], "y": y[k*1001:(k+1)*1001], "u": u[k*1001:(k+1)*1001]} for k in range(n)]
    return {"train": train, "val": [{"t": g[f"s{i}_val_t"], "y": g[f"s{i}_val_y"], "u": g[f"s{i}_val_u"]}]}
ctx = kra.Context(0)
base = [system(i) for i in range(3)]
for nb in (3, 64, 1024):
    systems = [base[i % 3] for i in range(nb)]
    sweep.rand_models_sweep_batched(systems[:3], ctx)
    t0 = time.perf_counter(); tab = sweep.rand_models_sweep_batched(systems, ctx); dt = time.perf_counter() - t0
    print("nb %4d batched sweep %.3f s  (%.2f ms per system)  linear deg-13 mean err %.4f" % (nb, dt, dt / nb * 1e3, np.nanmean(tab["linear"][-1])))
import cProfile, pstats
systems = [base[i % 3] for i in range(1024)]
pr = cProfile.Profile(); pr.enable(); sweep.rand_models_sweep_batched(systems, ctx); pr.disable()
pstats.Stats(pr).sort_stats('cumulative').print_stats(22)
