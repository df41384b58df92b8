import sys, time, numpy as np
sys.path.insert(0, '.')
import koopman_realizations_amd as kra
from koopman_realizations_amd import sweep
g = np.load('tests/golden/rand_systems.npz')
def system(i):
    t, y, u = g[f"s{i}_train_t"], g[f"s{i}_train_y"], g[f"s{i}_train_u"]
    n = t.shape[0] // 1001
    train = [{"t": t[k*1001:(k+1)*1001], "y": y[k*1001:(k+1)*1001], "u": u[k*1001:(k+1)*1001]} for k in range(n)]
    return {"train": train, "val": [{"t": g[f"s{i}_val_t"], "y": g[f"s{i}_val_y"], "u": g[f"s{i}_val_u"]}]}
ctx = kra.Context(0)
d = system(0)
sweep.eval_system(d, ctx=ctx)
t0 = time.perf_counter(); r = sweep.eval_system(d, ctx=ctx); dt = time.perf_counter() - t0
print("eval_system (23 fits + rollouts): %.3f s" % dt)
import cProfile, pstats
pr = cProfile.Profile(); pr.enable(); sweep.eval_system(d, ctx=ctx); pr.disable()
pstats.Stats(pr).sort_stats('cumulative').print_stats(14)
