"""1024 generated systems through sweep.rand_models_sweep_batched, repeated: end-to-end wall time per call."""
import os, sys, time, gc, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
chunks = bench.gen_rand_systems(list(range(8)))
gc.collect(); gc.freeze()
import koopman_realizations_amd as kra
from koopman_realizations_amd import sweep
ctx = kra.Context(0)
mine = [s for c in sorted(chunks) for s in chunks[c]]
sweep.rand_models_sweep_batched(mine, ctx)
t = []
for rep in range(10):
    t0 = time.perf_counter(); tab = sweep.rand_models_sweep_batched(mine, ctx); t.append(time.perf_counter() - t0)
print("ms per call:", " ".join("%.1f" % (x * 1e3) for x in t), "| median %.1f" % (np.median(t) * 1e3))
