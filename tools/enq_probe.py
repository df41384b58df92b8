"""Why is the host slow to enqueue fits after bench.py generated its random systems?  mode: none | pool | alloc | pool_small"""
import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import bench
mode = sys.argv[1] if len(sys.argv) > 1 else "none"
keep = None
if mode == "pool":
    keep = bench.gen_rand_systems(list(range(8)))
elif mode == "pool_small":
    keep = bench.gen_rand_systems([0, 1])
elif mode == "serial":
    keep = {c: bench._rand_chunk(c) for c in range(2)}
elif mode == "alloc":
    keep = [np.random.default_rng(i).standard_normal((1001, 3)) for i in range(12000)]
elif mode == "mp":
    import multiprocessing as mp
    with mp.get_context("fork").Pool(8) as pool:
        pool.map(abs, range(64))
import gc; gc.collect(); gc.freeze()
import koopman_realizations_amd as kra
ctx = kra.Context(0)
a, b, u = bench.synth_pairs(100000)
basis = kra.Basis(ctx, "bilinear", 6, 3, [("poly", kra.poly_exponent_table(6, 3)[6:])]); snaps = kra.Snapshots(ctx, a, b, u)
ctx.fit_async_slots(200)
for _ in range(64): kra.fit(ctx, basis, snaps, fetch=False)
ctx.synchronize()
t0 = time.perf_counter()
for _ in range(200): kra.fit(ctx, basis, snaps, fetch=False)
t1 = time.perf_counter(); ctx.synchronize(); t2 = time.perf_counter()
print(mode, "enqueue ms/step %.4f  total ms/step %.4f" % ((t1 - t0) / 200 * 1e3, (t2 - t0) / 200 * 1e3), "threads", len(os.listdir("/proc/self/task")), "affinity", len(os.sched_getaffinity(0)))
