"""Where the batched random-system sweep spends its time (host cProfile + device timers)."""
import sys, os, time, cProfile, pstats, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
chunks = bench.gen_rand_systems(list(range(max(1, n // bench.RAND_CHUNK))))
import koopman_realizations_amd as kra
from koopman_realizations_amd import sweep
ctx = kra.Context(0)
systems = [s for c in sorted(chunks) for s in chunks[c]][:n]
sweep.rand_models_sweep_batched(systems[:8], ctx)
t0 = time.perf_counter(); tab = sweep.rand_models_sweep_batched(systems, ctx); print("sweep", n, "systems:", time.perf_counter() - t0, "s")
cProfile.run("sweep.rand_models_sweep_batched(systems, ctx)", "/tmp/sw.prof")
pstats.Stats("/tmp/sw.prof").sort_stats("cumtime").print_stats(18)
