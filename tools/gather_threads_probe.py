"""Host gather of the 1024-system population into page-locked blocks: time per quantity against the helper's thread count."""
import os, sys, time, gc, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
chunks = bench.gen_rand_systems(list(range(8)))
gc.collect(); gc.freeze()
import koopman_realizations_amd as kra
from koopman_realizations_amd import sweep, _kp_gather
ctx = kra.Context(0)
mine = [s for c in sorted(chunks) for s in chunks[c]]
tr = [d["train"] for d in mine]
print("cores", len(os.sched_getaffinity(0)))
t0 = time.perf_counter(); arrs = [x["y"] for t in tr for x in t]; tl = time.perf_counter() - t0
ts = [x["t"] for t in tr for x in t]
out = ctx.host_array("probe", (1024, 10010, 1))
outp = np.empty((1024, 10010, 1)); outp[:] = 0
print("list build %.2f ms" % (tl * 1e3))
for nt in (1, 2, 4, 8, 12, 16):
    best = 1e9; bestp = 1e9; bc = 1e9
    for rep in range(4):
        t0 = time.perf_counter(); _kp_gather.gather(arrs, out.ctypes.data, out.nbytes, nt); best = min(best, time.perf_counter() - t0)
        t0 = time.perf_counter(); _kp_gather.gather(arrs, outp.ctypes.data, outp.nbytes, nt); bestp = min(bestp, time.perf_counter() - t0)
        t0 = time.perf_counter(); _kp_gather.trials_increasing(ts, 10, nt); bc = min(bc, time.perf_counter() - t0)
    print("threads %2d: gather 82 MB -> pinned %.2f ms (%.1f GB/s), -> pageable %.2f ms, seam check %.2f ms" % (nt, best * 1e3, 0.082 / best, bestp * 1e3, bc * 1e3))
for rep in range(3):
    t0 = time.perf_counter(); sweep._stack_raw(mine, ctx); print("_stack_raw %.2f ms" % ((time.perf_counter() - t0) * 1e3))
