"""Host -> HBM path of the snapshot arrays: kp_snapshots_upload (new object), kp_snapshots_update (refill in place) and
fits streamed from host memory through two alternating objects, against the resident-data rate.
Usage: python tools/upload_probe.py [Ns]   (KP_COPY_THREADS / KP_COPY_CHUNK_KB select the staging configuration)"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import koopman_realizations_amd as kra
import bench

Ns = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
ctx = kra.Context(0)
a, b, u = (np.asfortranarray(x) for x in bench.synth_pairs(Ns))     # column-major, as MATLAB hands them over
basis = kra.Basis(ctx, "bilinear", 6, 3, [("poly", kra.poly_exponent_table(6, 3)[6:])])
mb = (a.nbytes + b.nbytes + u.nbytes) / 1e6
t = []
for _ in range(7):
    t0 = time.perf_counter(); s = kra.Snapshots(ctx, a, b, u); t.append(time.perf_counter() - t0); s.close()
print(f"upload (new object, {mb:.1f} MB): median {np.median(t)*1e3:.3f} ms = {mb/np.median(t)/1e3:.1f} GB/s")
ring = [kra.Snapshots(ctx, a, b, u) for _ in range(2)]
t = []
for i in range(9):
    t0 = time.perf_counter(); ring[i % 2].update(a, b, u); t.append(time.perf_counter() - t0)
print(f"update (staged, returns before the DMA ends): median {np.median(t)*1e3:.3f} ms")
for s in ring:
    kra.fit(ctx, basis, s, fetch=False)
ctx.synchronize()
for name, refill in (("resident", False), ("streamed from host memory", True)):
    for rep in range(2):
        n = 64
        t0 = time.perf_counter()
        for i in range(n):
            if refill:
                ring[i % 2].update(a, b, u)
            kra.fit(ctx, basis, ring[i % 2], fetch=False)
        ctx.synchronize()
        dt = time.perf_counter() - t0
    print(f"{name}: {dt/n*1e3:.3f} ms per fit = {Ns*n/dt:.3e} pairs/s")
# sources that are not in the host caches: 16 distinct snapshot matrices (192 MB at the default size) in turn
srcs = [tuple(np.asfortranarray(x + 0.0) for x in (a, b, u)) for _ in range(16)]
n = 64
for rep in range(2):
    t0 = time.perf_counter(); tu = 0.0
    for i in range(n):
        t1 = time.perf_counter(); ring[i % 2].update(*srcs[i % 16]); tu += time.perf_counter() - t1
        kra.fit(ctx, basis, ring[i % 2], fetch=False)
    ctx.synchronize()
    dt = time.perf_counter() - t0
print(f"streamed, 16 rotating host sources: {dt/n*1e3:.3f} ms per fit = {Ns*n/dt:.3e} pairs/s (update call {tu/n*1e3:.3f} ms)")
# one synchronous get_Koopman-like call from host data: refill + fit + K back on the host
t = []
for i in range(7):
    t0 = time.perf_counter(); ring[0].update(a, b, u); K = kra.fit(ctx, basis, ring[0])[0]; t.append(time.perf_counter() - t0)
print(f"update + fit + K fetched: median {np.median(t)*1e3:.3f} ms")
