"""Ksysid.get_Koopman end to end at config 2 (99 999 pairs, W = 336): where the 1.1 ms go."""
import os, sys, time, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import koopman_realizations_amd as kra, bench
from koopman_realizations_amd import _ffi as F
ctx = kra.Context(0)
a, b, u = bench.synth_pairs(99999)
basis = kra.Basis(ctx, "bilinear", 6, 3, [("poly", kra.poly_exponent_table(6, 3)[6:])])
snaps = kra.Snapshots(ctx, a, b, u)
af, bf, uf = np.asfortranarray(a), np.asfortranarray(b), np.asfortranarray(u)
for name, (x, y, z) in (("C-ordered (row-major) inputs", (a, b, u)), ("column-major inputs", (af, bf, uf))):
    for _ in range(5):
        snaps.update(x, y, z); kra.fit(ctx, basis, snaps)
    tu, tf, tt = [], [], []
    for _ in range(20):
        t0 = time.perf_counter(); snaps.update(x, y, z); t1 = time.perf_counter(); K = kra.fit(ctx, basis, snaps); t2 = time.perf_counter()
        tu.append(t1 - t0); tf.append(t2 - t1); tt.append(t2 - t0)
    print(name, "update %.3f ms, fit+fetch %.3f ms, total %.3f ms" % (np.median(tu) * 1e3, np.median(tf) * 1e3, np.median(tt) * 1e3))
os.environ["KP_NO_ASYNC"] = "1"
t = []
for _ in range(10):
    t0 = time.perf_counter(); kra.fit(ctx, basis, snaps, fetch=False); t.append(time.perf_counter() - t0)
print("fit without fetch (sync) %.3f ms; timers: gram %.3f reduce %.3f solve %.3f" % (np.median(t) * 1e3, ctx.timer(0), ctx.timer(6), ctx.timer(1)))
