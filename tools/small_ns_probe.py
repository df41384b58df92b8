"""Gram kernel + reduction time against the snapshot count (the shipped arm example has 11 999 pairs): how far below
linear scaling do small fits fall?  Usage: python tools/small_ns_probe.py"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import koopman_realizations_amd as kra
import bench
ctx = kra.Context(0)
basis = kra.Basis(ctx, "bilinear", 6, 3, [("poly", kra.poly_exponent_table(6, 3)[6:])])
for Ns in (1000, 3000, 11999, 30000, 100000, 300000):
    a, b, u = (np.asfortranarray(x) for x in bench.synth_pairs(Ns))
    s = kra.Snapshots(ctx, a, b, u)
    for _ in range(80):
        kra.fit_gram(ctx, basis, s, fetch=False)
    ctx.synchronize()
    g, r = [], []
    for _ in range(20):
        kra.fit_gram(ctx, basis, s, fetch=False); ctx.synchronize()
        g.append(ctx.timer(0)); r.append(ctx.timer(6))
    t0 = time.perf_counter()
    for _ in range(50):
        kra.fit(ctx, basis, s, fetch=False)
    ctx.synchronize()
    dt = (time.perf_counter() - t0) / 50
    print(f"Ns {Ns:7d}: gram kernel {np.median(g)*1e3:8.1f} us (+ reduce {np.median(r)*1e3:5.1f}), pipelined fit {dt*1e6:8.1f} us = {Ns/dt:.3e} pairs/s; "
          f"linear-scaling kernel time from 1e5: {408.0*Ns/1e5:7.1f} us")
    s.close()
