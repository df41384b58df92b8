"""Fetching the K matrices of a batch of pipelined fits to the host: pageable destination against a page-locked block of
the context (kp_host_alloc) - the output-side counterpart of tools/upload_probe.py."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes as C
import koopman_realizations_amd as kra
from koopman_realizations_amd import _ffi as F
import bench
ctx = kra.Context(0)
a, b, u = (np.asfortranarray(x) for x in bench.synth_pairs(100000))
basis = kra.Basis(ctx, "bilinear", 6, 3, [("poly", kra.poly_exponent_table(6, 3)[6:])])
s = kra.Snapshots(ctx, a, b, u)
W = basis.W; n = 64
for _ in range(2):
    for i in range(n):
        kra.fit(ctx, basis, s, fetch=False)
    ctx.synchronize()
pag = np.zeros((n, W, W)); pin = ctx.host_array("Kstack", (n, W, W)); pin[:] = 0
for name, dst in (("pageable", pag), ("page-locked", pin), ("pageable", pag), ("page-locked", pin)):
    t0 = time.perf_counter()
    for i in range(n):
        F.check(F.lib().kp_fit_get_K(ctx.handle, i, W, dst[i].ctypes.data_as(F.c_dp)), ctx.handle)
    dt = time.perf_counter() - t0
    print(f"{name}: {n} x K ({dst.nbytes/1e6:.0f} MB) in {dt*1e3:.2f} ms = {dst.nbytes/dt/1e9:.1f} GB/s, {dt/n*1e6:.0f} us per K")
assert np.array_equal(pag, pin)
