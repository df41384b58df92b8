"""Keeps the GPU busy with pipelined fits for N seconds (a stand-in for another tenant on the same device; used to check that
the kernels with cross-workgroup barriers degrade instead of failing)."""
import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, koopman_realizations_amd as kra, bench
secs = float(sys.argv[1]) if len(sys.argv) > 1 else 30.0
ctx = kra.Context(0)
a, b, u = bench.synth_pairs(100000)
basis = kra.Basis(ctx, "bilinear", 6, 3, [("poly", kra.poly_exponent_table(6, 3)[6:])]); snaps = kra.Snapshots(ctx, a, b, u)
ctx.fit_async_slots(64)
t0 = time.time(); n = 0
while time.time() - t0 < secs:
    for _ in range(64): kra.fit(ctx, basis, snaps, fetch=False)
    ctx.synchronize(); n += 64
print("hog: fits", n)
