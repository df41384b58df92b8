"""Kernel times of the fourier / gaussian fits of bench.bench_width_points (kp_gram3_prelift_ext_kernel + kp_gram3_kernel<.,3,false,false,true>).
Run under rocprofv3 --kernel-trace --stats."""
import os, sys; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, koopman_realizations_amd as kra, bench
ctx = kra.Context(0)
a, b, u = bench.synth_pairs(100000, seed=5)
snaps = kra.Snapshots(ctx, a, b, u)
centres = np.random.default_rng(3).uniform(-1, 1, (6, 20))
basis = kra.Basis(ctx, "bilinear", 6, 3, [("gaussian", centres)])
for rep in range(2):
    for _ in range(64): kra.fit(ctx, basis, snaps, fetch=False)
    ctx.synchronize(); print("gaussian20 gram ms", ctx.timer(0))
