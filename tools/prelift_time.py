"""Kernel times of the dim_red (W = 136) fit: kp_gram3_prelift_kernel + kp_gram3_kernel<.,3,false,false,true> (default) or the in-kernel
projection (KP_GRAM3_NO_PRELIFT=1).  Run under rocprofv3 --kernel-trace --stats."""
import os, sys; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, koopman_realizations_amd as kra, bench
ctx = kra.Context(0)
a, b, u = bench.synth_pairs(100000)
tab = kra.poly_exponent_table(6, 3)
pcs = np.linalg.qr(np.random.default_rng(0).standard_normal((84, 27)))[0]
basis = kra.Basis(ctx, "bilinear", 6, 3, [("poly", tab[6:])], pcs)
snaps = kra.Snapshots(ctx, a, b, u)
for rep in range(2):
    for _ in range(64): kra.fit(ctx, basis, snaps, fetch=False)
    ctx.synchronize(); print("W136 pcs gram ms", ctx.timer(0))
