import sys, os, numpy as np
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import koopman_realizations_amd as kra
from conftest import synth_pairs
ctx = kra.Context(0)
pcs = np.linalg.qr(np.random.default_rng(0).standard_normal((84, 27)))[0]
p = synth_pairs(2000003, seed=2)
b = kra.Basis(ctx, "bilinear", 6, 3, [("poly", kra.poly_exponent_table(6, 3)[6:])], pcs)
s = kra.Snapshots(ctx, p["alpha"], p["beta"], p["u"])
G, C = kra.fit_gram(ctx, b, s)
print("gram ms", ctx.timer(0), "G[0,0]", G[0, 0], "sum", G.sum(), C.sum())
np.save(sys.argv[1], np.concatenate([G.ravel(), C.ravel()]))
