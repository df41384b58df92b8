import os, sys; sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, koopman_realizations_amd as kra, bench
ctx = kra.Context(0)
a, b, u = bench.synth_pairs(100000)
tab = kra.poly_exponent_table(6, 3)
pcs = np.linalg.qr(np.random.default_rng(0).standard_normal((84, 27)))[0]
basis = kra.Basis(ctx, "bilinear", 6, 3, [("poly", tab[6:])], pcs)
snaps = kra.Snapshots(ctx, a, b, u)
for _ in range(3): kra.fit_gram(ctx, basis, snaps, fetch=False)
