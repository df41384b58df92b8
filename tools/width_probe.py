"""Fused lift+Gram kernel time at SURVEY 8(d)'s three fit shapes (W = 336, 200, 136 with pcs) + parity of the pcs path."""
import sys, os, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import koopman_realizations_amd as kra, bench
from oracle import koopman_oracle as ko
ctx = kra.Context(0)
Ns = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
print(bench.bench_width_points(ctx, kra, Ns))
# parity of the econ path against the numpy oracle on a small sample
a, b, u = bench.synth_pairs(3000, seed=2)
tab = kra.poly_exponent_table(6, 3)
pcs = np.linalg.qr(np.random.default_rng(3).standard_normal((84, 27)))[0]
basis = kra.Basis(ctx, "bilinear", 6, 3, [("poly", tab[6:])], pcs)
snaps = kra.Snapshots(ctx, a, b, u)
G, C = kra.fit_gram(ctx, basis, snaps)
dic = ko.Dictionary("bilinear", 6, 3, ko.make_basis(6, ["poly"], [3]), pcs)
Px, Py = ko.px_py(dic, {"alpha": a, "beta": b, "u": u})
print("pcs Gram parity: G", np.abs(G - Px.T @ Px).max() / np.abs(G).max(), "C", np.abs(C - Px.T @ Py).max() / np.abs(C).max())
