"""dim_red Gram kernel (kp_gram3_kernel<.,.,true,NW>): kernel time in a pipelined queue + parity against the numpy oracle.
KP_GRAM3_PCS_NW=8|12 / KP_GRAM3_PCS_NW4=1 select the workgroup shape."""
import sys, os, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import koopman_realizations_amd as kra, bench
from oracle import koopman_oracle as ko
ctx = kra.Context(0)
Ns = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
tab = kra.poly_exponent_table(6, 3)
for k in (27, 16, 32):
    pcs = np.linalg.qr(np.random.default_rng(3).standard_normal((84, k)))[0]
    a, b, u = bench.synth_pairs(Ns, seed=5)
    basis = kra.Basis(ctx, "bilinear", 6, 3, [("poly", tab[6:])], pcs)
    snaps = kra.Snapshots(ctx, a, b, u)
    for _ in range(48):
        kra.fit(ctx, basis, snaps, fetch=False)
    ctx.synchronize()
    for _ in range(64):
        kra.fit(ctx, basis, snaps, fetch=False)
    ctx.synchronize()
    ms = ctx.timer(0)
    a2, b2, u2 = bench.synth_pairs(3001, seed=2)
    s2 = kra.Snapshots(ctx, a2, b2, u2)
    G, C = kra.fit_gram(ctx, basis, s2)
    dic = ko.Dictionary("bilinear", 6, 3, ko.make_basis(6, ["poly"], [3]), pcs)
    Px, Py = ko.px_py(dic, {"alpha": a2, "beta": b2, "u": u2})
    print(f"k_pcs {k} W {basis.W}: gram {ms:.4f} ms, executed {ctx.timer(10):.0f} flop/pair; parity G",
          np.abs(G - Px.T @ Px).max() / np.abs(G).max(), "C", np.abs(C - Px.T @ Py).max() / np.abs(C).max())
