"""One synchronous least-squares fit on the arm data WITHOUT dim_red (bilinear poly-3: W = 336, rank 252 - the rank-revealing path with the
basic solution MATLAB's `\\` returns), averaged over 50 calls."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, koopman_realizations_amd as kra
g = np.load(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "arm_data.npz"))
lens = g["train_len"]; off = np.concatenate([[0], np.cumsum(lens)])
train = [{"t": g["train_t"][a:b], "y": g["train_y"][a:b], "u": g["train_u"][a:b]} for a, b in zip(off[:-1], off[1:])]
data = {"train": train, "val": [{"t": g["val_t"], "y": g["val_y"], "u": g["val_u"]}]}
ctx = kra.Context(0)
import warnings; warnings.simplefilter("ignore")
for mt, deg, dr in (("bilinear", 3, False), ("bilinear", 3, True), ("linear", 3, False)):
    ks = kra.Ksysid(data, ctx=ctx, model_type=mt, obs_type=["poly"], obs_degree=[deg], snapshots=np.inf, lasso=[np.inf], delays=0, dim_red=dr)
    sp = ks.snapshotPairs
    s = ks._resident_snapshots(sp["alpha"], sp["beta"], sp["u"])
    for _ in range(5): kra.fit(ctx, ks.basis_dev, s)
    t0 = time.perf_counter()
    for _ in range(50): kra.fit(ctx, ks.basis_dev, s)
    dt = (time.perf_counter() - t0) / 50
    print("%s poly-%d dim_red=%s W %d rank %d: %.3f ms per fit (gram %.3f, solve %.3f)" % (mt, deg, dr, ks.basis_dev.W, ctx.last_rank(), dt * 1e3, ctx.timer(0), ctx.timer(1)))
