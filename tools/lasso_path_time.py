"""Lasso fits on the arm data's rank-deficient Gram (bilinear poly-3 without dim_red: W = 336, the homotopy with the inverse in
global memory): best of three wall times per budget.  python tools/lasso_path_time.py"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import koopman_realizations_amd as kra
g = np.load(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "arm_data.npz"))
lens = g["train_len"]; off = np.concatenate([[0], np.cumsum(lens)])
train = [{"t": g["train_t"][a:b], "y": g["train_y"][a:b], "u": g["train_u"][a:b]} for a, b in zip(off[:-1], off[1:])]
data = {"train": train, "val": [{"t": g["val_t"], "y": g["val_y"], "u": g["val_u"]}]}
ctx = kra.Context(0)
import warnings; warnings.simplefilter("ignore")
kb = kra.Ksysid(data, ctx=ctx, model_type="bilinear", obs_type=["poly"], obs_degree=[3], snapshots=np.inf, lasso=[1.0], delays=0, dim_red=False)
sp = kb.snapshotPairs
s = kb._resident_snapshots(sp["alpha"], sp["beta"], sp["u"])
Kls = kra.fit(ctx, kb.basis_dev, s)[0]
N = kb.params["N"]
for f in (0.5, 0.2, 0.1, 0.03):
    las = f * np.abs(Kls).sum() / N
    ts = []
    for rep in range(3):
        t0 = time.perf_counter()
        K = kra.fit(ctx, kb.basis_dev, s, [las])[0]
        ts.append((time.perf_counter() - t0) * 1e3)
    print("budget %.2f |K_LS|_1: %.1f ms (runs %s), nnz %d" % (f, min(ts), " ".join("%.1f" % t for t in ts), (K != 0).sum()))
