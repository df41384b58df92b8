"""Probe: the lasso on the arm data with the bilinear poly-2 dim_red dictionary (cond(G) = 3.5e10), where the FISTA + active-set
solver ran into its iteration cap.  KP_LASSO_TRACE=1 prints the per-block state."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import koopman_realizations_amd as kra

g = np.load(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "arm_data.npz"))
lens = g["train_len"]; off = np.concatenate([[0], np.cumsum(lens)])
train = [{"t": g["train_t"][a:b], "y": g["train_y"][a:b], "u": g["train_u"][a:b]} for a, b in zip(off[:-1], off[1:])]
data = {"train": train, "val": [{"t": g["val_t"], "y": g["val_y"], "u": g["val_u"]}]}
ctx = kra.Context(0)
deg = int(sys.argv[1]) if len(sys.argv) > 1 else 2
dimred = (sys.argv[2] != "0") if len(sys.argv) > 2 else True
mtype = sys.argv[3] if len(sys.argv) > 3 else "bilinear"
kb = kra.Ksysid(data, ctx=ctx, model_type=mtype, obs_type=["poly"], obs_degree=[deg], snapshots=np.inf, lasso=[1.0], delays=0, dim_red=dimred)
sp = kb.get_snapshotPairs() if not hasattr(kb, "snapshotPairs") or kb.snapshotPairs is None else kb.snapshotPairs
s = kb._resident_snapshots(sp["alpha"], sp["beta"], sp["u"])
Kls = kra.fit(ctx, kb.basis_dev, s)[0]
G, C = kra.fit_gram(ctx, kb.basis_dev, s)
print(mtype, deg, dimred, "W", G.shape[0], "N", kb.params["N"], "cond(G) %.3e" % np.linalg.cond(G), "|Kls|_1/N", np.abs(Kls).sum() / kb.params["N"])
for f in ((2.0, 0.5, 0.1, 0.01) if not os.environ.get("KP_DUMP") else ()):
    las = f * np.abs(Kls).sum() / kb.params["N"]
    t0 = time.time()
    try:
        K = kra.fit(ctx, kb.basis_dev, s, [las])[0]
        print("factor", f, "ok  %.1f ms" % ((time.time() - t0) * 1e3), "|K|_1", np.abs(K).sum(), "t", las * kb.params["N"], "nnz", (K != 0).sum())
    except kra.KoopmanHipError as e:
        print("factor", f, "FAILED %.1f ms" % ((time.time() - t0) * 1e3), e)
if os.environ.get("KP_DUMP"):
    os.makedirs("gpurun_out", exist_ok=True)
    np.savez(os.environ["KP_DUMP"], G=G, C=C, Kls=Kls, N=kb.params["N"])
