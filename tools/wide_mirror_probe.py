"""The host mirror end to end on WIDE dictionaries (round 5): example_sysid.m's flow - Ksysid -> train_models -> valNplot-style validation -
with the 728-function fourier dictionary on the arm data (linear W = 738, bilinear W = 2 940), no dim_red."""
import os, sys, time, warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, koopman_realizations_amd as kra
g = np.load(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "arm_data.npz"))
lens = g["train_len"]; off = np.concatenate([[0], np.cumsum(lens)])
train = [{"t": g["train_t"][a:b], "y": g["train_y"][a:b], "u": g["train_u"][a:b]} for a, b in zip(off[:-1], off[1:])]
data = {"train": train, "val": [{"t": g["val_t"], "y": g["val_y"], "u": g["val_u"]}]}
ctx = kra.Context(0)
warnings.simplefilter("ignore")
for mt in ("linear", "bilinear"):
    t0 = time.perf_counter()
    ks = kra.Ksysid(data, ctx=ctx, model_type=mt, obs_type=["fourier"], obs_degree=[1], snapshots=np.inf, lasso=[np.inf], delays=0, dim_red=False)
    t1 = time.perf_counter()
    ks.train_models()
    t2 = time.perf_counter()
    res = ks.valNplot_model() if hasattr(ks, "valNplot_model") else None
    t3 = time.perf_counter()
    err = None
    try:
        err = res[0]["error"]["mean"] if isinstance(res, (list, tuple)) else None
    except Exception:
        pass
    print("%s fourier-1: W %d N %d rank %d; ctor %.1f ms, train_models %.1f ms, validation %.1f ms; A %s B %s; val %s" % (
        mt, ks.basis_dev.W, ks.params["N"], ctx.last_rank(), (t1 - t0) * 1e3, (t2 - t1) * 1e3, (t3 - t2) * 1e3, np.shape(ks.model["A"]), np.shape(ks.model["B"]), err), flush=True)
