import sys, os, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, koopman_realizations_amd as kra
gd = os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "tests", "golden")
g = np.load(gd + "/arm_data.npz")
lens = g["train_len"]; off = np.concatenate([[0], np.cumsum(lens)])
train = [{"t": g["train_t"][a:b], "y": g["train_y"][a:b], "u": g["train_u"][a:b]} for a, b in zip(off[:-1], off[1:])]
val = [{"t": g["val_t"], "y": g["val_y"], "u": g["val_u"]}]
ctx = kra.Context(0)
for delays, deg, mt in ((1, 2, "linear"), (1, 2, "bilinear"), (1, 3, "linear"), (0, 3, "nonlinear")):
    t0 = time.perf_counter()
    try:
        ks = kra.Ksysid({"train": train, "val": val}, ctx=ctx, model_type=mt, obs_type=["poly"], obs_degree=[deg], delays=delays, dim_red=True)
        t1 = time.perf_counter()
        ksh = kra.Ksysid({"train": train, "val": val}, ctx=ctx, model_type=mt, obs_type=["poly"], obs_degree=[deg], delays=delays, dim_red=True, _pca_host=True)
        t2 = time.perf_counter()
        a, b = ks.basis["pcs"], ksh.basis["pcs"]
        print(mt, "delays", delays, "deg", deg, "nfull", a.shape[0], "k", a.shape[1], "host k", b.shape[1], "ctor device pca %.3f s, host pca %.3f s" % (t1 - t0, t2 - t1),
              "max|pcs dev - pcs host|", np.abs(a - b[:, :a.shape[1]]).max() if a.shape[1] <= b.shape[1] else None)
    except Exception as e:
        print(mt, delays, deg, "FAILED", repr(e)[:300])
