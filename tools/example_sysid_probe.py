"""The three models of the reference's example_sysid.m (linear / bilinear / nonlinear, poly-3, dim_red) on the shipped arm data
through the host mirror: wall time of the constructor (scaling, pairs, pca + econ lift), train_models and one get_Koopman."""
import os, sys, time, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import koopman_realizations_amd as kra
gd = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")
g = np.load(os.path.join(gd, "arm_data.npz"))
lens = g["train_len"]; off = np.concatenate([[0], np.cumsum(lens)])
train = [{"t": g["train_t"][a:b], "y": g["train_y"][a:b], "u": g["train_u"][a:b]} for a, b in zip(off[:-1], off[1:])]
val = [{"t": g["val_t"], "y": g["val_y"], "u": g["val_u"]}]
ctx = kra.Context(0)
for mt in ("linear", "bilinear", "nonlinear"):
    for rep in range(2):
        t0 = time.perf_counter()
        ks = kra.Ksysid({"train": train, "val": val}, ctx=ctx, model_type=mt, obs_type=["poly"], obs_degree=[3], snapshots=np.inf,
                        lasso=[np.inf], delays=0, dim_red=True)
        t1 = time.perf_counter()
        ks.train_models()
        t2 = time.perf_counter()
        for _ in range(3):
            ks.get_Koopman(ks.snapshotPairs)
        t3 = time.perf_counter()
        res = ks.valNplot_model() if hasattr(ks, "valNplot_model") else None
        t4 = time.perf_counter()
    print(f"{mt:9s}: N {ks.params['N']:3d}  ctor {1e3*(t1-t0):7.1f} ms  train_models {1e3*(t2-t1):6.2f} ms  get_Koopman {1e3*(t3-t2)/3:6.3f} ms  validation {1e3*(t4-t3):6.1f} ms")
