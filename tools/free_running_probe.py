import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, koopman_realizations_amd as kra
gd = os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "tests", "golden")
g = np.load(gd + "/arm_data.npz"); gp = np.load(gd + "/arm_plant.npz"); r = np.load(gd + "/arm_blockM.npz"); ref = np.load(gd + "/blockM_ref.npz")["y"]
lens = g["train_len"]; off = np.concatenate([[0], np.cumsum(lens)])
train = [{"t": g["train_t"][a:b], "y": g["train_y"][a:b], "u": g["train_u"][a:b]} for a, b in zip(off[:-1], off[1:])]
val = [{"t": g["val_t"], "y": g["val_y"], "u": g["val_u"]}]
ctx = kra.Context(0)
params = {k[2:]: (float(gp[k]) if gp[k].ndim == 0 else gp[k]) for k in gp.files if k.startswith("p_")}
for mt, key in (("bilinear", "bilin"), ("linear", "lin")):
    ks = kra.Ksysid({"train": train, "val": val}, ctx=ctx, model_type=mt, obs_type=["poly"], obs_degree=[3], snapshots=np.inf, lasso=[np.inf], delays=0, dim_red=True).train_models()
    mpc = kra.Kmpc(ks, horizon=10, input_bounds=[], input_slopeConst=1e-1, input_smoothConst=None, state_bounds=[], cost_running=10, cost_terminal=100,
                   cost_input=0.1 * np.array([3e-2, 2e-2, 1e-2]), projmtx=ks.model["C"][-2:, :])
    sim = kra.Ksim(kra.Arm(params, output_type="markers"), mpc)
    res = sim.run_trial_mpc(ref, None, None)
    U, Y = r[key + "_U"], r[key + "_Y"]
    n = min(len(res["U"]), len(U))
    dU = np.abs(res["U"][:n] - U[:n]).max(axis=1); dY = np.abs(res["Y"][:n] - Y[:n]).max(axis=1)
    print(mt, "steps", n, "max|dU| at 10/50/100/200/300:", [float(dU[:k].max()) for k in (10, 50, 100, 200, n)], "max|dY|:", [float(dY[:k].max()) for k in (10, 50, 100, 200, n)], "mean err", float(res["err"].mean()), "stored", float(gp[key + "_err"].mean()))
