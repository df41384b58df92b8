"""example_control.m end to end with the true Arm plant; compares with the stored closed loops."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import koopman_realizations_amd as kra

G = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")
g = np.load(os.path.join(G, "arm_data.npz")); gp = np.load(os.path.join(G, "arm_plant.npz"))
gb = np.load(os.path.join(G, "arm_blockM.npz")); ref = np.load(os.path.join(G, "blockM_ref.npz"))["y"]
lens = g["train_len"]; off = np.concatenate([[0], np.cumsum(lens)])
train = [{"t": g["train_t"][a:b], "y": g["train_y"][a:b], "u": g["train_u"][a:b]} for a, b in zip(off[:-1], off[1:])]
val = [{"t": g["val_t"], "y": g["val_y"], "u": g["val_u"]}]
params = {k[2:]: (float(gp[k]) if gp[k].ndim == 0 else gp[k]) for k in gp.files if k.startswith("p_")}
arm = kra.Arm(params, output_type="markers")
ctx = kra.Context(0)
for mt in ("bilinear", "linear"):
    ks = kra.Ksysid({"train": train, "val": val}, ctx=ctx, model_type=mt, obs_type=["poly"], obs_degree=[3],
                    snapshots=np.inf, lasso=[np.inf], delays=0, dim_red=True).train_models()
    mpc = kra.Kmpc(ks, horizon=10, input_bounds=[-7 * np.pi / 8, 7 * np.pi / 8], input_slopeConst=1e-1, input_smoothConst=None,
                   state_bounds=None, cost_running=10, cost_terminal=100, cost_input=0.1 * np.array([3e-2, 2e-2, 1e-2]),
                   projmtx=ks.model["C"][-2:, :])
    t0 = time.time()
    res = kra.Ksim(arm, mpc).run_trial_mpc(ref, None, None)
    key = "bilin" if mt == "bilinear" else "lin"
    n = res["Y"].shape[0]
    print(mt, "steps", n, "wall", time.time() - t0, "mean err", res["err"].mean(), "stored", gp[key + "_err"].mean(),
          "max|U-Ustored|", np.abs(res["U"] - gb[key + "_U"][:n]).max(), "max|Y-Ystored|", np.abs(res["Y"] - gb[key + "_Y"][:n]).max(),
          "comp_time mean", res["comp_time"].mean(), "stored comp_time", gb[key + "_comp_time"].mean())
    for k in (1, 2, 5, 20, 100, 299):
        if k < n: print("  k", k, np.abs(res["U"][k] - gb[key + "_U"][k]).max(), np.abs(res["Y"][k] - gb[key + "_Y"][k]).max())
