"""Debug: replay res_lin teacher-forced, report the first step whose QP fails, dump its QP."""
import os, sys, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import koopman_realizations_amd as kra
g = np.load(os.path.join(ROOT, "tests/golden/arm_data.npz")); r = np.load(os.path.join(ROOT, "tests/golden/arm_blockM.npz"))
refy = np.load(os.path.join(ROOT, "tests/golden/blockM_ref.npz"))["y"]
ctx = kra.Context(0)
lens = g["train_len"]; off = np.concatenate([[0], np.cumsum(lens)])
train = [{"t": g["train_t"][a:b], "y": g["train_y"][a:b], "u": g["train_u"][a:b]} for a, b in zip(off[:-1], off[1:])]
val = [{"t": g["val_t"], "y": g["val_y"], "u": g["val_u"]}]
mt = sys.argv[1] if len(sys.argv) > 1 else "linear"
ks = kra.Ksysid({"train": train, "val": val}, ctx=ctx, model_type=mt, obs_type=["poly"], obs_degree=[3], snapshots=np.inf,
                lasso=[np.inf], delays=0, dim_red=True).train_models()
mpc = kra.Kmpc(ks, horizon=10, input_bounds=[], input_slopeConst=1e-1, input_smoothConst=None, state_bounds=[], cost_running=10,
               cost_terminal=100, cost_input=0.1 * np.array([3e-2, 2e-2, 1e-2]), projmtx=ks.model["C"][-2:, :])
key = "lin" if mt == "linear" else "bilin"
Y, U = r[key + "_Y"], r[key + "_U"]
ref_sc = mpc.scaledown_ref(refy)
step = mpc.get_mpcInput if mt == "linear" else (lambda c, rh: mpc.get_mpcInput_bilinear_iter(c, rh, 1))
nbad = 0
for k in range(299):
    cur = {"y": ks.scaledown_y(Y[k])[None, :], "u": ks.scaledown_u(U[k])[None, :]}
    Uk, z = step(cur, ref_sc[k:k + 11])
    if 236 <= k <= 242:
        import ctypes as C
        from koopman_realizations_amd import _ffi as F
        us = np.zeros(8); cnt = (C.c_int * 2)()
        F.lib().kp_mpc_last_profile(mpc.dev.handle, F.dptr(us), cnt)
        print("k", k, "iterations", cnt[0], "active", cnt[1], "nan" if np.isnan(Uk).any() else "ok", np.round(Uk[1], 6))
    if np.isnan(Uk).any():
        nbad += 1
        if nbad == 1:
            Hq, f, Aq, bq = mpc.dev.last_qp()
            ev = np.linalg.eigvalsh(Hq)
            print("first failure at k", k, "eig(H) min/max", ev[0], ev[-1], "cond", ev[-1] / ev[0])
            np.savez(os.path.join(ROOT, "gpurun_out/mpc_nan_qp.npz"), H=Hq, f=f, A=Aq, b=bq)
print("failed steps:", nbad)
