"""How robust is the device active-set QP at the (often primal-degenerate) vertices of the stored MATLAB runs?
Replays res_lin / res_bilin teacher-forced, takes every step's QP from the device (kp_mpc_last_qp) and solves it COLD
through kp_qp_solve as it is and with H, f perturbed at the 1e-12 ... 1e-8 relative level (what another rounding order of
the assembly, or a less accurate inverse, would do; b is left alone: the pinned first input is a pair of opposite
inequalities, Kmpc.m:865-870, which independent noise turns into a genuinely infeasible pair); counts failures and the worst distance to the oracle's exact optimum on a sample.
Usage: python tools/qp_robustness_probe.py [linear|bilinear]"""
import os, sys, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import koopman_realizations_amd as kra
from oracle import koopman_oracle as ko
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
rng = np.random.default_rng(0)
fails = {0.0: 0, 1e-12: 0, 1e-10: 0, 1e-8: 0}; total = 0; worst = 0.0; warm_fail = 0; degenerate = 0
for k in range(299):
    cur = {"y": ks.scaledown_y(Y[k])[None, :], "u": ks.scaledown_u(U[k])[None, :]}
    Uk, z = step(cur, ref_sc[k:k + 11])
    warm_fail += int(np.isnan(Uk).any())
    H, f, A, b = mpc.dev.last_qp()
    total += 1
    for eps in fails:
        for rep in range(1 if eps == 0.0 else 3):
            fp = f * (1 + eps * rng.standard_normal(f.shape))
            E = eps * rng.standard_normal(H.shape) * np.sqrt(np.outer(np.diag(H), np.diag(H)))
            x, st = ctx.qp_solve(H + 0.5 * (E + E.T), fp, A, b)
            if st != 0 or np.isnan(x).any():
                fails[eps] += 1
            elif eps == 0.0 and k % 10 == 0:
                xo, lam, ok = ko.qp_solve(H, f, A, b)
                worst = max(worst, np.abs(x - xo).max())
                degenerate += int((np.abs(A @ xo - b) < 1e-9).sum() > H.shape[0])
print(f"{mt}: {total} QPs; warm-started step failures {warm_fail}; cold failures as is {fails[0.0]}, "
      f"perturbed 1e-12: {fails[1e-12]}, 1e-10: {fails[1e-10]}, 1e-8: {fails[1e-8]} of {3 * total} each; worst |x - oracle| on every 10th: {worst:.2e}; "
      f"sampled QPs with more tight rows than variables: {degenerate} of 30")
