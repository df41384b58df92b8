"""Does the ORDER in which the dual active-set method picks violated rows matter for the cold MPC solves of bench.py
(independent random states)?  CPU only: the oracle's model / QP assembly, a numpy Goldfarb-Idnani with a pluggable selection
rule, iteration counts (full + partial steps, what kp_mpc_last_profile counts) per rule."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from oracle import koopman_oracle as ko

def gi(Hq, f, A, b, select, tol=1e-10, maxit=5000):
    n = Hq.shape[0]
    Hinv = np.linalg.inv(Hq)
    x = -Hinv @ f
    nrm = np.linalg.norm(A, axis=1); safe = np.where(nrm > 0, nrm, 1.0)
    act, lam = [], np.zeros(0)
    it = 0; adds = 0; drops = 0
    while it < maxit:
        it += 1
        viol = (A @ x - b) / safe
        cand = viol.copy(); cand[act] = -np.inf; cand[nrm == 0] = -np.inf
        if cand.max() <= tol:
            return x, act, it, adds, drops
        p = select(cand, tol, act)
        ap = A[p]; lam_p = 0.0
        while it < maxit:
            it += 1
            if act:
                Na = A[act].T; HN = Hinv @ Na
                r = np.linalg.solve(Na.T @ HN, HN.T @ ap); z = Hinv @ ap - HN @ r
            else:
                r = np.zeros(0); z = Hinv @ ap
            t1, l = np.inf, -1
            pos = np.nonzero(r > 1e-13)[0]
            if pos.size:
                ratios = lam[pos] / r[pos]; j = int(np.argmin(ratios)); t1 = ratios[j]; l = int(pos[j])
            apz = ap @ z
            t2 = (ap @ x - b[p]) / apz if apz > 1e-13 * (ap @ Hinv @ ap) else np.inf
            t = min(t1, t2)
            if not np.isfinite(t):
                return None, act, it, adds, drops
            lam = lam - t * r; lam_p += t
            if np.isfinite(t2): x = x - t * z
            if t2 <= t1:
                act.append(p); lam = np.append(lam, lam_p); adds += 1
                break
            act.pop(l); lam = np.delete(lam, l); drops += 1
    return None, act, it, adds, drops

Ns = 20000
a, b_, u = bench.synth_pairs(Ns)
dic = ko.build_dictionary("bilinear", 6, 3, ["poly"], [3])
Px, Py = ko.px_py(dic, {"alpha": a, "beta": b_, "u": u})
K = ko.koopman_ls(Px, Py)
N = dic.N
A_ = K[:N, :N].T.copy(); B_ = K[N:, :N].T.copy()
proj = np.zeros((2, N)); proj[0, 4] = proj[1, 5] = 1.0
u_fac = 2.8
s = ko.MpcSetup("bilinear", A_, B_, 3, 10, proj, 10.0, 100.0, 0.1 * np.array([3e-2, 2e-2, 1e-2]),
                np.stack([np.full(3, -7 * np.pi / 8 / u_fac), np.full(3, 7 * np.pi / 8 / u_fac)], axis=1), 1e-1 * u_fac, None, None, 6)
zeta, u_prev, Yr = bench.mpc_inputs(40)
qps = []
for i in range(40):
    z = ko.econ_full(dic, zeta[i][None, :])[0]
    qps.append(ko.mpc_qp(s, z, u_prev[i], Yr[i].reshape(-1, 2)))
mr = qps[0][2].shape[0]
# time index of every row: the variable with the largest index it touches (x = [u_0; u_1; ...], 3 per step)
Aq = qps[0][2]
tidx = np.array([np.nonzero(Aq[r])[0].max() // 3 if np.any(Aq[r]) else 99 for r in range(mr)])
rules = {
    "most violated (product)": lambda c, tol, act: int(np.argmax(c)),
    "earliest time step, then most violated": lambda c, tol, act: int(min(np.nonzero(c > tol)[0], key=lambda r: (tidx[r], -c[r]))),
    "latest time step, then most violated": lambda c, tol, act: int(min(np.nonzero(c > tol)[0], key=lambda r: (-tidx[r], -c[r]))),
    "least violated": lambda c, tol, act: int(min(np.nonzero(c > tol)[0], key=lambda r: c[r])),
}
for name, rule in rules.items():
    its, adds, drops, bad = [], [], [], 0
    for (Hq, f, Aq, bq) in qps:
        x, act, it, ad, dr = gi(Hq, f, Aq, bq, rule)
        if x is None: bad += 1; continue
        xr, _, ok = ko.qp_solve(Hq, f, Aq, bq)
        assert ok and np.abs(x - xr).max() < 1e-7, np.abs(x - xr).max()
        its.append(it); adds.append(ad); drops.append(dr)
    print(f"{name:42s} iterations mean {np.mean(its):6.1f} (min {min(its)}, max {max(its)})  adds {np.mean(adds):5.1f} drops {np.mean(drops):5.1f} failures {bad}  |active| {np.mean(adds)-np.mean(drops):.1f}")
