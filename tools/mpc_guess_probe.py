"""How good is 'rows violated by the unconstrained optimum' as a first active set of the MPC QP?  For the independent random
states of bench.py: size of the optimal active set F, of the guess G0 (rows with a_i'x0 > b_i), of the previous problem's set
P; rows to release / to add from each start; rank deficiency of the guess."""
import os, sys, ctypes as C
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import koopman_realizations_amd as kra, bench
from koopman_realizations_amd import _ffi as F
ctx = kra.Context(0)
a, b, u = bench.synth_pairs(100000)
basis = kra.Basis(ctx, "bilinear", 6, 3, [("poly", kra.poly_exponent_table(6, 3)[6:])])
snaps = kra.Snapshots(ctx, a, b, u)
mpc, setup = bench.mpc_problem(kra, ctx, basis, snaps)
zeta, u_prev, Yr = bench.mpc_inputs(40)
P = set()
for i in range(40):
    U, z, st = mpc.step_zeta(basis, zeta[i], u_prev[i], Yr[i])
    us = np.zeros(8); cnt = (C.c_int * 2)()
    F.lib().kp_mpc_last_profile(mpc.handle, F.dptr(us), cnt)
    H, f, A, bq = mpc.last_qp()
    x = U.reshape(-1)                     # [u_0; u_1; ...]
    x0 = -np.linalg.solve(H, f)
    r = A @ x - bq
    Fset = set(np.nonzero(np.abs(r) < 1e-9)[0].tolist())
    v0 = A @ x0 - bq
    nrm = np.linalg.norm(A, axis=1)
    G0 = set(np.nonzero(v0 > 1e-9 * np.maximum(1, nrm))[0].tolist())
    rank = np.linalg.matrix_rank(A[sorted(G0)]) if G0 else 0
    # greedy independent subset by decreasing violation
    order = sorted(G0, key=lambda j: -v0[j] / max(nrm[j], 1e-300))
    sel = []
    for j in order:
        if np.linalg.matrix_rank(A[sel + [j]]) == len(sel) + 1: sel.append(j)
    S = set(sel)
    print(f"{i:2d} st {st} iters {cnt[0]:3d} kernel {us[5]:6.1f} us |F| {len(Fset):2d} |G0| {len(G0):2d} rank {rank:2d} "
          f"G0: release {len(G0 - Fset):2d} add {len(Fset - G0):2d} | indep subset {len(S):2d}: release {len(S - Fset):2d} add {len(Fset - S):2d} "
          f"| prev P {len(P):2d}: release {len(P - Fset):2d} add {len(Fset - P):2d}")
    P = Fset
