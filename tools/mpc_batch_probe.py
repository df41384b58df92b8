"""Wall and kernel time of kp_mpc_step_batch (4096 problems of the N = 84 bilinear controller)."""
import sys, os, time, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import koopman_realizations_amd as kra, bench
from koopman_realizations_amd import _ffi as F
ctx = kra.Context(0)
a, b, u = bench.synth_pairs(100000)
basis = kra.Basis(ctx, "bilinear", 6, 3, [("poly", kra.poly_exponent_table(6, 3)[6:])]); snaps = kra.Snapshots(ctx, a, b, u)
mpc, setup = bench.mpc_problem(kra, ctx, basis, snaps)
nb = 4096
zeta, u_prev, Yr = bench.mpc_inputs(nb)
Z = basis.lift(F.LIFT_ECON, zeta)
for i in range(5):
    t0 = time.perf_counter(); U, st = mpc.step_batch(Z, u_prev, Yr); dt = time.perf_counter() - t0
    print("step_batch wall %.3f ms kernel %.3f ms solved %d" % (dt * 1e3, ctx.timer(2), int((st == 0).sum())))
# where the wall time goes: the C call alone (ctypes, arrays prepared) against the Python wrapper
import ctypes as C
Zc = np.ascontiguousarray(Z); UP = np.ascontiguousarray(u_prev); YR = np.ascontiguousarray(Yr)
U = np.zeros((nb, mpc.m, mpc.Np)); st = np.zeros(nb, dtype=np.int32)
for i in range(4):
    t0 = time.perf_counter()
    F.lib().kp_mpc_step_batch(mpc.handle, nb, F.dptr(Zc), F.dptr(UP), F.dptr(YR), F.dptr(U), st.ctypes.data_as(F.c_ip))
    dt = time.perf_counter() - t0
    print("C call alone %.3f ms, kernel %.3f ms" % (dt * 1e3, ctx.timer(2)))
