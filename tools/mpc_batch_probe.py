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
