"""Closed loop (model as plant) with the N=84 synthetic bilinear model of bench.py: steps/s and solver iterations."""
import os, sys, time, ctypes as C
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import koopman_realizations_amd as kra, bench
from koopman_realizations_amd import _ffi as F
ctx = kra.Context(0)
a, b, u = bench.synth_pairs(100000)
basis = kra.Basis(ctx, "bilinear", 6, 3, [("poly", kra.poly_exponent_table(6, 3)[6:])])
snaps = kra.Snapshots(ctx, a, b, u)
mpc, setup = bench.mpc_problem(kra, ctx, basis, snaps)
A, B, N = setup["A"], setup["B"], setup["N"]
T = 300
th = 0.05 * np.arange(T + 12)
ref = np.stack([0.3 * np.cos(th), 0.3 * np.sin(th)], axis=1)
zeta = np.zeros(6); zeta[4], zeta[5] = 0.3, 0.0
up = np.zeros(3)
its = []; acts = []; phases = []; fine = []
t0 = time.perf_counter(); kern = []
for k in range(T):
    Yr = ref[k:k + 11].reshape(-1)
    U, z, st = mpc.step_zeta(basis, zeta, up, Yr)
    assert st == 0, (k, st)
    us = np.zeros(8); cnt = (C.c_int * 2)()
    F.lib().kp_mpc_last_profile(mpc.handle, F.dptr(us), cnt); its.append(cnt[0]); acts.append(cnt[1]); kern.append(us[5]); phases.append(us[:6].copy())
    st16 = np.zeros(16); F.lib().kp_mpc_last_stamps(mpc.handle, F.dptr(st16)); fine.append(st16)
    z1 = A @ z + sum(B[:, i * N:(i + 1) * N] @ z * U[0, i] for i in range(3))
    zeta = z1[:6]; up = U[0]
dt = time.perf_counter() - t0
print("closed loop %d steps: %.1f steps/s (incl. profile readback), kernel us mean %.1f, iterations mean %.1f, active mean %.1f" % (T, T / dt, np.mean(kern), np.mean(its), np.mean(acts)))
print("phase stamps (us, cumulative: lift, S, H/f, inverse+warm setup, QP, end):", np.round(np.mean(phases, axis=0), 1))
fm = np.mean(fine[5:], axis=0)     # (the first step starts cold: no warm-start stamps)
print("stamps (us since kernel start): inputs landed %.1f, lifted %.1f, tracking error %.1f, iteration loop entered %.1f, S %.1f, H/f %.1f, H^-1 %.1f, products %.1f, S^-1 %.1f, solver entered %.1f, solved %.1f" % (fm[10], fm[11], fm[1], fm[6], fm[2], fm[3], fm[12], fm[13], fm[14], fm[4], fm[5]))
