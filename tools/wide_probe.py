"""Wide dictionaries (W > 512: lifted panels + TN products, blocked Cholesky; csrc/kp_wide.hip, kp_fit.hip): time of the Gram
pass and of one synchronous fit for the fourier dictionary of `def_fourierLift` on six states (Ksysid.m:694-731; linear W = 738,
bilinear W = 2 940) at the arm data set's size and at 1e5 pairs, and of the solve alone on random SPD systems."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, koopman_realizations_amd as kra
from oracle import koopman_oracle as ko
from conftest import synth_pairs
from test_gpu_fit import make_basis
ctx = kra.Context(0)
for mt in ("linear", "bilinear"):
    dic = ko.build_dictionary(mt, 6, 3, ["fourier"], [1])
    b = make_basis(ctx, dic)
    for Ns in (11999, 100000):
        p = synth_pairs(Ns, 6, 3, seed=1)
        s = kra.Snapshots(ctx, p["alpha"], p["beta"], p["u"])
        for _ in range(2): kra.fit_gram(ctx, b, s, fetch=False)
        t0 = time.perf_counter(); reps = 5
        for _ in range(reps): kra.fit_gram(ctx, b, s, fetch=False)
        tg = (time.perf_counter() - t0) / reps
        gk = ctx.timer(0)
        for _ in range(2): kra.fit(ctx, b, s, fetch=False)
        t0 = time.perf_counter()
        for _ in range(reps): kra.fit(ctx, b, s, fetch=False)
        tf = (time.perf_counter() - t0) / reps
        F = b.W * (b.W + 1) + 2.0 * b.W * b.W
        print("%s fourier-1 on 6 states W %d Ns %d: gram %.3f ms (device %.3f ms = %.1f TFLOP/s dense-equivalent), fit %.3f ms (solve %.3f ms), rank %d"
              % (mt, b.W, Ns, tg * 1e3, gk, F * Ns / (gk * 1e-3) / 1e12, tf * 1e3, ctx.timer(1), ctx.last_rank()), flush=True)
        s.close() if hasattr(s, "close") else None
rng = np.random.default_rng(0)
for W in (738, 1472, 2940):
    A = rng.standard_normal((W + 64, W)); G = A.T @ A / W + 0.1 * np.eye(W); C = rng.standard_normal((W, W))
    for bs in (128, 256, 352):
        os.environ["KP_WIDE_BS"] = str(bs)
        ctx.fit_solve(G, C)
        t0 = time.perf_counter()
        for _ in range(3): ctx.fit_solve(G, C)
        dt = (time.perf_counter() - t0) / 3
        print("solve W %d, %d right-hand sides, block %d: device %.3f ms (wall with host copies %.1f ms)" % (W, W, bs, ctx.timer(1), dt * 1e3), flush=True)
    os.environ.pop("KP_WIDE_BS", None)
