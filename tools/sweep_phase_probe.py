"""Where the batched sweep of 1024 (and of 128) systems spends its wall time: gather, upload, per model type passes."""
import os, sys, time, gc, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
chunks = bench.gen_rand_systems(list(range(8)))
gc.collect(); gc.freeze()
import koopman_realizations_amd as kra
from koopman_realizations_amd import sweep
from koopman_realizations_amd.device import Basis, Traj
from koopman_realizations_amd.ksysid import poly_exponent_table
ctx = kra.Context(0)
allsys = [s for c in sorted(chunks) for s in chunks[c]]
for nb in (1024, 128):
    mine = allsys[:nb]
    sweep.rand_models_sweep_batched(mine, ctx)
    for rep in range(2):
        T = {}
        t0 = time.perf_counter(); raw = sweep._stack_raw(mine, ctx); T["gather"] = time.perf_counter() - t0
        Y, U, k, Yv, Uv = raw
        t0 = time.perf_counter(); traj = Traj(ctx, Y, U, k, Yv, Uv); T["traj upload"] = time.perf_counter() - t0
        for mt in ("linear", "bilinear", "nonlinear"):
            nv = 1 + (1 if mt == "nonlinear" else 0); D = sweep.MAX_DEGREE[mt]
            t0 = time.perf_counter(); basis = Basis(ctx, mt, 1, 1, [("poly", poly_exponent_table(nv, D)[nv:])], None); T[mt + " basis"] = time.perf_counter() - t0
            t0 = time.perf_counter(); err, st = traj.sweep_eval_nested(basis, D, 4.0 if mt == "nonlinear" else np.inf); T[mt + " eval"] = time.perf_counter() - t0
            T[mt + " device ms (timer 0 gram)"] = ctx.timer(0) * 1e-3
            t0 = time.perf_counter(); basis.close(); T[mt + " basis close"] = time.perf_counter() - t0
        t0 = time.perf_counter(); traj.close(); T["traj close"] = time.perf_counter() - t0
        print(nb, {k_: round(v * 1e3, 2) for k_, v in T.items()}, flush=True)
