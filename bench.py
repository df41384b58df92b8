#!/usr/bin/env python3
"""bench.py — headline benchmark of the Koopman hot path on MI355X.

Metric (BASELINE.json): EDMD snapshot-pairs/s (+ MPC steps/s reported beside it) for the
3-link-arm bilinear model.  Workload at N=1 = BASELINE configs[1]: bilinear Koopman fit,
poly degree 3 (N=84, W=336), 1e5 synthetic snapshot pairs, one MI355X.  A "step" is one
complete fit (fused lift+Gram kernel, partial reduction, Cholesky solve -> K) of the
resident snapshot matrix.  Inputs are uploaded to HBM before the timed region.

Multi-GPU (N>1): one process per GPU; every rank fits its own snapshot matrix of the same
shape (the units of the reference's sweeps are independent fits: lasso grid
Ksysid.m:1372-1387, evaluate_rand_models.m:45-144) => weak scaling, no data-path
collective; one RCCL all_gather of the resulting K matrices closes the timed region.

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_F64_MFMA_TFLOPS = 78.6   # MI355X FP64 matrix (vendor datasheet; the guide lists no f64 row)
PEAK_HBM_GBS = 8000.0


def synth_pairs(Ns, nz=6, m=3, seed=0):
    rng = np.random.default_rng(seed)
    alpha = rng.uniform(-1, 1, (Ns, nz)); u = rng.uniform(-1, 1, (Ns, m))
    Mx = rng.standard_normal((nz + m, nz)) * 0.3
    beta = np.clip(alpha + 0.05 * np.tanh(np.hstack([alpha, u]) @ Mx), -1, 1)
    return alpha, beta, u


def cpu_baseline(Ns_sample, degree):
    """The oracle's restatement of get_Koopman (lift of every row + `\\`) on the host cores,
    on a bounded sample of the same workload."""
    from oracle import koopman_oracle as ko
    alpha, beta, u = synth_pairs(Ns_sample, seed=123)
    pairs = {"alpha": alpha, "beta": beta, "u": u}
    dic = ko.build_dictionary("bilinear", 6, 3, ["poly"], [degree])
    t0 = time.perf_counter()
    reps = 0
    while True:
        ko.get_koopman(dic, pairs)
        reps += 1
        if time.perf_counter() - t0 > 12.0:   # bounded sample: ~12 s of host work
            break
    dt = (time.perf_counter() - t0) / reps
    try:
        import threadpoolctl
        cores = max([p.get("num_threads", 1) for p in threadpoolctl.threadpool_info()] + [1])
    except Exception:
        cores = os.cpu_count()
    return {"value": Ns_sample / dt, "unit": "snapshot-pairs/s", "cores": int(cores), "kind": "port",
            "sample": f"{reps} x get_Koopman (numpy lift + LAPACK lstsq) on {Ns_sample} pairs, W=336, {dt*1e3:.0f} ms each"}


def mpc_problem(kra, ctx, basis, snaps, horizon=10):
    """Bilinear Kmpc on the model of the synthetic fit (N=84), example_control.m:20-28 settings."""
    from koopman_realizations_amd.device import Mpc
    N, m = basis.N, 3
    K = kra.fit(ctx, basis, snaps)[0]
    A = np.asfortranarray(K[:N, :N].T); B = np.asfortranarray(K[N:, :N].T)      # Ksysid.m:1258-1259
    proj = np.zeros((2, N)); proj[0, 4] = proj[1, 5] = 1.0                      # C(end-1:end,:)
    u_fac = 2.8
    setup = dict(A=A, B=B, N=N, m=m, Np=horizon, proj=proj, q_run=10.0, q_term=100.0,
                 r=0.1 * np.array([3e-2, 2e-2, 1e-2]), lo=np.full(3, -7 * np.pi / 8 / u_fac), hi=np.full(3, 7 * np.pi / 8 / u_fac),
                 slope=1e-1 * u_fac)
    mpc = Mpc(ctx, "bilinear", A, B, horizon, proj, setup["q_run"], setup["q_term"], setup["r"], setup["lo"], setup["hi"], setup["slope"])
    return mpc, setup


def mpc_inputs(nb, seed=7):
    rng = np.random.default_rng(seed)
    zeta = rng.uniform(-0.6, 0.6, (nb, 6))
    u_prev = rng.uniform(-0.3, 0.3, (nb, 3))
    th = rng.uniform(0, 2 * np.pi, (nb, 1)) + 0.15 * np.arange(11)[None, :]
    ref = np.stack([zeta[:, [4]] + 0.2 * (np.cos(th) - np.cos(th[:, :1])), zeta[:, [5]] + 0.2 * (np.sin(th) - np.sin(th[:, :1]))], axis=2)
    return zeta, u_prev, ref.reshape(nb, -1)      # Yr = vec(ref') per problem


def bench_mpc(ctx, kra, basis, snaps, args):
    """MPC steps/s: (a) latency-bound stream of single steps on INDEPENDENT random states (lift + assembly + QP in
    one launch, host round trip per step, as Ksim.run_trial_mpc calls it), (b) batched independent problems,
    (c) a 300-step closed loop (the host plant update is inside the timed loop)."""
    from koopman_realizations_amd import _ffi as F
    mpc, setup = mpc_problem(kra, ctx, basis, snaps)
    zeta, u_prev, Yr = mpc_inputs(max(args.mpc_steps, args.mpc_batch))
    for i in range(5):
        mpc.step_zeta(basis, zeta[i], u_prev[i], Yr[i])
    t0 = time.perf_counter(); ok = 0; kern = []
    for i in range(args.mpc_steps):
        U, z, st = mpc.step_zeta(basis, zeta[i], u_prev[i], Yr[i])
        ok += st == 0; kern.append(ctx.timer(2))
    dt1 = time.perf_counter() - t0
    # (c) the closed loop of BASELINE configs[2]: 300 steps, the identified model as the plant (Kmpc.run_simulation,
    # Kmpc.m:403-512), circular end-effector reference; consecutive QPs share most of their active set, which the
    # single-problem path uses as a warm start
    A_, B_, N_ = setup["A"], setup["B"], setup["N"]
    th = 0.05 * np.arange(args.mpc_steps + 12)
    cref = np.stack([0.3 * np.cos(th), 0.3 * np.sin(th)], axis=1)
    zc = np.zeros(6); zc[4] = 0.3
    uc = np.zeros(3)
    t0 = time.perf_counter(); okc = 0; kernc = []
    for k in range(args.mpc_steps):
        U, z, st = mpc.step_zeta(basis, zc, uc, cref[k:k + 11].reshape(-1))
        okc += st == 0; kernc.append(ctx.timer(2))
        if st != 0:
            break
        z1 = A_ @ z + sum(B_[:, i * N_:(i + 1) * N_] @ z * U[0, i] for i in range(3))
        zc = z1[:6]; uc = U[0]
    dtc = time.perf_counter() - t0
    Z = basis.lift(F.LIFT_ECON, zeta[:args.mpc_batch])
    mpc.step_batch(Z, u_prev[:args.mpc_batch], Yr[:args.mpc_batch])
    dtb = 1e30
    for _ in range(3):       # best of 3: one call is a few ms, so a single wall-clock sample is noisy
        t0 = time.perf_counter()
        Ub, stb = mpc.step_batch(Z, u_prev[:args.mpc_batch], Yr[:args.mpc_batch])
        dtb = min(dtb, time.perf_counter() - t0)
    return {"single_steps_per_s": args.mpc_steps / dt1, "single_us_per_step": dt1 / args.mpc_steps * 1e6,
            "single_kernel_us": float(np.mean(kern)) * 1e3, "single_solved": int(ok), "single_steps": args.mpc_steps,
            "closed_loop_steps_per_s": okc / dtc, "closed_loop_us_per_step": dtc / max(okc, 1) * 1e6,
            "closed_loop_kernel_us": float(np.mean(kernc)) * 1e3, "closed_loop_solved": int(okc),
            "batch_problems_per_s": args.mpc_batch / dtb, "batch": args.mpc_batch, "batch_kernel_ms": ctx.timer(2),
            "batch_solved": int((stb == 0).sum()),
            "workload": "bilinear Kmpc, N=84 model of the synthetic fit, horizon 10, 30 variables x 126 rows (BASELINE configs[2] shape)",
            "reference_recorded": "MATLAB R2019a stored comp_time: median 8.7 ms/step (~104 steps/s), N=34, unknown PC",
            "_setup": (setup, basis.N, zeta[:20], u_prev[:20], Yr[:20])}


def bench_lasso(ctx, kra, basis, snaps, n_values=8):
    """BASELINE configs[3] on one GPU: a lasso grid on the bilinear fit through ONE kp_fit call (snapshots lifted once,
    least-squares solution and Lipschitz constant shared, Ksysid.train_models with a vector of lasso values).  The
    grid is chosen so that the L1 constraint is active for every value (t between 0.9 and 0.1 of ||K_LS||_1)."""
    G, C = kra.fit_gram(ctx, basis, snaps)
    l1 = float(np.abs(ctx.fit_solve(G, C)).sum())
    vals = list(np.geomspace(0.9, 0.1, n_values) * l1 / basis.N)          # t = lasso * N (Ksysid.m:996)
    kra.fit(ctx, basis, snaps, vals[:1])
    t0 = time.perf_counter()
    Ks = kra.fit(ctx, basis, snaps, vals)
    dt = time.perf_counter() - t0
    return {"values": n_values, "seconds": dt, "values_per_s": n_values / dt, "W": basis.W,
            "l1_fraction_reached": [float(np.abs(K).sum() / l1) for K in (Ks[0], Ks[-1])],
            "workload": "lasso grid on the bilinear poly-3 fit, 1e5 pairs, constraint active (BASELINE configs[3] shape, one GPU's share)"}


def bench_rand_sweep(ctx, kra, n_systems=256):
    """BASELINE configs[4] on one GPU: evaluate_rand_models.m (linear / bilinear / nonlinear fits of every degree +
    validation rollouts per random system) through the batched path; the three systems committed under tests/golden
    (taken from the reference's data set) are repeated to n_systems."""
    from koopman_realizations_amd import sweep
    g = np.load(os.path.join(ROOT, "tests", "golden", "rand_systems.npz"))
    def system(i):
        t, y, u = g[f"s{i}_train_t"], g[f"s{i}_train_y"], g[f"s{i}_train_u"]
        n = t.shape[0] // 1001
        train = [{"t": t[k * 1001:(k + 1) * 1001], "y": y[k * 1001:(k + 1) * 1001], "u": u[k * 1001:(k + 1) * 1001]} for k in range(n)]
        return {"train": train, "val": [{"t": g[f"s{i}_val_t"], "y": g[f"s{i}_val_y"], "u": g[f"s{i}_val_u"]}]}
    base = [system(i) for i in range(3)]
    systems = [base[i % 3] for i in range(n_systems)]
    sweep.rand_models_sweep_batched(systems[:3], ctx)
    t0 = time.perf_counter()
    tab = sweep.rand_models_sweep_batched(systems, ctx)
    dt = time.perf_counter() - t0
    return {"systems": n_systems, "seconds": dt, "systems_per_s": n_systems / dt, "fits_per_system": int(sum(len(v) for v in tab.values())),
            "workload": "evaluate_rand_models.m: 23 fits + validation rollouts per 1-D random system (BASELINE configs[4] shape, one GPU's share)"}


def cpu_baseline_mpc(pack):
    """The oracle's literal Kmpc step (Bhat from dense matrix powers, 4 rebuilds folded into one,
    exact active-set QP) on the host, on a bounded sample of the same problems."""
    from oracle import koopman_oracle as ko
    setup, N, zeta, u_prev, Yr = pack
    dic = ko.build_dictionary("bilinear", 6, 3, ["poly"], [3])
    s = ko.MpcSetup("bilinear", setup["A"], setup["B"], 3, setup["Np"], setup["proj"], setup["q_run"], setup["q_term"], setup["r"],
                    np.stack([setup["lo"], setup["hi"]], axis=1), setup["slope"], None, None, 6)
    t0 = time.perf_counter(); n = 0
    for i in range(len(zeta)):
        z = ko.econ_full(dic, zeta[i][None, :])[0]
        ko.mpc_step(s, z, u_prev[i], Yr[i].reshape(-1, 2))
        n += 1
        if time.perf_counter() - t0 > 10.0:
            break
    dt = (time.perf_counter() - t0) / n
    return {"value": 1.0 / dt, "unit": "MPC steps/s", "cores": 1, "kind": "port",
            "sample": f"{n} literal Kmpc steps (numpy), N={N}, {dt*1e3:.1f} ms each"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--snapshots", type=int, default=100000)
    ap.add_argument("--degree", type=int, default=3)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-mpc", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the lasso-grid and random-sweep sections")
    ap.add_argument("--mpc-steps", type=int, default=300)
    ap.add_argument("--mpc-batch", type=int, default=4096)
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    dist = None
    if world > 1:
        import torch
        import torch.distributed as dist
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    import koopman_realizations_amd as kra
    from koopman_realizations_amd import _ffi as F

    ctx = kra.Context(local_rank)
    Ns = args.snapshots
    alpha, beta, u = synth_pairs(Ns, seed=rank)
    basis = kra.Basis(ctx, "bilinear", 6, 3, [("poly", kra.poly_exponent_table(6, args.degree)[6:])])
    snaps = kra.Snapshots(ctx, alpha, beta, u)      # resident in HBM before timing
    W = basis.W

    def barrier():
        if dist is not None:
            import torch
            torch.cuda.synchronize()
            dist.barrier()

    # Steps are independent fits (the unit of the reference's sweeps), issued through the library's
    # asynchronous pipeline: the solve of fit i (second HIP stream) overlaps the fused Gram kernel of
    # fit i+1; everything is drained (kp_synchronize) inside the timed region.
    for _ in range(args.warmup):
        kra.fit(ctx, basis, snaps, fetch=False)
    ctx.synchronize()
    t_gram, t_red, t_solve = [], [], []
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        kra.fit(ctx, basis, snaps, fetch=False)     # enqueue: Gram on stream 1, solve on stream 2
    ctx.synchronize()                               # all K fits complete (HIP events of the last step are read after this)
    # timer 0 = mean HIP-event duration of the last min(steps, 64) Gram launches of this timed region (timer 7 = count)
    t_gram.append(ctx.timer(0)); t_red.append(ctx.timer(6)); t_solve.append(ctx.timer(1)); n_timed = int(ctx.timer(7))
    if dist is not None:
        import torch
        K = np.zeros((W, W), order="F")
        F.check(F.lib().kp_fit_get_K(ctx.handle, 0, W, F.dptr(K)), ctx.handle)
        Kt = torch.from_numpy(K).cuda()
        out = [torch.empty_like(Kt) for _ in range(world)]
        dist.all_gather(out, Kt)                    # the sweep's only collective: final gather
        torch.cuda.synchronize()
    barrier()
    dt = time.perf_counter() - t0
    if dist is not None:
        import torch
        tt = torch.tensor([dt], dtype=torch.float64, device="cuda")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())

    # one-fit latency (no overlap): synchronous path of the same entry point
    os.environ["KP_NO_ASYNC"] = "1"
    lat = []
    for _ in range(5):
        t1 = time.perf_counter(); kra.fit(ctx, basis, snaps, fetch=False); lat.append(time.perf_counter() - t1)
    del os.environ["KP_NO_ASYNC"]
    fit_latency_ms = float(np.median(lat)) * 1e3
    mpc_res = None
    if rank == 0 and not args.no_mpc and world == 1:     # secondary sections only in the single-GPU run
        mpc_res = bench_mpc(ctx, kra, basis, snaps, args)
    extras = None
    if rank == 0 and not args.no_extras and world == 1 and args.degree == 3:
        extras = {"lasso_grid": bench_lasso(ctx, kra, basis, snaps), "rand_sweep": bench_rand_sweep(ctx, kra)}

    if rank == 0:
        flops_pair = W * (W + 1) + 2.0 * W * W                 # F(336) = 339 024
        g_ms = float(np.mean(t_gram))
        achieved = flops_pair * Ns / (g_ms * 1e-3) / 1e12
        # HBM traffic of the dominant kernel: PMC counters are collected in separate rocprofv3 passes of this
        # same command (tools/prof_round.sh) and committed under profiles/; null when the workload differs
        traffic, prof_note = None, None
        pj = os.path.join(ROOT, "profiles", "r01_pmc_summary.json")
        if os.path.exists(pj) and Ns == 100000 and args.degree == 3:
            try:
                pr = json.load(open(pj))
                traffic = pr["gram_traffic_bytes_per_launch"]
                prof_note = {"executed_flop_per_launch": pr["gram_executed_flop"], "mfma_busy_cycles_per_instr": pr["gram_mfma_busy_cycles_per_instr"],
                             "source": "profiles/r01_pmc_summary.json"}
            except Exception:
                pass
        res = {
            "metric": "EDMD snapshot-pairs/sec (bilinear fit, 3-link arm, poly-3)",
            "value": world * Ns * args.steps / dt,
            "unit": "snapshot-pairs/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "config": {"workload": f"bilinear Koopman fit, poly degree {args.degree}, {Ns} synthetic snapshot pairs per GPU, "
                                   f"N={basis.N}, W={W} (BASELINE configs[1])",
                       "snapshots_per_gpu": Ns, "W": W, "parallelism": f"{world} independent fits + final all_gather"},
            "fit_latency_ms": fit_latency_ms,
            "kernel_ms": {"gram": g_ms, "gram_launches_averaged": n_timed, "gram_reduce": float(np.mean(t_red)), "solve": float(np.mean(t_solve))},
            "roofline": {"bound": "mfma", "kernel": "kp_gram3_kernel<6,3>", "achieved": achieved, "peak": PEAK_F64_MFMA_TFLOPS,
                         "unit": "TFLOP/s", "frac": achieved / PEAK_F64_MFMA_TFLOPS, "traffic": traffic,
                         "algorithmic_flop_per_launch": flops_pair * Ns, "algorithmic_bytes_per_launch": 120.0 * Ns,
                         "hbm_algorithmic_GBs": 120.0 * Ns / (g_ms * 1e-3) / 1e9,
                         "note": "achieved = dense algorithmic flop W(W+1)+2W^2 per pair / measured kernel time; the Kronecker-structured "
                                 "kernel executes ~63% of them (exact same G, C), so frac can exceed the executed-MFMA utilisation",
                         "pmc": prof_note},
        }
        if mpc_res is not None:
            res["mpc"] = mpc_res
        if extras is not None:
            res.update(extras)
        if not args.no_cpu_baseline and world == 1:
            res["cpu_baseline"] = cpu_baseline(min(Ns, 20000), args.degree)
            if mpc_res is not None:
                res["cpu_baseline"]["mpc"] = cpu_baseline_mpc(mpc_res.pop("_setup"))
        if mpc_res is not None:
            mpc_res.pop("_setup", None)
        print(json.dumps(res))
    if dist is not None:
        barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
