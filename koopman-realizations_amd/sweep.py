"""Embarrassingly parallel sweeps over independent fits, one process per GPU.

The reference runs these loops serially in one MATLAB interpreter:
  * lasso grid            train_models with a vector of lasso values (Ksysid.m:1372-1387)
  * random-system sweep   evaluate_rand_models.m:45-144 (per system: linear deg 1..13,
                          bilinear deg 1..6, nonlinear deg 1..4 with lasso 4)
Units (lasso values / systems) are dealt round-robin to ranks; there is NO collective on the
data path.  The only communication is the final gather of the per-unit results through a `comm`
object (comm.RcclComm: the kp_comm_* entry points of the library, RCCL over xGMI; the CPU tests pass a
gloo-backed stand-in with the same members).  No torch in this package.
"""
from __future__ import annotations

import numpy as np

from . import comm as _comm

try:                                     # host-side gather helper (plain C extension built by csrc/Makefile); numpy otherwise
    from . import _kp_gather
except ImportError:                      # pragma: no cover - the helper is optional
    _kp_gather = None
import os as _os
_GATHER_THREADS = max(1, min(12, len(_os.sched_getaffinity(0)) if hasattr(_os, "sched_getaffinity") else (_os.cpu_count() or 1)))


def shard_units(n_units: int, rank: int, world: int):
    """Unit ids owned by `rank`: round-robin, so ragged counts differ by at most one."""
    return list(range(rank, n_units, world))


def gather_results(local: dict, n_units: int, comm=None):
    """Final gather.  `local` maps unit id -> result (numpy array or picklable object).
    Returns the list of all results ordered by unit id on every rank."""
    if comm is None or comm.world == 1:
        merged = dict(local)
    else:
        merged = {}
        for p in _comm.all_gather_object(comm, local):
            merged.update(p)
    missing = [i for i in range(n_units) if i not in merged]
    if missing:
        raise RuntimeError(f"sweep: units {missing[:5]}... were not computed by any rank")
    return [merged[i] for i in range(n_units)]


def gather_matrices(local: dict, n_units: int, shape, comm=None):
    """Final gather of equally shaped f64 matrices with ONE all-gather (RCCL on GPUs).
    Ranks own round-robin shards, so every rank contributes ceil(n/world) slots (padded)."""
    if comm is None or comm.world == 1:
        return [local[i] for i in range(n_units)]
    world, rank = comm.world, comm.rank
    per = (n_units + world - 1) // world
    buf = np.zeros((per,) + tuple(shape))
    for slot, uid in enumerate(shard_units(n_units, rank, world)):
        buf[slot] = local[uid]
    out = _comm.all_gather_array(comm, buf)
    res = [None] * n_units
    for r in range(world):
        for slot, uid in enumerate(shard_units(n_units, r, world)):
            res[uid] = out[r, slot]
    return res


def lasso_sweep(fit_one, lassos, comm=None, shape=None, fit_many=None):
    """Config 4: K for every lasso value.  fit_one(lasso) -> K (W x W); fit_many(list of lassos) -> list of K lets a rank
    hand its whole shard to ONE kp_fit call (snapshots lifted once, values batched on the device)."""
    lassos = list(lassos)
    rank, world = (0, 1) if comm is None else (comm.rank, comm.world)
    mine = shard_units(len(lassos), rank, world)
    if fit_many is not None:
        local = dict(zip(mine, fit_many([lassos[i] for i in mine]))) if mine else {}
    else:
        local = {i: fit_one(lassos[i]) for i in mine}
    if shape is not None:
        return gather_matrices(local, len(lassos), shape, comm)
    return gather_results(local, len(lassos), comm)


def lasso_sweep_device(ctx, fit_device, lassos, W: int, comm=None, root=None):
    """Config 4 with the K stack kept in HBM until the one gather: `fit_device(list of lassos)` runs this rank's shard as ONE
    kp_fit call that leaves its results in the device result buffer (device.fit(..., fetch=False)); the stacks of all ranks
    are gathered device to device (comm.all_gather_fits -> kp_comm_allgather_fits) and land in a page-locked block.
    Returns the K of every value (views into that block, valid until the context's next gather), ordered like `lassos`.
    `root`: the stack goes to that rank only (kp_comm_gather_fits) - the reference's caller of train_models is ONE host
    (Ksysid.m:1370-1387) - and the other ranks return None."""
    lassos = list(lassos)
    rank, world = (0, 1) if comm is None else (comm.rank, comm.world)
    mine = shard_units(len(lassos), rank, world)
    per = (len(lassos) + world - 1) // world
    if mine:
        fit_device([lassos[i] for i in mine])
    out = _comm.all_gather_fits(comm, ctx, 0, per, W, have=len(mine), root=root)
    if out is None:
        return None
    return [out[uid % world, uid // world].T for uid in range(len(lassos))]


def shard_rows(n_rows: int, rank: int, world: int):
    """Contiguous snapshot range [lo, hi) of `rank` (sizes differ by at most one)."""
    base, rem = divmod(n_rows, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def fit_sharded(gram_fn, solve_fn, comm=None):
    """ONE large fit sharded over snapshots (pattern 2): every rank accumulates the Grams
    G = Px'Px, C = Px'Py of its snapshot shard (gram_fn() -> (G, C)), a single all-reduce(sum) of the
    stacked [G; C] (2 W^2 doubles, 1.8 MB at W = 336) is the only exchange step, and every rank solves
    G K = C (solve_fn(G, C) -> K), so all ranks hold the same K.  (Host-buffer form; device.fit_sharded keeps the
    Grams in HBM and all-reduces them there: kp_fit_sharded.)"""
    G, C = gram_fn()
    if comm is not None and comm.world > 1:
        t = comm.all_reduce_sum(np.stack([np.asarray(G), np.asarray(C)]))
        G, C = np.asfortranarray(t[0]), np.asfortranarray(t[1])
    return solve_fn(G, C)


# evaluate_rand_models.m:14-16
MAX_DEGREE = {"linear": 13, "bilinear": 6, "nonlinear": 4}


def eval_system(data4sysid, ctx=None, degrees=None, Ksysid=None):
    """evaluate_rand_models.m:47-143 for ONE system: normalised mean validation error of every
    model type / degree.  Returns dict model_type -> (errors[deg], dims[deg])."""
    if Ksysid is None:
        from .ksysid import Ksysid
    degrees = degrees or MAX_DEGREE
    out = {}
    for mt in ("linear", "bilinear", "nonlinear"):
        errs, dims = [], []
        for j in range(1, degrees[mt] + 1):
            ks = Ksysid(data4sysid, ctx=ctx, model_type=mt, obs_type=["poly"], obs_degree=[j], snapshots=np.inf,
                        lasso=[4.0] if mt == "nonlinear" else [np.inf],      # :56,:89,:122
                        delays=0, loaded=False, dim_red=False)
            ks.train_models()
            vd = ks.valdata[0]
            res = {"linear": ks.val_model, "bilinear": ks.val_BLmodel, "nonlinear": ks.val_NLmodel}[mt](ks.model, vd)
            mean_error = res["error"]["mean"]                                   # :70
            mean_error_zeros = np.abs(res["real"]["y"]).sum(axis=0) / res["real"]["y"].shape[0]   # :71
            errs.append(float(np.ravel(mean_error / mean_error_zeros)[0]))      # :72 (n = 1 systems)
            dims.append(ks.basis_dev.W if mt == "bilinear" else ks.basis_dev.nfull)   # :76,:109,:142
        out[mt] = (np.array(errs), np.array(dims))
    return out


def rand_models_sweep(systems, comm=None, ctx=None, degrees=None, eval_fn=None, batched=False):
    """Config 5: every rank evaluates its systems; final gather of the error tables.
    Returns dict model_type -> array (max_degree x n_systems), as err_*_models in
    evaluate_rand_models.m:38-43.  batched=True: the rank's whole shard goes through
    `rand_models_sweep_batched` (one launch per model type and degree) instead of one Ksysid per fit."""
    rank, world = (0, 1) if comm is None else (comm.rank, comm.world)
    mine = shard_units(len(systems), rank, world)
    if batched and eval_fn is None:
        tab = rand_models_sweep_batched([systems[i] for i in mine], ctx, degrees=degrees) if mine else {}
        local = {i: {mt: (tab[mt][:, k], None) for mt in tab} for k, i in enumerate(mine)}
        allres = gather_results(local, len(systems), comm)
        return {mt: np.stack([r[mt][0] for r in allres], axis=1) for mt in ("linear", "bilinear", "nonlinear")}
    eval_fn = eval_fn or (lambda d: eval_system(d, ctx=ctx, degrees=degrees))
    local = {i: eval_fn(systems[i]) for i in mine}
    allres = gather_results(local, len(systems), comm)
    return {mt: np.stack([r[mt][0] for r in allres], axis=1) for mt in ("linear", "bilinear", "nonlinear")}


def _stack_raw(systems, ctx=None, slot="", on_block=None):  # slot: suffix of the buffer names (a caller that keeps two gathers alive)
    """The systems' trials as stacked raw arrays (no arithmetic): Y (nb, k T, n), U (nb, k T, m), trial count k, and the
    validation trial Yv, Uv - or None unless every system has the same trial layout (equal counts and lengths, time
    restarting at every trial: the generated and shipped rand-systems sets).  With a device context the blocks are
    gathered into its page-locked host arrays (Context.host_array, reused from call to call: no first-touch page faults -
    they were two thirds of this function's time - and the upload that follows is a direct DMA); the returned arrays are
    then views that stay valid until the next call with the same context and `slot`.  `on_block(name, array)`, if given, is
    called for 'Y', 'U', 'Yv', 'Uv' as soon as each block is gathered - the batched sweep enqueues its upload there, so the
    DMA of one block runs beside the gather of the next and the seam test of the time vectors; when a later step fails
    (None is returned) the caller discards what it started."""
    try:
        tr = [d["train"] for d in systems]
        k = len(tr[0])
        if any(len(t) != k for t in tr):
            return None

        def stacked(key, src, name):                                       # ONE C-level gather per quantity
            a0 = np.asarray(src[0][0][key])
            Tn = a0.shape[0]
            w = a0.size // max(Tn, 1)
            shape = (len(src), len(src[0]) * Tn, w)
            out = ctx.host_array("sweep_" + name + slot, shape) if ctx is not None else np.empty(shape)
            if _kp_gather is not None:
                # buffer-protocol pointers + multi-threaded memcpy with the GIL released (csrc/kp_pygather.c): np.concatenate
                # spends ~1.7 us of set-up per 8 KB trial array, 60 ms for the 33 000 arrays of 1024 systems.  The helper
                # walks systems -> trials -> trial[key] itself (the flattening comprehension cost as much as the copy)
                try:
                    nbytes, same = _kp_gather.gather(src, out.ctypes.data, out.nbytes, _GATHER_THREADS, key)
                    if nbytes != out.nbytes or not same:
                        raise ValueError("ragged trials")
                    return out
                except TypeError:                                            # lists / other dtypes among the trials: numpy converts
                    pass
            arrs = [x[key] for t in src for x in t]
            try:
                uniform = {(a.shape, a.dtype) for a in arrs} == {(a0.shape, np.dtype(np.float64))}
            except AttributeError:                                           # lists / scalars among the trials
                uniform = False
            if not uniform:
                arrs = [np.asarray(a, dtype=np.float64) for a in arrs]
                if len({a.shape for a in arrs}) != 1:
                    raise ValueError("ragged trials")
            flat = out.reshape((-1,) + a0.shape[1:]) if a0.ndim > 1 else out.reshape(-1)
            np.concatenate(arrs, axis=0, out=flat)
            return out
        note = on_block if on_block is not None else (lambda name, arr: None)
        Y = stacked("y", tr, "Y"); note("Y", Y)
        U = stacked("u", tr, "U"); note("U", U)
        T = Y.shape[1] // k
        va = [[d["val"][0]] for d in systems]
        Yv = stacked("y", va, "Yv"); note("Yv", Yv)
        Uv = stacked("u", va, "Uv"); note("Uv", Uv)
        # Ksysid.m:948: seams between trials exactly at the trial joins, nowhere else.  The time vectors are only LOOKED at
        # (threads, no copy: csrc/kp_pygather.c); stacking them and comparing in numpy cost a third of this function
        seams_ok = None
        if _kp_gather is not None and hasattr(_kp_gather, "trials_increasing"):
            try:
                ok_, same_ = _kp_gather.trials_increasing(tr, k, _GATHER_THREADS, "t")
                seams_ok = bool(ok_) and bool(same_) and len(tr[0][0]["t"]) == T
            except TypeError:
                seams_ok = None
        if seams_ok is None:
            Tm = stacked("t", tr, "t")[:, :, 0]
            good = Tm[:, :-1] < Tm[:, 1:]
            seams = good[:, T - 1::T]
            seams_ok = not (seams.any() or int(np.count_nonzero(good)) != good.size - seams.size)
        if not seams_ok:
            return None
    except ValueError:
        return None
    return Y, U, k, Yv, Uv


def rand_models_sweep_batched(systems, ctx, degrees=None, nested=True):
    """Config 5 on ONE GPU with the data resident on the device: the trajectories of the shard are uploaded once
    (`kp_traj_upload`, scaling computed there), then for every (model type, degree) ONE `kp_sweep_eval` call does snapshot
    pairs + fit + model extraction + validation rollout + normalised error for all systems and returns only the error
    column.  nested=True: per model type ONE `kp_sweep_eval_nested` call serves all degrees from a single pass over the
    data (the degree-j dictionary is a column subset of the degree-D one).  Same table as `rand_models_sweep`
    (evaluate_rand_models.m:38-43).  Systems whose trials are not equally shaped go through the host-prepared path
    (`_sweep_batched_host`)."""
    from .device import Basis, Traj
    from .ksysid import poly_exponent_table
    degrees = degrees or MAX_DEGREE
    # every gathered block goes on its way to the device at once (kp_traj_create / kp_traj_put / kp_traj_finish): the DMA of
    # one block (3.5 ms per 1024 systems in all) runs beside the gather of the next and the seam test of the time vectors
    from .device import Traj
    traj = None
    try:
        d0 = systems[0]
        y0, u0, yv0 = np.asarray(d0["train"][0]["y"]), np.asarray(d0["train"][0]["u"]), np.asarray(d0["val"][0]["y"])
        dims = (len(systems), len(d0["train"]), y0.shape[0], y0.size // max(y0.shape[0], 1), u0.size // max(u0.shape[0], 1), yv0.shape[0])
        traj = Traj.begin(ctx, *dims)
    except Exception:                                   # unusual layouts: the blocks decide (Traj(...) below or the host path)
        traj = None

    def put(name, arr):
        nonlocal traj
        if traj is not None:
            try:
                traj.put(name, arr)
            except Exception:
                ctx.synchronize()
                traj.close()
                traj = None
    raw = _stack_raw(systems, ctx, on_block=put)
    if raw is None:
        if traj is not None:
            ctx.synchronize()                               # copies out of the gather buffers may still be in flight
            traj.close()
        return _sweep_batched_host(systems, ctx, degrees)
    if traj is not None:
        traj.finish()
    return rand_models_sweep_arrays(*raw, ctx=ctx, degrees=degrees, nested=nested, traj=traj)


def rand_models_sweep_arrays(Y, U, k, Yv, Uv, ctx, degrees=None, nested=True, traj=None):
    """The batched sweep on already stacked raw trajectories: Y (nb, k T, n), U (nb, k T, m) = the k training trials of
    every system back to back, Yv / Uv (nb, Tv, ·) the validation trial (what `_stack_raw` builds from the reference's
    data4sysid structs; a generator or loader that produces the blocks directly skips that gathering)."""
    from .device import Basis, Traj
    from .ksysid import poly_exponent_table
    degrees = degrees or MAX_DEGREE
    n, m = Y.shape[2], U.shape[2]
    if traj is None:                                   # (else: the caller has uploaded these very blocks already)
        traj = Traj(ctx, Y, U, k, Yv, Uv)
    out = {}
    try:
        for mt in ("linear", "bilinear", "nonlinear"):
            nv = n + (m if mt == "nonlinear" else 0)
            D = degrees[mt]
            basis = Basis(ctx, mt, n, m, [("poly", poly_exponent_table(nv, D)[nv:])], None)
            try:
                if nested and basis.W <= 16:          # all degrees from one pass over the data (sub-blocks of the degree-D Grams)
                    err, st = traj.sweep_eval_nested(basis, D, 4.0 if mt == "nonlinear" else np.inf)      # lasso 4: evaluate_rand_models.m:122
                    # a system whose batched fit failed (Gram not positive definite, lasso not converged) has no model: NaN,
                    # as the reference's own `\` would propagate, instead of whatever the rollout made of it
                    out[mt] = np.where(st != 0, np.nan, err[:, :, 0])
                    # kernel time of this model type's Gram pass (kp_traj_gram_kernel), its width and pair count: bench line
                    ctx.__dict__.setdefault("_sweep_gram", {})[mt] = (ctx.timer(0), basis.W, Y.shape[0] * (k * (Y.shape[1] // k - 1) - 1), ctx.timer(10))
                    continue
            finally:
                basis.close()
            rows = []
            for j in range(1, D + 1):
                basis = Basis(ctx, mt, n, m, [("poly", poly_exponent_table(nv, j)[nv:])], None)
                try:
                    err, st = traj.sweep_eval(basis, 4.0 if mt == "nonlinear" else np.inf)
                    rows.append(np.where(st != 0, np.nan, err[:, 0]))
                finally:
                    basis.close()
            out[mt] = np.stack(rows, axis=0)
    finally:
        traj.close()
    return out


def _sweep_batched_host(systems, ctx, degrees=None):
    """The batched sweep with scaling / snapshot pairs prepared on the host (systems of differing trial layouts): for
    every (model type, degree) one `kp_fit_batch` launch and one batched rollout launch."""
    from .device import Basis, Snapshots
    from .ksysid import Ksysid, poly_exponent_table
    from . import _ffi as F
    degrees = degrees or MAX_DEGREE
    nb = len(systems)
    # scaling, snapshot pairs and scaled validation data: once per system (independent of type and degree)
    pre = _prep_stacked(systems)
    if pre is None:                                                        # differently shaped systems: one Ksysid each
        prep = [Ksysid(d, ctx=ctx, model_type="linear", obs_type=["poly"], obs_degree=[1], snapshots=np.inf, lasso=[np.inf],
                       delays=0, loaded=False, dim_red=False, _host_only=True) for d in systems]
        n, m = prep[0].params["n"], prep[0].params["m"]
        Ns = prep[0].snapshotPairs["alpha"].shape[0]
        if any(k.snapshotPairs["alpha"].shape[0] != Ns or k.params["n"] != n or k.params["m"] != m for k in prep):
            raise ValueError("rand_models_sweep_batched: systems must share dimensions and snapshot counts")
        alpha = np.vstack([k.snapshotPairs["alpha"] for k in prep]); beta = np.vstack([k.snapshotPairs["beta"] for k in prep])
        uu = np.vstack([k.snapshotPairs["u"] for k in prep])
        val = [k._val_common(k.valdata[0]) for k in prep]                  # (t, yreal, ureal, zetareal)
        T = min(v[1].shape[0] for v in val)
        yreal = np.stack([v[1][:T] for v in val]); ureal = np.stack([v[2][:T] for v in val]); zeta0 = np.stack([v[3][0] for v in val])
    else:
        n, m, alpha, beta, uu, yreal, ureal, zeta0 = pre
        T = yreal.shape[1]
    snaps = Snapshots(ctx, alpha, beta, uu)
    mean_zero = np.abs(yreal).sum(axis=1) / T                              # evaluate_rand_models.m:71
    out = {}
    try:
        for mt in ("linear", "bilinear", "nonlinear"):
            rows = []
            for j in range(1, degrees[mt] + 1):
                nv = n + (m if mt == "nonlinear" else 0)
                basis = Basis(ctx, mt, n, m, [("poly", poly_exponent_table(nv, j)[nv:])], None)
                try:
                    K, G, Cm, st = ctx.fit_batch(basis, snaps, nb)
                    N, W = basis.N, basis.W
                    if mt == "nonlinear":                                   # lasso = 4 (:122): active only if ||K_ls||_1 > 4 N
                        K = np.array(K)
                        for s_ in np.nonzero(np.abs(K).sum(axis=(1, 2)) > 4.0 * N)[0]:
                            K[s_] = ctx.fit_lasso(G[s_], Cm[s_], 4.0 * N)[0]
                        Kf = np.transpose(K[:, :, :n], (0, 2, 1))           # Ksysid.m:1325
                        Z = ctx.rollout_nl_batch(basis, Kf, zeta0, ureal)
                        Y = Z[:, :, :n]
                    else:
                        UT = np.transpose(K, (0, 2, 1))
                        A, B = UT[:, :N, :N], UT[:, :N, N:]                 # Ksysid.m:1199-1200 / 1250-1251
                        if mt == "linear":                                  # M-projection of get_model (:1206-1225) from the Grams, on the device
                            bad = ~(np.isfinite(K).all(axis=(1, 2)))
                            A, B, _ = ctx.model_project_batch(np.nan_to_num(K), np.nan_to_num(G), np.nan_to_num(Cm), N, m)
                            A = np.array(A); B = np.array(B)
                            A[bad] = np.nan; B[bad] = np.nan
                        z0 = basis.lift(F.LIFT_ECON, zeta0)
                        Y = np.array(ctx.rollout(mt, np.nan_to_num(A), np.nan_to_num(B), z0, ureal, n))
                        Y[~np.isfinite(A).all(axis=(1, 2))] = np.nan
                    Y[:, 0] = yreal[:, 0]                                   # Ksysid.m:1654
                    mean_err = np.abs(Y - yreal).mean(axis=1)               # get_error (:1882-1898): mean abs error per output
                    with np.errstate(divide="ignore", invalid="ignore"):
                        rows.append((mean_err / mean_zero)[:, 0])
                finally:
                    basis.close()
            out[mt] = np.stack(rows, axis=0)
    finally:
        snaps.close()
    return out


def _prep_stacked(systems):
    """The constructor steps of Ksysid that the sweep needs - merge_trials (Ksysid.m:380-401), get_scale (:180-229),
    scale_data of the validation trial (:308-343), get_snapshotPairs (:941-978, delays = 0, all pairs) - for all systems
    at once with stacked numpy arrays (the per-system objects cost 0.2 ms each, a third of the whole sweep).  Returns
    None unless every system has the same trial layout.  The pairs are taken in time order instead of a random
    permutation: the least-squares fit does not depend on the order (up to rounding)."""
    try:
        tr = [d["train"] for d in systems]
        k = len(tr[0])
        if any(len(t) != k for t in tr):
            return None
        def stacked(key, src):                                             # nb x (k T) x width in one conversion
            a = np.asarray([[x[key] for x in t] for t in src], dtype=np.float64)   # ValueError when ragged
            if a.ndim == 3:                                                # vectors (t, or 1-D y / u): nb x k x T
                a = a[..., None]
            return a.reshape(a.shape[0], a.shape[1] * a.shape[2], -1)
        Y, U = stacked("y", tr), stacked("u", tr)
        Tm = stacked("t", tr)[:, :, 0]
        va = [[d["val"][0]] for d in systems]
        vy, vu = stacked("y", va), stacked("u", va)
    except ValueError:                                                     # ragged shapes
        return None
    good = Tm[:, :-1] < Tm[:, 1:]                                          # :948 seams between trials
    if not (good == good[0]).all():
        return None
    nb, _, n = Y.shape
    m = U.shape[2]
    def scale(V):                                                          # :187-210
        mn, mx = V.min(axis=1, keepdims=True), V.max(axis=1, keepdims=True)
        off, fac = (mx + mn) / 2.0, (mx - mn) / 2.0
        return off, np.where(fac == 0, 1.0, fac)
    oy, fy = scale(Y); ou, fu = scale(U)
    Ys, Us = (Y - oy) / fy, (U - ou) / fu
    idx = np.nonzero(good[0])[0][:-1]                                      # num_max = #good - 1 (:960): the last good pair is dropped
    alpha = Ys[:, idx].reshape(nb * len(idx), n); beta = Ys[:, idx + 1].reshape(nb * len(idx), n); uu = Us[:, idx].reshape(nb * len(idx), m)
    yreal, ureal = (vy - oy) / fy, (vu - ou) / fu
    return n, m, alpha, beta, uu, yreal, ureal, yreal[:, 0].copy()


def _keep_systems(err):
    """evaluate_rand_models.m:155-157: a system (column) is kept only if ALL its degrees have an
    error below 10 (NaN fails the comparison, so NaN systems drop out as well)."""
    err = np.asarray(err, dtype=np.float64)
    with np.errstate(invalid="ignore"):
        return err[:, np.all(err < 10, axis=0)]


def sweep_statistics(err):
    """evaluate_rand_models.m:149-171: drop systems with NaN / >= 10 errors, then mean and sample
    standard deviation over the remaining systems for each degree."""
    kept = _keep_systems(err)
    if kept.shape[1] == 0:
        return np.full(kept.shape[0], np.nan), np.full(kept.shape[0], np.nan)
    mean = kept.mean(axis=1)
    std = kept.std(axis=1, ddof=1) if kept.shape[1] > 1 else np.full(kept.shape[0], np.nan)
    return mean, std


def sweep_percentiles(err, q=(0, 25, 50, 75, 100)):
    """evaluate_rand_models.m:209-211: prctile(err, [0 25 50 75 100], 2) of the kept systems.
    MATLAB's prctile places sample i of n at 100 (i - 0.5)/n and interpolates linearly (numpy's
    'hazen' rule)."""
    kept = _keep_systems(err)
    if kept.shape[1] == 0:
        return np.full((kept.shape[0], len(q)), np.nan)
    return np.percentile(kept, q, axis=1, method="hazen").T
