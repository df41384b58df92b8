"""Thin object wrappers over the C ABI handles (kp_ctx, kp_basis, kp_snapshots, kp_mpc).

Host-side plumbing only: argument marshalling (column-major f64) and handle lifetime.
All arithmetic of the hot path happens inside libkoopman_hip.so on the GPU.
"""
from __future__ import annotations

import ctypes as C
import weakref

import numpy as np

from . import _ffi as F


class Context:
    """kp_ctx: one per process/GPU (one process per GPU in multi-GPU runs)."""

    def __init__(self, device_id: int = 0):
        self._h = F.vp()
        F.check(F.lib().kp_create(int(device_id), C.byref(self._h)))
        self.device_id = device_id
        self._children = weakref.WeakSet()       # Basis / Snapshots / Mpc / Traj handles: they hold pointers into this context

    @property
    def handle(self):
        return self._h

    def info(self):
        name = C.create_string_buffer(256)
        ncu = C.c_int()
        hbm = C.c_int64()
        F.check(F.lib().kp_device_info(self._h, name, 256, C.byref(ncu), C.byref(hbm)), self._h)
        return {"name": name.value.decode(), "num_cu": ncu.value, "hbm_bytes": hbm.value}

    def timer(self, which: int) -> float:
        ms = C.c_double()
        F.check(F.lib().kp_timer_get(self._h, which, C.byref(ms)), self._h)
        return ms.value

    def stream(self):
        return F.lib().kp_stream(self._h)

    def synchronize(self):
        """Waits for asynchronous fits (fit(..., fetch=False)); raises if one of them failed."""
        F.check(F.lib().kp_synchronize(self._h), self._h)

    def last_rank(self) -> int:
        """Rank found by the most recent solve (kp_fit_last_rank): W unless the dictionary was rank deficient."""
        r = C.c_int()
        F.check(F.lib().kp_fit_last_rank(self._h, C.byref(r)), self._h)
        return r.value

    def last_pivot_ratio(self) -> float:
        """min_i L_ii^2 / G_ii of the most recent synchronous least-squares solve (~1 / cond(G))."""
        r = C.c_double()
        F.check(F.lib().kp_fit_last_pivot_ratio(self._h, C.byref(r)), self._h)
        return r.value

    def fit_async_slots(self, n_slots: int):
        """Size of the result ring of asynchronous fits (fit(..., fetch=False)): the last n_slots fits of a batch stay
        retrievable with fit_result(q)."""
        F.check(F.lib().kp_fit_async_slots(self._h, int(n_slots)), self._h)

    def fit_result(self, index: int, W: int):
        """K of fit number `index` of the last batch of asynchronous fits (kp_fit_get_K); synchronises."""
        K = np.zeros((W, W), order="F")
        F.check(F.lib().kp_fit_get_K(self._h, int(index), int(W), F.dptr(K)), self._h)
        return K

    def fit_results(self, first: int, count: int, W: int):
        """K of fits first .. first + count - 1 of the last asynchronous batch as one (count, W, W) stack of column-major
        blocks in a page-locked host array of the context ("Kstack": direct DMA, 30 us per 0.9 MB K against 70-260 us
        into pageable memory; valid until the next fit_results call).  K[i].T is fit first + i in numpy's row-major view."""
        out = self.host_array("Kstack", (int(count), W, W))
        for i in range(int(count)):
            F.check(F.lib().kp_fit_get_K(self._h, int(first) + i, int(W), out[i].ctypes.data_as(F.c_dp)), self._h)
        return out

    def host_array(self, name: str, shape):
        """A float64 C-ordered array in page-locked host memory owned by this context (kp_host_alloc), kept under `name`
        and reused by later calls that fit into it: gathers written there cause no page faults and upload by direct DMA.
        The contents belong to the caller until the next host_array(name, ...) call."""
        n = int(np.prod(shape))
        pool = self.__dict__.setdefault("_host_pool", {})
        ent = pool.get(name)
        if ent is None or ent[1] < n:
            if ent is not None:
                F.check(F.lib().kp_host_free(self._h, ent[0]), self._h)
            p = F.vp()
            cap = n + n // 8 + 512
            F.check(F.lib().kp_host_alloc(self._h, cap * 8, C.byref(p)), self._h)
            ent = (p, cap, np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_double)), shape=(cap,)))
            pool[name] = ent
        return ent[2][:n].reshape(shape)

    def close(self):
        """Destroys the child handles first (their destroy calls dereference the context), then the context."""
        self.__dict__.pop("_host_pool", None)        # the blocks themselves are freed by kp_destroy
        if self._h:
            for ch in list(getattr(self, "_children", ())):
                try:
                    ch.close()
                except Exception:
                    pass
            F.lib().kp_destroy(self._h)
            self._h = F.vp()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- stateless entry points ----------------------------------------------------
    def fit_solve(self, G, Cm):
        """K = G \\ C by Cholesky on the device (normal equations of Ksysid.m:1069)."""
        G = F.fcol(G); Cm = F.fcol(Cm)
        W, nc = G.shape[0], Cm.shape[1]
        K = np.zeros((W, nc), order="F")
        F.check(F.lib().kp_fit_solve(self._h, F.dptr(G), F.dptr(Cm), W, nc, F.dptr(K)), self._h)
        return K

    def fit_lasso(self, G, Cm, t, max_iter=20000, tol=1e-10):
        G = F.fcol(G); Cm = F.fcol(Cm)
        W, nc = G.shape[0], Cm.shape[1]
        K = np.zeros((W, nc), order="F")
        it = C.c_int()
        F.check(F.lib().kp_fit_lasso(self._h, F.dptr(G), F.dptr(Cm), W, nc, float(t), int(max_iter), float(tol),
                                     F.dptr(K), C.byref(it)), self._h)
        return K, it.value

    def fit_lasso_batch(self, G, Cm, t, max_iter=20000, tol=1e-10):
        """kp_fit_lasso_batch: all L1 budgets t[v] on the same Grams at once.  Returns ([K_v], iters)."""
        G = F.fcol(G); Cm = F.fcol(Cm)
        W, nc = G.shape[0], Cm.shape[1]
        tv = np.ascontiguousarray(np.atleast_1d(np.asarray(t, dtype=np.float64)))
        K = np.zeros((len(tv), nc, W))                      # each W x nc block column-major
        it = np.zeros(len(tv), dtype=np.int32)
        F.check(F.lib().kp_fit_lasso_batch(self._h, F.dptr(G), F.dptr(Cm), W, nc, F.dptr(tv), len(tv), int(max_iter), float(tol),
                                           F.dptr(K), it.ctypes.data_as(F.c_ip)), self._h)
        return [np.asfortranarray(K[v].T) for v in range(len(tv))], it

    def fit_batch(self, basis, snaps, nb):
        """Least-squares fits of nb systems that share one dictionary (W <= 16): `snaps` holds the merged snapshot
        pairs, nb x Ns_each rows.  Returns K, G, C as (nb, W, W) arrays and the status vector."""
        W = basis.W
        Ns_each = snaps.Ns // nb
        K = np.zeros((nb, W, W)); G = np.zeros((nb, W, W)); Cm = np.zeros((nb, W, W))
        st = np.zeros(nb, dtype=np.int32)
        F.check(F.lib().kp_fit_batch(self._h, basis.handle, snaps.handle, nb, Ns_each, F.dptr(K), F.dptr(G), F.dptr(Cm),
                                     st.ctypes.data_as(C.POINTER(C.c_int))), self._h)
        # the library writes column-major W x W blocks: transpose the last two axes of the C-ordered buffers
        return K.transpose(0, 2, 1), G.transpose(0, 2, 1), Cm.transpose(0, 2, 1), st

    def model_project_batch(self, K, G, Cm, N, m):
        """get_model's M-projection (Ksysid.m:1206-1225) for nb small linear models on the device: K, G, Cm (nb, W, W)
        as fit_batch returns them -> A (nb, N, N), B (nb, N, m), status (nb,).  Singular systems give NaN."""
        nb = K.shape[0]
        to_cm = lambda X: np.ascontiguousarray(np.transpose(np.asarray(X, dtype=np.float64), (0, 2, 1)))     # per system column-major
        Kc, Gc, Cc = to_cm(K), to_cm(G), to_cm(Cm)
        A = np.zeros((nb, N, N)); B = np.zeros((nb, max(m, 1), N)); st = np.zeros(nb, dtype=np.int32)
        F.check(F.lib().kp_model_project_batch(self._h, F.dptr(Kc), F.dptr(Gc), F.dptr(Cc), nb, N, m, F.dptr(A), F.dptr(B), None,
                                               st.ctypes.data_as(C.POINTER(C.c_int))), self._h)
        return A.transpose(0, 2, 1), B[:, :m].transpose(0, 2, 1), st

    def sym_eig(self, S):
        """kp_sym_eig: eigenvalues (descending) and eigenvectors (columns) of a symmetric matrix, on the device."""
        S = F.fcol(np.array(S, dtype=np.float64))
        n = S.shape[0]
        V = np.zeros((n, n), order="F"); lam = np.zeros(n); sw = C.c_int()
        F.check(F.lib().kp_sym_eig(self._h, F.dptr(S), n, F.dptr(V), F.dptr(lam), C.byref(sw)), self._h)
        order = np.argsort(-lam, kind="stable")
        return lam[order], np.asfortranarray(V[:, order]), sw.value

    def rollout_nl_batch(self, basis, Kf, zeta0, U):
        """Batched nonlinear rollouts: Kf (nb, nzeta, N), zeta0 (nb, nzeta), U (nb, T, m) -> Z (nb, T, nzeta)."""
        nb, nz, N = Kf.shape
        T = U.shape[1]
        Kf_ = np.ascontiguousarray(np.transpose(Kf, (0, 2, 1)))          # per system column-major nzeta x N
        z0 = np.ascontiguousarray(zeta0, dtype=np.float64)
        U_ = np.ascontiguousarray(np.transpose(U, (0, 2, 1)))            # per system column-major T x m
        Z = np.zeros((nb, nz, T))
        F.check(F.lib().kp_rollout_nl(self._h, basis.handle, nb, F.dptr(Kf_), F.dptr(z0), F.dptr(U_), T, F.dptr(Z)), self._h)
        return np.transpose(Z, (0, 2, 1))

    def model_project(self, K, G, Cm, N, m):
        K = F.fcol(K); G = F.fcol(G); Cm = F.fcol(Cm)
        A = np.zeros((N, N), order="F"); B = np.zeros((N, m), order="F"); M = np.zeros((N, N), order="F")
        F.check(F.lib().kp_model_project(self._h, F.dptr(K), F.dptr(G), F.dptr(Cm), N, m, F.dptr(A), F.dptr(B), F.dptr(M)),
                self._h)
        return A, B, M

    def rollout(self, model_type, A, B, z0, U, n_out):
        """Batched rollouts: A (batch,N,N), B (batch,N,mb), z0 (batch,N), U (batch,T,m) -> Y (batch,T,n_out)."""
        A = np.asarray(A, dtype=np.float64); B = np.asarray(B, dtype=np.float64)
        z0 = np.asarray(z0, dtype=np.float64); U = np.asarray(U, dtype=np.float64)
        single = A.ndim == 2
        if single:
            A, B, z0, U = A[None], B[None], z0[None], U[None]
        batch, N = A.shape[0], A.shape[1]
        T, m = U.shape[1], U.shape[2]
        Ac = np.ascontiguousarray(np.transpose(A, (0, 2, 1)))  # each matrix column-major
        Bc = np.ascontiguousarray(np.transpose(B, (0, 2, 1)))
        Uc = np.ascontiguousarray(np.transpose(U, (0, 2, 1)))
        Y = np.zeros((batch, n_out, T))
        F.check(F.lib().kp_rollout(self._h, F.MODEL[model_type], batch, F.dptr(Ac), F.dptr(Bc), N, m,
                                   F.dptr(np.ascontiguousarray(z0)), F.dptr(Uc), T, n_out, F.dptr(Y)), self._h)
        Y = np.transpose(Y, (0, 2, 1))
        return Y[0] if single else Y

    def rollout_nl(self, basis, Kf, zeta0, U):
        """Nonlinear rollout zeta+ = Kf * econ_full([zeta;u]) on the device: Kf (nzeta,N), zeta0 (nzeta,), U (T,m)
        -> Z (T,nzeta)."""
        Kf = F.fcol(Kf); z0 = np.ascontiguousarray(zeta0, dtype=np.float64); U = F.fcol(np.atleast_2d(U))
        T = U.shape[0]
        Z = np.zeros((T, Kf.shape[0]), order="F")
        F.check(F.lib().kp_rollout_nl(self._h, basis.handle, 1, F.dptr(Kf), F.dptr(z0), F.dptr(U), T, F.dptr(Z)), self._h)
        return Z

    def qp_solve(self, H, f, A, b):
        """quadprog_gurobi(H,f,A,b) shim: NaN vector on failure (quadprog_gurobi.m:22-23)."""
        H = F.fcol(H); A = F.fcol(A)
        f = np.ascontiguousarray(f, dtype=np.float64); b = np.ascontiguousarray(b, dtype=np.float64)
        n, mr = H.shape[0], A.shape[0]
        x = np.zeros(n)
        st = C.c_int()
        F.check(F.lib().kp_qp_solve(self._h, F.dptr(H), F.dptr(f), F.dptr(A), F.dptr(b), n, mr, F.dptr(x), C.byref(st)),
                self._h)
        return x, st.value


def basis_desc(model_type, nzeta, m, blocks, pcs=None):
    """kp_basis_desc of a host-side dictionary description (blocks as Basis takes them).  Returns (desc, keep): `keep`
    holds the arrays the descriptor points into and must outlive every call that is handed the descriptor."""
    nvars = nzeta + (m if model_type == "nonlinear" else 0)
    btype, bcount, exps, centres = [], [], [], []
    for kind, arg in blocks:
        btype.append(F.BLOCK[kind])
        if kind in ("poly", "hermite"):
            e = np.ascontiguousarray(arg, dtype=np.uint8).reshape(-1, nvars)
            bcount.append(e.shape[0]); exps.append(e)
        elif kind == "fourier_sparser":
            e = np.ascontiguousarray(arg, dtype=np.uint8).reshape(-1, 2 * nvars)
            bcount.append(e.shape[0]); exps.append(e.reshape(-1, nvars))      # two table rows per function
        elif kind == "fourier":
            bcount.append(int(arg))
        else:
            c = np.asarray(arg, dtype=np.float64).reshape(nvars, -1)
            bcount.append(c.shape[1]); centres.append(np.ascontiguousarray(c.T))  # centre-major
    bt = np.array(btype, dtype=np.int32); bc = np.array(bcount, dtype=np.int32)
    ex = np.ascontiguousarray(np.vstack(exps)) if exps else np.zeros((0, nvars), np.uint8)
    ce = np.ascontiguousarray(np.vstack(centres)) if centres else np.zeros((0, nvars))
    pc = None if pcs is None else F.fcol(pcs)
    d = F.KpBasisDesc()
    d.model_type, d.nzeta, d.m, d.n_blocks = F.MODEL[model_type], nzeta, m, len(blocks)
    d.block_type = bt.ctypes.data_as(C.POINTER(C.c_int32))
    d.block_count = bc.ctypes.data_as(C.POINTER(C.c_int32))
    d.poly_exps = ex.ctypes.data_as(C.POINTER(C.c_uint8))
    d.gauss_centres = F.dptr(ce)
    d.k_pcs = 0 if pcs is None else pc.shape[1]
    d.pcs = F.dptr(pc)
    return d, (bt, bc, ex, ce, pc)


class Basis:
    """kp_basis built from a host-side dictionary description (see basis.py)."""

    def __init__(self, ctx: Context, model_type, nzeta, m, blocks, pcs=None):
        """blocks: list of ('poly', exps[rows,nvars] uint8) | ('fourier', deg) | ('gaussian', centres[nvars,k])
        | ('hermite', orders[rows,nvars] uint8) | ('fourier_sparser', multipliers[rows,2*nvars] uint8)."""
        self.ctx = ctx
        ctx._children.add(self)
        d, self._keep = basis_desc(model_type, nzeta, m, blocks, pcs)
        self._h = F.vp()
        F.check(F.lib().kp_basis_create(ctx.handle, C.byref(d), C.byref(self._h)), ctx.handle)
        nv, nf, N, W = C.c_int(), C.c_int(), C.c_int(), C.c_int()
        F.check(F.lib().kp_basis_dims(self._h, C.byref(nv), C.byref(nf), C.byref(N), C.byref(W)))
        self.nvars, self.nfull, self.N, self.W = nv.value, nf.value, N.value, W.value
        self.model_type, self.nzeta, self.m = model_type, nzeta, m

    @property
    def handle(self):
        return self._h

    def lift(self, what, zeta, u=None):
        zeta = F.fcol(np.atleast_2d(zeta))
        rows = zeta.shape[0]
        uu = None if u is None else F.fcol(np.atleast_2d(u))
        width = {F.LIFT_FULL: self.nfull, F.LIFT_ECON: self.N, F.LIFT_ROW: self.W}[what]
        out = np.zeros((rows, width), order="F")
        F.check(F.lib().kp_lift(self.ctx.handle, self._h, what, F.dptr(zeta), F.dptr(uu), rows, F.dptr(out)),
                self.ctx.handle)
        return out

    def close(self):
        if self._h:
            F.lib().kp_basis_destroy(self._h)
            self._h = F.vp()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Snapshots:
    """kp_snapshots: snapshot pairs resident in HBM."""

    def __init__(self, ctx: Context, alpha, beta, u):
        self.ctx = ctx
        ctx._children.add(self)
        a = F.fcol(alpha); b = F.fcol(beta); uu = F.fcol(u)
        self.Ns, self.nzeta, self.m = a.shape[0], a.shape[1], uu.shape[1]
        self._h = F.vp()
        F.check(F.lib().kp_snapshots_upload(ctx.handle, F.dptr(a), F.dptr(b), F.dptr(uu), self.Ns, self.nzeta, self.m,
                                            C.byref(self._h)), ctx.handle)

    def update(self, alpha, beta, u):
        """kp_snapshots_update: new pairs into the same device arrays.  Returns once the host arrays are staged; the
        transfer overlaps whatever the device is doing with OTHER snapshot objects (keep two and alternate)."""
        a = F.fcol(alpha); b = F.fcol(beta); uu = F.fcol(u)
        if a.shape[1] != self.nzeta or uu.shape[1] != self.m or b.shape != a.shape or uu.shape[0] != a.shape[0]:
            raise ValueError("Snapshots.update: column counts must match the object")
        F.check(F.lib().kp_snapshots_update(self.ctx.handle, self._h, F.dptr(a), F.dptr(b), F.dptr(uu), a.shape[0]), self.ctx.handle)
        self.Ns = a.shape[0]
        return self

    @property
    def handle(self):
        return self._h

    def close(self):
        if self._h:
            F.lib().kp_snapshots_destroy(self._h)
            self._h = F.vp()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Traj:
    """kp_traj: training / validation trajectories of nb systems with one layout, resident on the device (raw values;
    the per-system scaling of get_scale is computed there).  Y (nb, rows, n), U (nb, rows, m) with rows = ntrials * T
    (merged trials), Yv (nb, Tv, n), Uv (nb, Tv, m)."""

    def __init__(self, ctx: Context, Y, U, ntrials, Yv, Uv):
        self.ctx = ctx
        ctx._children.add(self)
        Y = np.asarray(Y, dtype=np.float64); U = np.asarray(U, dtype=np.float64)
        Yv = np.asarray(Yv, dtype=np.float64); Uv = np.asarray(Uv, dtype=np.float64)
        self.nb, rows, self.n = Y.shape
        self.m = U.shape[2]
        self.ntrials, self.T, self.Tv = int(ntrials), rows // int(ntrials), Yv.shape[1]
        if self.T * self.ntrials != rows:
            raise ValueError("Traj: rows must be ntrials * T")
        tr = Traj._blocks
        self._h = F.vp()
        F.check(F.lib().kp_traj_upload(ctx.handle, F.dptr(tr(Y)), F.dptr(tr(U)), self.nb, self.ntrials, self.T, self.n, self.m,
                                       F.dptr(tr(Yv)), F.dptr(tr(Uv)), self.Tv, C.byref(self._h)), ctx.handle)

    @staticmethod
    def _blocks(a):
        """(nb, rows, width) -> each system column-major rows x width (no copy for width 1)."""
        return np.ascontiguousarray(np.transpose(a, (0, 2, 1)))

    @classmethod
    def begin(cls, ctx: Context, nb, ntrials, T, n, m, Tv):
        """kp_traj_create: the object in three steps (begin, put x 4, finish), so that a caller that assembles the blocks one
        after the other has each on its way to the device while it prepares the next."""
        self = cls.__new__(cls)
        self.ctx = ctx
        ctx._children.add(self)
        self.nb, self.ntrials, self.T, self.n, self.m, self.Tv = int(nb), int(ntrials), int(T), int(n), int(m), int(Tv)
        self._h = F.vp()
        self._keep = []
        F.check(F.lib().kp_traj_create(ctx.handle, self.nb, self.ntrials, self.T, self.n, self.m, self.Tv, C.byref(self._h)), ctx.handle)
        return self

    def put(self, which, a):
        """kp_traj_put: block 'Y' | 'U' | 'Yv' | 'Uv' as (nb, rows, width); asynchronous when `a` lies in page-locked memory
        (Context.host_array), which must stay untouched until finish()."""
        w = {"Y": 0, "U": 1, "Yv": 2, "Uv": 3}[which]
        a = np.asarray(a, dtype=np.float64)
        want = (self.nb, (self.ntrials * self.T) if w < 2 else self.Tv, self.n if w in (0, 2) else self.m)
        if a.shape != want:
            raise ValueError(f"Traj.put: block {which} has shape {a.shape}, expected {want}")
        blk = Traj._blocks(a)
        self._keep.append(blk)                                                # alive until the copy has been waited for
        F.check(F.lib().kp_traj_put(self._h, w, F.dptr(blk)), self.ctx.handle)

    def finish(self):
        """kp_traj_finish: scaling on the device, stream synchronised - the object is ready."""
        F.check(F.lib().kp_traj_finish(self._h), self.ctx.handle)
        self._keep = []
        return self

    @property
    def handle(self):
        return self._h

    def scale(self):
        """Per system [y offset (n) | y factor (n) | u offset (m) | u factor (m)] as computed on the device."""
        sc = np.zeros((self.nb, 2 * (self.n + self.m)))
        F.check(F.lib().kp_traj_scale(self._h, F.dptr(sc)), self.ctx.handle)
        return sc

    def sweep_eval(self, basis: "Basis", lasso=np.inf, want_K=False):
        """kp_sweep_eval: fit + model + validation rollout + normalised mean error of every system for one dictionary.
        Returns err (nb, n) [, K (nb, W, W)], status (nb,)."""
        err = np.zeros((self.nb, self.n)); st = np.zeros(self.nb, dtype=np.int32)
        W = basis.W
        K = np.zeros((self.nb, W, W)) if want_K else None
        las = 1e6 if (lasso is None or not np.isfinite(lasso)) else float(lasso)
        F.check(F.lib().kp_sweep_eval(self.ctx.handle, self._h, basis.handle, las, F.dptr(err), F.dptr(K), st.ctypes.data_as(F.c_ip)),
                self.ctx.handle)
        if want_K:
            return err, np.transpose(K, (0, 2, 1)), st
        return err, st

    def sweep_eval_nested(self, basis: "Basis", n_deg, lasso=np.inf):
        """kp_sweep_eval_nested: degrees 1..n_deg of the polynomial dictionary `basis` (the highest degree) from one pass
        over the data.  Returns err (n_deg, nb, n), status (n_deg, nb)."""
        err = np.zeros((n_deg, self.nb, self.n)); st = np.zeros((n_deg, self.nb), dtype=np.int32)
        las = 1e6 if (lasso is None or not np.isfinite(lasso)) else float(lasso)
        F.check(F.lib().kp_sweep_eval_nested(self.ctx.handle, self._h, basis.handle, las, int(n_deg), F.dptr(err), st.ctypes.data_as(F.c_ip)),
                self.ctx.handle)
        self._last_nested = (basis.W, int(n_deg))
        return err, st

    def nested_K(self, deg_index, W):
        """K (monomial basis, W x W per system) of degree deg_index + 1 of the most recent sweep_eval_nested call."""
        Wmax, n_deg = self._last_nested
        K = np.zeros((self.nb, W, W))
        F.check(F.lib().kp_sweep_nested_get_K(self.ctx.handle, self.nb, Wmax, n_deg, int(deg_index), int(W), F.dptr(K)), self.ctx.handle)
        return np.transpose(K, (0, 2, 1))

    def close(self):
        if self._h:
            F.lib().kp_traj_destroy(self._h)
            self._h = F.vp()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def fit_gram(ctx: Context, basis: Basis, snaps: Snapshots, fetch=True):
    W = basis.W
    if not fetch:
        F.check(F.lib().kp_fit_gram(ctx.handle, basis.handle, snaps.handle, None, None), ctx.handle)
        return None, None
    G = np.zeros((W, W), order="F"); Cm = np.zeros((W, W), order="F")
    F.check(F.lib().kp_fit_gram(ctx.handle, basis.handle, snaps.handle, F.dptr(G), F.dptr(Cm)), ctx.handle)
    return G, Cm


def fit(ctx: Context, basis: Basis, snaps: Snapshots, lasso=None, fetch=True):
    """get_Koopman on resident snapshots for each lasso value (None/inf/>=1e6 => least squares)."""
    if lasso is None:
        lasso = [np.inf]
    las = np.ascontiguousarray(np.atleast_1d(np.asarray(lasso, dtype=np.float64)))
    W = basis.W
    K = np.zeros((len(las), W, W)) if fetch else None  # each W x W block column-major
    F.check(F.lib().kp_fit(ctx.handle, basis.handle, snaps.handle, F.dptr(las), len(las), F.dptr(K)), ctx.handle)
    if not fetch:
        return None
    return [np.asfortranarray(K[i].T) for i in range(len(las))]


def fit_sharded(ctx: Context, basis: Basis, snaps_local: Snapshots, lasso=None):
    """ONE fit whose snapshot pairs are sharded over the ranks of the context's RCCL communicator (pattern 2;
    kp_fit_sharded): fused Gram kernel on the local shard, one device-to-device all-reduce of [G | C], the same solve on every
    rank - identical K everywhere.  Without a communicator this is `fit` on the local snapshots."""
    if lasso is None:
        lasso = [np.inf]
    las = np.ascontiguousarray(np.atleast_1d(np.asarray(lasso, dtype=np.float64)))
    W = basis.W
    K = np.zeros((len(las), W, W))
    F.check(F.lib().kp_fit_sharded(ctx.handle, basis.handle, snaps_local.handle, F.dptr(las), len(las), F.dptr(K)), ctx.handle)
    return [np.asfortranarray(K[i].T) for i in range(len(las))]


def fit_gram_sharded(ctx: Context, basis: Basis, snaps_local: Snapshots):
    """(G, C) of the union of the ranks' shards: local Gram kernel + one all-reduce (kp_fit_gram_sharded)."""
    W = basis.W
    G = np.zeros((W, W), order="F"); Cm = np.zeros((W, W), order="F")
    F.check(F.lib().kp_fit_gram_sharded(ctx.handle, basis.handle, snaps_local.handle, F.dptr(G), F.dptr(Cm)), ctx.handle)
    return G, Cm


def fit_refine(ctx: Context, basis: Basis, snaps: Snapshots, K, steps=1):
    """kp_fit_refine: `steps` x  K += G^-1 Px'(Py - Px K) with the residual taken from the data (QR-level accuracy for
    ill-conditioned dictionaries; MATLAB's mldivide is a QR solve, Ksysid.m:1069)."""
    Kc = F.fcol(np.array(K, dtype=np.float64))
    F.check(F.lib().kp_fit_refine(ctx.handle, basis.handle, snaps.handle, int(steps), F.dptr(Kc)), ctx.handle)
    return Kc


class Mpc:
    """kp_mpc: condensed MPC problem of a linear / bilinear Koopman model on the device."""

    def __init__(self, ctx: Context, model_type, A, B, Np, proj, q_run, q_term, r, lo=None, hi=None,
                 slope_lim=None, smooth_lim=None):
        self.ctx = ctx
        ctx._children.add(self)
        A = F.fcol(A); B = F.fcol(B); proj = F.fcol(np.atleast_2d(proj))
        self.N, self.m, self.Np, self.nproj = A.shape[0], len(np.atleast_1d(r)), int(Np), proj.shape[0]
        self._zcall = None
        r = np.ascontiguousarray(np.atleast_1d(r), dtype=np.float64)
        lo_ = None if lo is None else np.ascontiguousarray(lo, dtype=np.float64)
        hi_ = None if hi is None else np.ascontiguousarray(hi, dtype=np.float64)
        nan = float("nan")
        self._h = F.vp()
        F.check(F.lib().kp_mpc_create(ctx.handle, F.MODEL[model_type], F.dptr(A), F.dptr(B), self.N, self.m, self.Np,
                                      F.dptr(proj), self.nproj, float(q_run), float(q_term), F.dptr(r), F.dptr(lo_),
                                      F.dptr(hi_), nan if slope_lim is None else float(slope_lim),
                                      nan if smooth_lim is None else float(smooth_lim), C.byref(self._h)), ctx.handle)
        nv, nr = C.c_int(), C.c_int()
        F.check(F.lib().kp_mpc_dims(self._h, C.byref(nv), C.byref(nr)))
        self.nvar, self.nrows = nv.value, nr.value

    @property
    def handle(self):
        return self._h

    def set_state_bounds(self, lo, hi):
        """kp_mpc_set_state_bounds: scaled-down bounds on the first n outputs (Kmpc.m:300-318); lo = None removes them."""
        if lo is None:
            F.check(F.lib().kp_mpc_set_state_bounds(self._h, 0, None, None), self.ctx.handle)
            return
        lo_ = np.ascontiguousarray(lo, dtype=np.float64); hi_ = np.ascontiguousarray(hi, dtype=np.float64)
        F.check(F.lib().kp_mpc_set_state_bounds(self._h, len(lo_), F.dptr(lo_), F.dptr(hi_)), self.ctx.handle)

    def step(self, z, u_prev, Yr, iters=1):
        """Returns (U [Np x m], status).  U is NaN when the QP failed (quadprog_gurobi.m:22-23)."""
        z = np.ascontiguousarray(z, dtype=np.float64); up = np.ascontiguousarray(u_prev, dtype=np.float64)
        yr = np.ascontiguousarray(Yr, dtype=np.float64)
        U = np.zeros((self.Np, self.m), order="F")
        st = C.c_int()
        F.check(F.lib().kp_mpc_step(self._h, F.dptr(z), F.dptr(up), F.dptr(yr), int(iters), F.dptr(U), C.byref(st)),
                self.ctx.handle)
        return U, st.value

    def step_zeta(self, basis: Basis, zeta, u_prev, Yr, iters=1):
        """One controller call with the lift fused in front (Kmpc.m:842).  This is the call inside a closed loop, so the
        marshalling is kept off its path: argument buffers and their ctypes pointers are made once per (controller,
        dictionary) and the inputs are copied into them."""
        cb = self._zcall
        if cb is None or cb[0] is not basis:
            zb = np.zeros(basis.nzeta); ub = np.zeros(self.m); yb = np.zeros(self.nproj * (self.Np + 1))
            Ub = np.zeros((self.Np, self.m), order="F"); zo = np.zeros(self.N); st = C.c_int()
            cb = self._zcall = (basis, zb, ub, yb, Ub, zo, st, F.dptr(zb), F.dptr(ub), F.dptr(yb), F.dptr(Ub), F.dptr(zo), C.byref(st),
                                F.lib().kp_mpc_step_zeta)
        _, zb, ub, yb, Ub, zo, st, pz, pu, py, pU, pzo, pst, fn = cb
        zb[...] = zeta; ub[...] = u_prev; yb[...] = Yr
        rc = fn(self._h, basis._h, pz, pu, py, int(iters), pU, pzo, pst)
        if rc:
            F.check(rc, self.ctx.handle)
        return Ub.copy(order="F"), zo.copy(), st.value

    def step_batch(self, Z, U_prev, YR):
        """Z (nb,N), U_prev (nb,m), YR (nb, nproj*(Np+1)) -> U (nb, Np, m), status (nb,)."""
        Z = np.ascontiguousarray(Z, dtype=np.float64); UP = np.ascontiguousarray(U_prev, dtype=np.float64)
        YR = np.ascontiguousarray(YR, dtype=np.float64)
        nb = Z.shape[0]
        U = np.zeros((nb, self.m, self.Np))          # each problem: Np x m column-major
        st = np.zeros(nb, dtype=np.int32)
        F.check(F.lib().kp_mpc_step_batch(self._h, nb, F.dptr(Z), F.dptr(UP), F.dptr(YR), F.dptr(U),
                                          st.ctypes.data_as(F.c_ip)), self.ctx.handle)
        return np.transpose(U, (0, 2, 1)), st

    def last_qp(self):
        """(Hq, f, Aq, bq) of the most recent single-problem step: quadprog(Hq, f, Aq, bq)."""
        Hq = np.zeros((self.nvar, self.nvar), order="F"); f = np.zeros(self.nvar)
        Aq = np.zeros((self.nrows, self.nvar), order="F"); bq = np.zeros(self.nrows)
        F.check(F.lib().kp_mpc_last_qp(self._h, F.dptr(Hq), F.dptr(f), F.dptr(Aq), F.dptr(bq)), self.ctx.handle)
        return Hq, f, Aq, bq

    def close(self):
        if self._h:
            F.lib().kp_mpc_destroy(self._h)
            self._h = F.vp()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
