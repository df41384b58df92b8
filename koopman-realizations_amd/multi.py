"""One caller, several GPUs: the host side of the kp_multi_* entry points (include/koopman_hip.h).

The reference's host is ONE MATLAB interpreter that runs its sweeps as serial loops - lasso values
(Ksysid.train_models, Ksysid.m:1372-1387), random systems (evaluate_rand_models.m:45-144), MPC problems.  `Multi` keeps
that shape: one process, one thread calling in; the library owns a worker thread and a context per listed device, deals
the units over them and has every device write its share of the result straight into the caller's arrays.  Marshalling
only - no arithmetic here."""
from __future__ import annotations

import ctypes as C
import weakref

import numpy as np

from . import _ffi as F
from .device import basis_desc, Traj


class Multi:
    """kp_multi: `device_ids` may name the same device more than once (two contexts on it; how a one-GPU box tests this)."""

    def __init__(self, device_ids):
        ids = np.ascontiguousarray(np.atleast_1d(device_ids), dtype=np.int32)
        self._h = F.vp()
        rc = F.lib().kp_multi_create(ids.ctypes.data_as(F.c_ip), len(ids), C.byref(self._h))
        F.check(rc)
        self.n_dev = len(ids)
        self.device_ids = [int(i) for i in ids]
        self._blocks = {}
        self._retired = []                   # page-locked blocks a host_array() call outgrew: kept until close() (views of them may live on)
        # trajectory / controller objects created from this one: kp_multi_destroy tears down the contexts they point into, so
        # close() releases them FIRST (include/koopman_hip.h: "destroy the objects before the kp_multi"), and their own
        # close() / __del__ afterwards is a no-op
        self._children = weakref.WeakSet()

    def _check(self, rc):
        if rc != F.KP_OK:
            msg = F.lib().kp_multi_last_error(self._h)
            raise F.KoopmanHipError(rc, msg.decode() if msg else "")

    @property
    def handle(self):
        return self._h

    def timers(self):
        """(n_dev, 4): per device [upload, device work, result transfer, whole job] of the most recent call, ms (wall)."""
        ms = np.zeros((self.n_dev, 4))
        self._check(F.lib().kp_multi_timers(self._h, F.dptr(ms)))
        return ms

    def host_array(self, name, shape):
        """float64 C-ordered array in memory that is page-locked for EVERY device (kp_multi_host_alloc), kept under `name`:
        results written there arrive by direct DMA from each device."""
        n = int(np.prod(shape))
        ent = self._blocks.get(name)
        if ent is None or ent[1] < n:
            if ent is not None:
                # arrays handed out earlier are views of the old block: it stays allocated until close()
                self._retired.append(ent[0])
            p = F.vp()
            cap = n + n // 8 + 512
            self._check(F.lib().kp_multi_host_alloc(self._h, cap * 8, C.byref(p)))
            ent = (p, cap, np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_double)), shape=(cap,)))
            self._blocks[name] = ent
        return ent[2][:n].reshape(shape)

    # ---- fits ---------------------------------------------------------------------------------------------------------
    def _fit(self, fn, dictionary, alpha, beta, u, lasso, out):
        model_type, nzeta, m, blocks, pcs = dictionary
        d, keep = basis_desc(model_type, nzeta, m, blocks, pcs)
        a = F.fcol(alpha); b = F.fcol(beta); uu = F.fcol(u)
        las = np.ascontiguousarray(np.atleast_1d(np.asarray(lasso, dtype=np.float64)))
        Wc = C.c_int()
        F.check(F.lib().kp_basis_desc_dims(C.byref(d), None, None, None, C.byref(Wc)))
        W = Wc.value
        K = out if out is not None else np.zeros((len(las), W, W))
        if K.shape != (len(las), W, W) or not K.flags.c_contiguous:
            raise ValueError(f"Multi.fit: the result stack must be a C-contiguous {(len(las), W, W)} array")
        self._check(fn(self._h, C.byref(d), F.dptr(a), F.dptr(b), F.dptr(uu), a.shape[0], F.dptr(las), len(las), F.dptr(K)))
        del keep
        return K

    def fit(self, dictionary, alpha, beta, u, lasso, out=None):
        """kp_multi_fit: the lasso grid of train_models (Ksysid.m:1372-1387), value i on device i mod n_dev.
        dictionary = (model_type, nzeta, m, blocks, pcs) as `Basis` takes them.  Returns the (n_lasso, W, W) stack of
        column-major blocks (K[i].T is value i in numpy's view); `out` may be a Multi.host_array block (direct DMA)."""
        return self._fit(F.lib().kp_multi_fit, dictionary, alpha, beta, u, lasso, out)

    def fit_sharded(self, dictionary, alpha, beta, u, lasso=(np.inf,), out=None):
        """kp_multi_fit_sharded: ONE fit, snapshot rows dealt over the devices, [G | C] summed on device 0."""
        return self._fit(F.lib().kp_multi_fit_sharded, dictionary, alpha, beta, u, lasso, out)

    # ---- random-system sweep ---------------------------------------------------------------------------------------------
    def traj_upload(self, Y, U, ntrials, Yv, Uv):
        """kp_multi_traj_upload: Y (nb, rows, n), U (nb, rows, m), Yv (nb, Tv, n), Uv (nb, Tv, m) as `Traj` takes them."""
        return MultiTraj(self, Y, U, ntrials, Yv, Uv)

    def close(self):
        if self._h:
            for child in list(self._children):
                child.close()
            for p in self._retired:
                F.lib().kp_multi_host_free(self._h, p)
            self._retired = []
            F.lib().kp_multi_destroy(self._h)
            self._h = F.vp()
            self._blocks = {}

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class MultiTraj:
    def __init__(self, mg: Multi, Y, U, ntrials, Yv, Uv):
        self.mg = mg
        Y = np.asarray(Y, dtype=np.float64); U = np.asarray(U, dtype=np.float64)
        Yv = np.asarray(Yv, dtype=np.float64); Uv = np.asarray(Uv, dtype=np.float64)
        self.nb, rows, self.n = Y.shape
        self.m = U.shape[2]
        self.ntrials, self.T, self.Tv = int(ntrials), rows // int(ntrials), Yv.shape[1]
        tr = Traj._blocks
        self._h = F.vp()
        mg._children.add(self)
        mg._check(F.lib().kp_multi_traj_upload(mg.handle, F.dptr(tr(Y)), F.dptr(tr(U)), self.nb, self.ntrials, self.T, self.n, self.m,
                                               F.dptr(tr(Yv)), F.dptr(tr(Uv)), self.Tv, C.byref(self._h)))

    def sweep_eval_nested(self, dictionary, n_deg, lasso=np.inf):
        """kp_multi_sweep_eval_nested: err (n_deg, nb, n), status (n_deg, nb) in system order."""
        model_type, nzeta, m, blocks, pcs = dictionary
        d, keep = basis_desc(model_type, nzeta, m, blocks, pcs)
        err = np.zeros((n_deg, self.nb, self.n)); st = np.zeros((n_deg, self.nb), dtype=np.int32)
        las = 1e6 if (lasso is None or not np.isfinite(lasso)) else float(lasso)
        self.mg._check(F.lib().kp_multi_sweep_eval_nested(self.mg.handle, self._h, C.byref(d), las, int(n_deg), F.dptr(err),
                                                          st.ctypes.data_as(F.c_ip)))
        del keep
        return err, st

    def close(self):
        if self._h and self.mg._h:           # (a closed parent has released this object already)
            F.lib().kp_multi_traj_destroy(self._h)
        self._h = F.vp()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class MultiMpc:
    """kp_multi_mpc: one controller replicated on every device, batches of problems dealt in contiguous chunks."""

    def __init__(self, mg: Multi, model_type, A, B, Np, proj, q_run, q_term, r, lo=None, hi=None, slope_lim=None, smooth_lim=None):
        self.mg = mg
        A = F.fcol(A); B = F.fcol(B); proj = F.fcol(np.atleast_2d(proj))
        self.N, self.m, self.Np, self.nproj = A.shape[0], len(np.atleast_1d(r)), int(Np), proj.shape[0]
        r = np.ascontiguousarray(np.atleast_1d(r), dtype=np.float64)
        lo_ = None if lo is None else np.ascontiguousarray(lo, dtype=np.float64)
        hi_ = None if hi is None else np.ascontiguousarray(hi, dtype=np.float64)
        nan = float("nan")
        self._h = F.vp()
        mg._children.add(self)
        mg._check(F.lib().kp_multi_mpc_create(mg.handle, F.MODEL[model_type], F.dptr(A), F.dptr(B), self.N, self.m, self.Np, F.dptr(proj),
                                              self.nproj, float(q_run), float(q_term), F.dptr(r), F.dptr(lo_), F.dptr(hi_),
                                              nan if slope_lim is None else float(slope_lim),
                                              nan if smooth_lim is None else float(smooth_lim), C.byref(self._h)))

    def set_state_bounds(self, lo, hi):
        if lo is None:
            self.mg._check(F.lib().kp_multi_mpc_set_state_bounds(self._h, 0, None, None))
            return
        lo_ = np.ascontiguousarray(lo, dtype=np.float64); hi_ = np.ascontiguousarray(hi, dtype=np.float64)
        self.mg._check(F.lib().kp_multi_mpc_set_state_bounds(self._h, len(lo_), F.dptr(lo_), F.dptr(hi_)))

    def step_batch(self, Z, U_prev, YR):
        """Z (nb, N), U_prev (nb, m), YR (nb, nproj (Np + 1)) -> U (nb, Np, m), status (nb,) - as Mpc.step_batch."""
        Z = np.ascontiguousarray(Z, dtype=np.float64); UP = np.ascontiguousarray(U_prev, dtype=np.float64)
        YR = np.ascontiguousarray(YR, dtype=np.float64)
        nb = Z.shape[0]
        U = np.zeros((nb, self.m, self.Np))
        st = np.zeros(nb, dtype=np.int32)
        self.mg._check(F.lib().kp_multi_mpc_step_batch(self._h, nb, F.dptr(Z), F.dptr(UP), F.dptr(YR), F.dptr(U), st.ctypes.data_as(F.c_ip)))
        return np.transpose(U, (0, 2, 1)), st

    def close(self):
        if self._h and self.mg._h:           # (a closed parent has released this object already)
            F.lib().kp_multi_mpc_destroy(self._h)
        self._h = F.vp()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
