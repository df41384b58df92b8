"""ctypes binding of oracle/libkoopman_cpu.so: the CPU backend of the SAME C ABI (include/koopman_hip.h) for the fit path -
TEST INFRASTRUCTURE / REPORTED CPU BASELINE ONLY (bench.py's cpu_baseline leg, tests/test_oracle_c.py).  The argument types are
the product binding's own table (koopman_realizations_amd._ffi.SIGNATURES), restricted to what the CPU library exports, so the
baseline is called exactly as the GPU library is: kp_create -> kp_basis_create -> kp_snapshots_upload -> kp_fit."""
import ctypes as C
import os

import numpy as np

_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "libkoopman_cpu.so")
EXPORTS = ("kp_device_count", "kp_create", "kp_destroy", "kp_last_error", "kp_basis_create", "kp_basis_destroy", "kp_basis_dims",
           "kp_snapshots_upload", "kp_snapshots_destroy", "kp_lift", "kp_fit", "kp_fit_gram")
_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_PATH):
            raise ImportError(f"{_PATH} not built (make -C oracle)")
        from koopman_realizations_amd import _ffi as F
        l = C.CDLL(_PATH)
        for name in EXPORTS:
            fn = getattr(l, name)
            fn.restype, fn.argtypes = F.SIGNATURES[name]
        l.ko_set_threads.argtypes = [C.c_int]
        l.ko_max_threads.restype = C.c_int
        _lib = l
    return _lib


class CpuFit:
    """One dictionary + one snapshot set on the CPU backend, through the ABI calls a MATLAB / Python host would make."""

    def __init__(self, model_type, nzeta, m, exps, alpha, beta, u):
        from koopman_realizations_amd import _ffi as F
        l = lib()
        self._F, self._l = F, l
        self.ctx = F.vp(); self.basis = F.vp(); self.snaps = F.vp()
        self._chk(l.kp_create(0, C.byref(self.ctx)))
        self._ex = np.ascontiguousarray(exps, dtype=np.uint8)
        self._bt = np.array([0], dtype=np.int32); self._bc = np.array([self._ex.shape[0]], dtype=np.int32)
        d = F.KpBasisDesc()
        d.model_type, d.nzeta, d.m, d.n_blocks = F.MODEL[model_type], nzeta, m, 1
        d.block_type = self._bt.ctypes.data_as(C.POINTER(C.c_int32)); d.block_count = self._bc.ctypes.data_as(C.POINTER(C.c_int32))
        d.poly_exps = self._ex.ctypes.data_as(C.POINTER(C.c_uint8)); d.gauss_centres = None; d.k_pcs = 0; d.pcs = None
        self._chk(l.kp_basis_create(self.ctx, C.byref(d), C.byref(self.basis)))
        nv, nf, N, W = C.c_int(), C.c_int(), C.c_int(), C.c_int()
        self._chk(l.kp_basis_dims(self.basis, C.byref(nv), C.byref(nf), C.byref(N), C.byref(W)))
        self.N, self.W = N.value, W.value
        a, b, uu = F.fcol(alpha), F.fcol(beta), F.fcol(u)
        self._chk(l.kp_snapshots_upload(self.ctx, F.dptr(a), F.dptr(b), F.dptr(uu), a.shape[0], nzeta, m, C.byref(self.snaps)))

    def _chk(self, rc):
        if rc:
            raise RuntimeError(f"CPU backend error {rc}: {self._l.kp_last_error(self.ctx).decode()}")

    def fit(self):
        """K = Px \\ Py (kp_fit, least squares)."""
        K = np.zeros((self.W, self.W), order="F")
        inf = np.array([np.inf])
        self._chk(self._l.kp_fit(self.ctx, self.basis, self.snaps, self._F.dptr(inf), 1, self._F.dptr(K)))
        return K

    def gram(self):
        G = np.zeros((self.W, self.W), order="F"); Cm = np.zeros((self.W, self.W), order="F")
        self._chk(self._l.kp_fit_gram(self.ctx, self.basis, self.snaps, self._F.dptr(G), self._F.dptr(Cm)))
        return G, Cm

    def lift(self, what, zeta, u=None):
        F = self._F
        z = F.fcol(np.atleast_2d(zeta)); uu = None if u is None else F.fcol(np.atleast_2d(u))
        out = np.zeros((z.shape[0], self.W if what == F.LIFT_ROW else self.N), order="F")
        self._chk(self._l.kp_lift(self.ctx, self.basis, what, F.dptr(z), F.dptr(uu), z.shape[0], F.dptr(out)))
        return out

    def close(self):
        if self.snaps:
            self._l.kp_snapshots_destroy(self.snaps); self.snaps = None
        if self.basis:
            self._l.kp_basis_destroy(self.basis); self.basis = None
        if self.ctx:
            self._l.kp_destroy(self.ctx); self.ctx = None
