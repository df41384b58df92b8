"""ctypes binding of oracle/libkoopman_oracle.so (plain-C restatement of the fit; TEST INFRASTRUCTURE / CPU baseline only)."""
import ctypes as C
import os

import numpy as np

_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "libkoopman_oracle.so")
_lib = None
MODEL = {"linear": 0, "bilinear": 1, "nonlinear": 2}


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_PATH):
            raise ImportError(f"{_PATH} not built (make -C oracle)")
        l = C.CDLL(_PATH)
        dp, bp = C.POINTER(C.c_double), C.POINTER(C.c_uint8)
        l.ko_get_koopman.restype = C.c_int
        l.ko_get_koopman.argtypes = [C.c_int, C.c_int, C.c_int, bp, C.c_int, dp, dp, dp, C.c_int64, dp]
        l.ko_lift_rows.restype = C.c_int
        l.ko_lift_rows.argtypes = [C.c_int, C.c_int, C.c_int, bp, C.c_int, dp, dp, C.c_int64, dp]
        l.ko_set_threads.argtypes = [C.c_int]
        l.ko_max_threads.restype = C.c_int
        _lib = l
    return _lib


def _d(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def get_koopman(model_type, nzeta, m, exps, alpha, beta, u):
    """K = Px \\ Py with the per-row lift of Ksysid.m:1030-1065 and a Householder QR solve (:1069).  exps: monomial rows
    after the first nvars (uint8, rows x nvars)."""
    ex = np.ascontiguousarray(exps, dtype=np.uint8)
    a = np.asfortranarray(alpha, dtype=np.float64); b = np.asfortranarray(beta, dtype=np.float64); uu = np.asfortranarray(u, dtype=np.float64)
    nv = nzeta + (m if model_type == "nonlinear" else 0)
    N = nv + ex.shape[0] + 1
    W = N + m if model_type == "linear" else N * (m + 1) if model_type == "bilinear" else N
    K = np.zeros((W, W), order="F")
    rc = lib().ko_get_koopman(MODEL[model_type], nzeta, m, ex.ctypes.data_as(C.POINTER(C.c_uint8)), ex.shape[0], _d(a), _d(b), _d(uu), a.shape[0], _d(K))
    if rc:
        raise RuntimeError(f"ko_get_koopman failed: {rc}")
    return K


def lift_rows(model_type, nzeta, m, exps, zeta, u):
    ex = np.ascontiguousarray(exps, dtype=np.uint8)
    z = np.asfortranarray(zeta, dtype=np.float64); uu = np.asfortranarray(u, dtype=np.float64)
    nv = nzeta + (m if model_type == "nonlinear" else 0)
    N = nv + ex.shape[0] + 1
    W = N + m if model_type == "linear" else N * (m + 1) if model_type == "bilinear" else N
    P = np.zeros((z.shape[0], W), order="F")
    rc = lib().ko_lift_rows(MODEL[model_type], nzeta, m, ex.ctypes.data_as(C.POINTER(C.c_uint8)), ex.shape[0], _d(z), _d(uu), z.shape[0], _d(P))
    if rc:
        raise RuntimeError(f"ko_lift_rows failed: {rc}")
    return P
