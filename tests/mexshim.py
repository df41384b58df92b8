"""Test infrastructure: drives matlab/kp_mex.c's mexFunction from Python.

matlab/kp_mex.c is compiled unchanged against tests/mex_shim/mex.h (a functional stand-in for MATLAB's MEX API) and linked
with libkoopman_hip.so into tests/mex_shim/kp_mex_shim.so.  `kp_mex(cmd, *args, nargout=...)` marshals numpy values into
mxArrays the way MATLAB would hold them (column-major, vectors as columns unless 2-D, handles as uint64 scalars, structs from
dicts), calls mexFunction through the shim and converts the outputs back.  mexErrMsgIdAndTxt surfaces as MexError."""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SHIM_DIR = os.path.join(ROOT, "tests", "mex_shim")
SHIM_SO = os.path.join(SHIM_DIR, "kp_mex_shim.so")
GATEWAY = os.path.join(ROOT, "matlab", "kp_mex.c")

# mxClassID values of tests/mex_shim/mex.h
CLS = {"char": 4, "double": 6, "uint8": 9, "int32": 12, "uint64": 15, "struct": 2}
NP_OF = {6: np.float64, 9: np.uint8, 12: np.int32, 15: np.uint64, 4: np.uint16}


class MexError(RuntimeError):
    def __init__(self, ident, msg):
        super().__init__(f"{ident}: {msg}")
        self.identifier = ident
        self.message = msg


class Handle(int):
    """An opaque uint64 handle as kp_mex returns it."""


def build(force=False):
    """gcc on matlab/kp_mex.c + the shim (no MATLAB involved); rebuilt when a source is newer than the .so."""
    lib = os.path.join(ROOT, "koopman-realizations_amd", "libkoopman_hip.so")
    srcs = [GATEWAY, os.path.join(SHIM_DIR, "mex_shim.c"), os.path.join(SHIM_DIR, "mex.h"), os.path.join(ROOT, "include", "koopman_hip.h"), lib]
    if not force and os.path.exists(SHIM_SO) and all(os.path.getmtime(SHIM_SO) >= os.path.getmtime(s) for s in srcs):
        return SHIM_SO
    cmd = ["gcc", "-O1", "-g", "-Wall", "-Wextra", "-Werror", "-fPIC", "-shared", "-I", SHIM_DIR, "-I", os.path.join(ROOT, "include"), GATEWAY,
           os.path.join(SHIM_DIR, "mex_shim.c"), "-o", SHIM_SO, "-L", os.path.dirname(lib), "-lkoopman_hip", "-lm",
           "-Wl,-rpath," + os.path.dirname(lib), "-Wl,--no-undefined"]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode:
        raise RuntimeError("building the MEX shim failed:\n" + r.stdout + r.stderr)
    return SHIM_SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        l = C.CDLL(build())
        l.shim_new.restype = C.c_void_p
        l.shim_new.argtypes = [C.c_int, C.c_int, C.POINTER(C.c_size_t)]
        l.shim_new_struct.restype = C.c_void_p
        l.shim_set_field.argtypes = [C.c_void_p, C.c_char_p, C.c_void_p]
        l.shim_call.argtypes = [C.c_int, C.POINTER(C.c_void_p), C.c_int, C.POINTER(C.c_void_p)]
        for f in ("shim_error_id", "shim_error_msg", "shim_warning_id", "shim_warning_msg"):
            getattr(l, f).restype = C.c_char_p
        l.mxGetData.restype = C.c_void_p
        l.mxGetData.argtypes = [C.c_void_p]
        l.mxGetClassID.argtypes = [C.c_void_p]
        l.mxGetNumberOfDimensions.restype = C.c_size_t
        l.mxGetNumberOfDimensions.argtypes = [C.c_void_p]
        l.mxGetDimensions.restype = C.POINTER(C.c_size_t)
        l.mxGetDimensions.argtypes = [C.c_void_p]
        l.mxDestroyArray.argtypes = [C.c_void_p]
        _lib = l
    return _lib


def _new(cls, dims):
    d = (C.c_size_t * len(dims))(*dims)
    return lib().shim_new(cls, len(dims), d)


def to_mx(v):
    """numpy / Python value -> mxArray* (caller destroys)."""
    l = lib()
    if isinstance(v, Handle):
        a = _new(CLS["uint64"], (1, 1))
        C.cast(l.mxGetData(a), C.POINTER(C.c_uint64))[0] = int(v)
        return a
    if isinstance(v, str):
        a = _new(CLS["char"], (1, len(v)))
        if v:
            buf = np.frombuffer(v.encode("latin1"), dtype=np.uint8).astype(np.uint16)
            C.memmove(l.mxGetData(a), buf.ctypes.data, buf.nbytes)
        return a
    if isinstance(v, dict):
        s = l.shim_new_struct()
        for k, x in v.items():
            l.shim_set_field(s, k.encode(), to_mx(x))
        return s
    if v is None:
        return _new(CLS["double"], (0, 0))
    arr = np.asarray(v)
    if arr.dtype == np.uint8:
        cls = CLS["uint8"]
    elif arr.dtype == np.int32:
        cls = CLS["int32"]
    elif arr.dtype == np.uint64:
        cls = CLS["uint64"]
    else:
        cls = CLS["double"]
        arr = arr.astype(np.float64)
    if arr.ndim == 0:
        dims = (1, 1)
    elif arr.ndim == 1:
        dims = (arr.shape[0], 1) if arr.shape[0] else (0, 0)          # vectors are columns, [] is 0 x 0
    else:
        dims = arr.shape
    a = _new(cls, dims)
    if arr.size:
        f = np.asfortranarray(arr)
        C.memmove(l.mxGetData(a), f.ctypes.data, f.nbytes)
    return a


def from_mx(a):
    l = lib()
    cls = l.mxGetClassID(a)
    nd = l.mxGetNumberOfDimensions(a)
    dims = tuple(int(l.mxGetDimensions(a)[i]) for i in range(nd))
    n = int(np.prod(dims))
    dt = NP_OF[cls]
    out = np.zeros(n, dtype=dt)
    if n:
        C.memmove(out.ctypes.data, l.mxGetData(a), out.nbytes)
    if cls == CLS["char"]:
        return out.astype(np.uint8).tobytes().decode("latin1")
    if cls == CLS["uint64"] and n == 1:
        return Handle(int(out[0]))
    return out.reshape(dims, order="F")


last_warning = ("", "")


def kp_mex(cmd, *args, nargout=1):
    """kp_mex(cmd, args...) as MATLAB would call it; returns one value (nargout <= 1) or a tuple."""
    global last_warning
    l = lib()
    rhs = [to_mx(cmd)] + [to_mx(a) for a in args]
    prhs = (C.c_void_p * len(rhs))(*rhs)
    nout = max(nargout, 1)
    plhs = (C.c_void_p * nout)()
    rc = l.shim_call(nargout, plhs, len(rhs), prhs)
    for r in rhs:
        l.mxDestroyArray(r)
    if rc:
        raise MexError(l.shim_error_id().decode(), l.shim_error_msg().decode())
    last_warning = (l.shim_warning_id().decode(), l.shim_warning_msg().decode())
    outs = []
    for i in range(nout):
        if plhs[i]:
            outs.append(from_mx(plhs[i]))
            l.mxDestroyArray(plhs[i])
        else:
            outs.append(None)
    if nargout <= 1:
        return outs[0]
    return tuple(outs[:nargout])


def commands():
    """{name: (nrhs_min, nrhs_max, nlhs_max)} - the gateway's own table (runs without a GPU)."""
    tab, names = kp_mex("commands", nargout=2)
    return {nm: tuple(int(x) for x in tab[i]) for i, nm in enumerate(names.split("\n"))}
