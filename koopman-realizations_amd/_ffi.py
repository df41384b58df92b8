"""ctypes binding of libkoopman_hip.so (the C ABI declared in include/koopman_hip.h).

The library is the product; there is no CPU fallback.  If the shared object is missing
or cannot be loaded this module raises, and every compute call fails loudly when no HIP
device is present (kp_create returns KP_ERR_HIP).
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("KP_LIB_PATH") or os.path.join(_HERE, "libkoopman_hip.so")     # KP_LIB_PATH: an experimental build (tools/)

KP_OK, KP_ERR_ARG, KP_ERR_HIP, KP_ERR_NOT_SPD, KP_ERR_QP_FAIL, KP_ERR_NOT_CONVERGED = 0, -1, -2, -3, -4, -5
MODEL = {"linear": 0, "bilinear": 1, "nonlinear": 2}
BLOCK = {"poly": 0, "fourier": 1, "gaussian": 2, "hermite": 3, "fourier_sparser": 4}
LIFT_FULL, LIFT_ECON, LIFT_ROW = 0, 1, 2

c_dp = C.POINTER(C.c_double)
c_ip = C.POINTER(C.c_int)
vp = C.c_void_p


class KpBasisDesc(C.Structure):
    _fields_ = [("model_type", C.c_int32), ("nzeta", C.c_int32), ("m", C.c_int32), ("n_blocks", C.c_int32),
                ("block_type", C.POINTER(C.c_int32)), ("block_count", C.POINTER(C.c_int32)),
                ("poly_exps", C.POINTER(C.c_uint8)), ("gauss_centres", c_dp),
                ("k_pcs", C.c_int32), ("pcs", c_dp)]


# name -> (restype, argtypes); every symbol of include/koopman_hip.h
SIGNATURES = {
    "kp_device_count": (C.c_int, [c_ip]),
    "kp_create": (C.c_int, [C.c_int, C.POINTER(vp)]),
    "kp_destroy": (C.c_int, [vp]),
    "kp_last_error": (C.c_char_p, [vp]),
    "kp_device_info": (C.c_int, [vp, C.c_char_p, C.c_int, c_ip, C.POINTER(C.c_int64)]),
    "kp_timer_get": (C.c_int, [vp, C.c_int, c_dp]),
    "kp_stream": (vp, [vp]),
    "kp_basis_create": (C.c_int, [vp, C.POINTER(KpBasisDesc), C.POINTER(vp)]),
    "kp_basis_destroy": (C.c_int, [vp]),
    "kp_basis_dims": (C.c_int, [vp, c_ip, c_ip, c_ip, c_ip]),
    "kp_basis_desc_dims": (C.c_int, [C.POINTER(KpBasisDesc), c_ip, c_ip, c_ip, c_ip]),
    "kp_lift": (C.c_int, [vp, vp, C.c_int, c_dp, c_dp, C.c_int64, c_dp]),
    "kp_snapshots_upload": (C.c_int, [vp, c_dp, c_dp, c_dp, C.c_int64, C.c_int, C.c_int, C.POINTER(vp)]),
    "kp_snapshots_update": (C.c_int, [vp, vp, c_dp, c_dp, c_dp, C.c_int64]),
    "kp_snapshots_destroy": (C.c_int, [vp]),
    "kp_host_alloc": (C.c_int, [vp, C.c_int64, C.POINTER(vp)]),
    "kp_host_free": (C.c_int, [vp, vp]),
    "kp_fit_gram": (C.c_int, [vp, vp, vp, c_dp, c_dp]),
    "kp_fit_solve": (C.c_int, [vp, c_dp, c_dp, C.c_int, C.c_int, c_dp]),
    "kp_fit_last_rank": (C.c_int, [vp, c_ip]),
    "kp_fit_last_pivot_ratio": (C.c_int, [vp, c_dp]),
    "kp_fit_lasso": (C.c_int, [vp, c_dp, c_dp, C.c_int, C.c_int, C.c_double, C.c_int, C.c_double, c_dp, c_ip]),
    "kp_fit_lasso_batch": (C.c_int, [vp, c_dp, c_dp, C.c_int, C.c_int, c_dp, C.c_int, C.c_int, C.c_double, c_dp, c_ip]),
    "kp_fit": (C.c_int, [vp, vp, vp, c_dp, C.c_int, c_dp]),
    "kp_synchronize": (C.c_int, [vp]),
    "kp_fit_async_slots": (C.c_int, [vp, C.c_int]),
    "kp_fit_get_K": (C.c_int, [vp, C.c_int, C.c_int, c_dp]),
    "kp_fit_batch": (C.c_int, [vp, vp, vp, C.c_int, C.c_int64, c_dp, c_dp, c_dp, C.POINTER(C.c_int)]),
    "kp_traj_upload": (C.c_int, [vp, c_dp, c_dp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, c_dp, c_dp, C.c_int, C.POINTER(vp)]),
    "kp_traj_destroy": (C.c_int, [vp]),
    "kp_traj_create": (C.c_int, [vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(vp)]),
    "kp_traj_put": (C.c_int, [vp, C.c_int, c_dp]),
    "kp_traj_finish": (C.c_int, [vp]),
    "kp_traj_scale": (C.c_int, [vp, c_dp]),
    "kp_traj_dims": (C.c_int, [vp, c_ip, c_ip, c_ip, c_ip, c_ip, c_ip]),
    "kp_sweep_eval": (C.c_int, [vp, vp, vp, C.c_double, c_dp, c_dp, c_ip]),
    "kp_sweep_eval_nested": (C.c_int, [vp, vp, vp, C.c_double, C.c_int, c_dp, c_ip]),
    "kp_sweep_nested_get_K": (C.c_int, [vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, c_dp]),
    "kp_fit_refine": (C.c_int, [vp, vp, vp, C.c_int, c_dp]),
    "kp_sym_eig": (C.c_int, [vp, c_dp, C.c_int, c_dp, c_dp, C.POINTER(C.c_int)]),
    "kp_mpc_set_state_bounds": (C.c_int, [vp, C.c_int, c_dp, c_dp]),
    "kp_model_project_batch": (C.c_int, [vp, c_dp, c_dp, c_dp, C.c_int, C.c_int, C.c_int, c_dp, c_dp, c_dp, C.POINTER(C.c_int)]),
    "kp_model_project": (C.c_int, [vp, c_dp, c_dp, c_dp, C.c_int, C.c_int, c_dp, c_dp, c_dp]),
    "kp_rollout": (C.c_int, [vp, C.c_int, C.c_int, c_dp, c_dp, C.c_int, C.c_int, c_dp, c_dp, C.c_int, C.c_int, c_dp]),
    "kp_rollout_nl": (C.c_int, [vp, vp, C.c_int, c_dp, c_dp, c_dp, C.c_int, c_dp]),
    "kp_mpc_create": (C.c_int, [vp, C.c_int, c_dp, c_dp, C.c_int, C.c_int, C.c_int, c_dp, C.c_int, C.c_double,
                                C.c_double, c_dp, c_dp, c_dp, C.c_double, C.c_double, C.POINTER(vp)]),
    "kp_mpc_destroy": (C.c_int, [vp]),
    "kp_mpc_dims": (C.c_int, [vp, c_ip, c_ip]),
    "kp_mpc_step": (C.c_int, [vp, c_dp, c_dp, c_dp, C.c_int, c_dp, c_ip]),
    "kp_mpc_step_zeta": (C.c_int, [vp, vp, c_dp, c_dp, c_dp, C.c_int, c_dp, c_dp, c_ip]),
    "kp_mpc_step_batch": (C.c_int, [vp, C.c_int, c_dp, c_dp, c_dp, c_dp, c_ip]),
    "kp_mpc_last_qp": (C.c_int, [vp, c_dp, c_dp, c_dp, c_dp]),
    "kp_mpc_last_profile": (C.c_int, [vp, c_dp, c_ip]),
    "kp_mpc_last_stamps": (C.c_int, [vp, c_dp]),
    "kp_qp_solve": (C.c_int, [vp, c_dp, c_dp, c_dp, c_dp, C.c_int, C.c_int, c_dp, c_ip]),
    "kp_comm_unique_id": (C.c_int, [vp]),
    "kp_comm_create": (C.c_int, [vp, vp, C.c_int, C.c_int]),
    "kp_comm_destroy": (C.c_int, [vp]),
    "kp_comm_info": (C.c_int, [vp, c_ip, c_ip]),
    "kp_comm_allgather": (C.c_int, [vp, vp, C.c_int64, vp]),
    "kp_comm_allreduce_sum": (C.c_int, [vp, c_dp, C.c_int64]),
    "kp_comm_abandon": (C.c_int, [vp]),
    "kp_comm_allgather_fit": (C.c_int, [vp, C.c_int, C.c_int, c_dp]),
    "kp_comm_allgather_fits": (C.c_int, [vp, C.c_int, C.c_int, C.c_int, c_dp]),
    "kp_comm_gather_fits": (C.c_int, [vp, C.c_int, C.c_int, C.c_int, C.c_int, c_dp]),
    "kp_fit_sharded": (C.c_int, [vp, vp, vp, c_dp, C.c_int, c_dp]),
    "kp_fit_gram_sharded": (C.c_int, [vp, vp, vp, c_dp, c_dp]),
    "kp_multi_create": (C.c_int, [c_ip, C.c_int, C.POINTER(vp)]),
    "kp_multi_destroy": (C.c_int, [vp]),
    "kp_multi_size": (C.c_int, [vp, c_ip]),
    "kp_multi_ctx": (vp, [vp, C.c_int]),
    "kp_multi_last_error": (C.c_char_p, [vp]),
    "kp_multi_host_alloc": (C.c_int, [vp, C.c_int64, C.POINTER(vp)]),
    "kp_multi_host_free": (C.c_int, [vp, vp]),
    "kp_multi_timers": (C.c_int, [vp, c_dp]),
    "kp_multi_fit": (C.c_int, [vp, C.POINTER(KpBasisDesc), c_dp, c_dp, c_dp, C.c_int64, c_dp, C.c_int, c_dp]),
    "kp_multi_fit_sharded": (C.c_int, [vp, C.POINTER(KpBasisDesc), c_dp, c_dp, c_dp, C.c_int64, c_dp, C.c_int, c_dp]),
    "kp_multi_traj_upload": (C.c_int, [vp, c_dp, c_dp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, c_dp, c_dp, C.c_int, C.POINTER(vp)]),
    "kp_multi_traj_destroy": (C.c_int, [vp]),
    "kp_multi_sweep_eval_nested": (C.c_int, [vp, vp, C.POINTER(KpBasisDesc), C.c_double, C.c_int, c_dp, c_ip]),
    "kp_multi_mpc_create": (C.c_int, [vp, C.c_int, c_dp, c_dp, C.c_int, C.c_int, C.c_int, c_dp, C.c_int, C.c_double,
                                      C.c_double, c_dp, c_dp, c_dp, C.c_double, C.c_double, C.POINTER(vp)]),
    "kp_multi_mpc_set_state_bounds": (C.c_int, [vp, C.c_int, c_dp, c_dp]),
    "kp_multi_mpc_destroy": (C.c_int, [vp]),
    "kp_multi_mpc_step_batch": (C.c_int, [vp, C.c_int, c_dp, c_dp, c_dp, c_dp, c_ip]),
}

_lib = None


def lib():
    """Loads libkoopman_hip.so (once).  Raises OSError if it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise OSError(f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                          "(make -C koopman-realizations_amd/csrc)")
        l = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(l, name)
            fn.restype, fn.argtypes = res, args
        _lib = l
    return _lib


class KoopmanHipError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"libkoopman_hip error {code}: {msg}")
        self.code = code


def check(rc, ctx=None):
    if rc != KP_OK:
        msg = lib().kp_last_error(ctx)
        raise KoopmanHipError(rc, msg.decode() if msg else "")


def fcol(a, dtype=np.float64):
    """Column-major (MATLAB layout) contiguous f64 copy/view."""
    return np.asfortranarray(np.asarray(a, dtype=dtype))


def dptr(a):
    return None if a is None else a.ctypes.data_as(c_dp)
