"""koopman-realizations_amd: MI355X-native Koopman realization fit (Ksysid) and MPC step
(Kmpc) behind the reference's class-method seam.  The compute path is
libkoopman_hip.so (hand-written HIP for gfx950, C ABI in include/koopman_hip.h); this
package is the Python host mirror of the reference's MATLAB classes.

Import name: `koopman_realizations_amd` (shim package at the repo root; the directory
name carries a hyphen to match the reference's repository name).
"""
from . import _ffi
from ._ffi import KoopmanHipError
from .device import Basis, Context, Snapshots, fit, fit_gram, fit_gram_sharded, fit_refine, fit_sharded
from .device import Mpc
from . import comm
from .multi import Multi, MultiMpc
from .arm import Arm
from .kmpc import Kmpc, Ksim, ModelPlant
from .ksysid import Ksysid, default_context, poly_exponent_table

__all__ = ["Arm", "Basis", "Context", "Snapshots", "fit", "fit_gram", "fit_gram_sharded", "fit_refine", "fit_sharded", "Ksysid", "Kmpc", "Ksim", "ModelPlant", "Mpc", "KoopmanHipError", "default_context",
           "poly_exponent_table", "_ffi", "comm", "Multi", "MultiMpc"]
