"""Sanitizer passes over the host-side C of the boundary ("race detection / sanitizers"; CPU only - GPU
AddressSanitizer is not available on this pool).

  * matlab/kp_mex.c (the MATLAB gateway) + tests/mex_shim/mex_shim.c (the functional mex.h stand-in) + oracle/koopman_cpu_abi.c
    (the CPU backend of the same C ABI, fit path) + generated stubs for the entry points the CPU backend does not have, built
    with -fsanitize=address,undefined.  A driver process (libasan / libubsan preloaded into python) runs, through mexFunction:
    every command of the gateway's table with too few / too many arguments and with arguments of the wrong class, the
    handle registry's error paths (stale, foreign, released twice), and a complete CPU-backend fit (create -> basis_create ->
    snapshots_upload -> lift / fit_gram / fit -> destroy) checked against numpy.
  * csrc/kp_pygather.c (the threaded marshalling helper of the Python mirror) built with -fsanitize=thread, gathering and seam-
    testing with 12 threads.

Green = exit code 0 and no sanitizer report on stderr.  The oracle's CPU backend is used here as test infrastructure only."""
import os
import re
import subprocess
import sys
import sysconfig
import tempfile

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HDR = os.path.join(ROOT, "include", "koopman_hip.h")


def _gcc_lib(name):
    p = subprocess.run(["gcc", "-print-file-name=" + name], capture_output=True, text=True).stdout.strip()
    return p if os.path.isabs(p) and os.path.exists(p) else None


def _prototypes():
    """(return type, name, parameter text) of every entry point the header declares."""
    txt = open(HDR).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    txt = re.sub(r"//[^\n]*", "", txt)
    return re.findall(r"^\s*((?:const\s+)?[A-Za-z_][A-Za-z0-9_]*\s*\**)\s*(kp_[A-Za-z0-9_]+)\s*\(([^;{}]*)\)\s*;", txt, flags=re.M)


def _stub_source():
    have = set(re.findall(r"^[A-Za-z_].*?\b(kp_[A-Za-z0-9_]+)\s*\(", open(os.path.join(ROOT, "oracle", "koopman_cpu_abi.c")).read(), flags=re.M))
    out = ['#include <math.h>', '#include "koopman_hip.h"',
           "/* generated: entry points the CPU backend of the C ABI (oracle/koopman_cpu_abi.c) does not implement */", _DESC_DIMS]
    n = 0
    for ret, name, params in _prototypes():
        if name in have or name == "kp_basis_desc_dims":
            continue
        ret = " ".join(ret.split())
        params = " ".join(params.split())
        body = "return 0;" if "*" in ret else "return KP_ERR_HIP;"
        out.append(f"{ret} {name}({params}) {{ {body} }}")
        n += 1
    assert n > 30
    return "\n".join(out) + "\n"


# the one host-only entry point the gateway needs before it can build a dictionary: sizes of a descriptor (csrc/kp_context.hip)
_DESC_DIMS = r'''
int kp_basis_desc_dims(const kp_basis_desc* d, int* nvars_out, int* nfull_out, int* N_out, int* W_out) {
  if (!d || d->nzeta < 1 || d->m < 0 || d->model_type < 0 || d->model_type > 2 || d->n_blocks < 0 || (d->n_blocks && (!d->block_type || !d->block_count)))
    return KP_ERR_ARG;
  const int nvars = d->nzeta + (d->model_type == KP_MODEL_NONLINEAR ? d->m : 0);
  double nf = nvars + 1.0;
  for (int b = 0; b < d->n_blocks; ++b) {
    const int c = d->block_count[b];
    if (c < 0) return KP_ERR_ARG;
    if (d->block_type[b] == KP_BLOCK_FOURIER) nf += pow(2.0 * c + 1.0, (double)nvars) - 1.0;
    else if (d->block_type[b] == KP_BLOCK_POLY || d->block_type[b] == KP_BLOCK_HERMITE || d->block_type[b] == KP_BLOCK_FOURIER_SPARSER ||
             d->block_type[b] == KP_BLOCK_GAUSSIAN) nf += c;
    else return KP_ERR_ARG;
    if (nf > 1e6) return KP_ERR_ARG;
  }
  const int nfull = (int)llround(nf), N = d->k_pcs > 0 ? d->k_pcs + nvars + 1 : nfull;
  if (nvars_out) *nvars_out = nvars;
  if (nfull_out) *nfull_out = nfull;
  if (N_out) *N_out = N;
  if (W_out) *W_out = d->model_type == KP_MODEL_BILINEAR ? N * (d->m + 1) : d->model_type == KP_MODEL_LINEAR ? N + d->m : N;
  return KP_OK;
}
'''

_DRIVER = r'''
import os, sys
import numpy as np
root, so = sys.argv[1], sys.argv[2]
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "tests"))
import mexshim as ms
ms.SHIM_SO = so
ms.build = lambda force=False: so          # the sanitized build, not the GPU one
tab, names = ms.kp_mex("commands", nargout=2)
names = names.split("\n")
assert len(names) == tab.shape[0] and len(names) >= 79
errors = 0
junk = [np.zeros((2, 2)), "text", np.zeros((1, 1), dtype=np.int32), ms.Handle(0xdeadbeef000), None]
for name, (lo, hi, nout) in zip(names, tab.astype(int)):
    if name in ("commands",):
        continue
    trials = []
    if lo > 0:
        trials.append(tuple(junk[i % len(junk)] for i in range(lo - 1)))          # too few
    trials.append(tuple(junk[i % len(junk)] for i in range(hi + 1)))              # too many
    for first in junk:                                                            # right count, wrong classes / foreign handles
        trials.append(tuple([first] + [junk[(i + 1) % len(junk)] for i in range(max(lo, 1) - 1)]))
    for args in trials:
        if name in ("device_count", "last_error", "create", "comm_unique_id") and lo <= len(args) <= hi:
            continue                                                              # (commands without a handle argument that may succeed)
        try:
            ms.kp_mex(name, *args, nargout=max(int(nout), 0))
        except ms.MexError:
            errors += 1
            continue
        except Exception as e:                                                   # the python side of the stand-in refused the value
            errors += 1
            continue
assert errors > 300, errors
# ---- the CPU backend through the gateway: a complete fit --------------------------------------------------------------
from conftest import synth_pairs
from oracle import koopman_oracle as ko
import koopman_realizations_amd as kra
h = ms.kp_mex("create", 0)
p = synth_pairs(400, 3, 2, seed=5)
e = kra.poly_exponent_table(3, 2)[3:].astype(np.uint8)
desc = dict(model_type=np.int32(1), nzeta=np.int32(3), m=np.int32(2), block_type=np.array([[0]], dtype=np.int32),
            block_count=np.array([[e.shape[0]]], dtype=np.int32), poly_exps=np.asfortranarray(e.T), gauss_centres=None, pcs=None)
b = ms.kp_mex("basis_create", h, desc)
dims = ms.kp_mex("basis_dims", b).ravel().tolist()
dic = ko.build_dictionary("bilinear", 3, 2, ["poly"], [2])
assert dims[2:] == [dic.N, dic.W]
s = ms.kp_mex("snapshots_upload", h, p["alpha"], p["beta"], p["u"])
Px, Py = ko.px_py(dic, p)
row = ms.kp_mex("lift", h, b, 2, p["alpha"][:20], p["u"][:20])
assert np.abs(row - Px[:20]).max() < 1e-13
G, Cm = ms.kp_mex("fit_gram", h, b, s, nargout=2)
assert np.abs(G - Px.T @ Px).max() < 1e-10 and np.abs(Cm - Px.T @ Py).max() < 1e-10
K = ms.kp_mex("fit", h, b, s, np.array([[np.inf]]))
K = np.asarray(K).reshape(dic.W, dic.W, order="F") if np.asarray(K).ndim != 2 else K
assert np.abs(K - ko.koopman_ls(Px, Py)).max() < 1e-8
# handle registry: released twice / used after release / never handed out
ms.kp_mex("snapshots_destroy", s, nargout=0)
ms.kp_mex("basis_destroy", b, nargout=0)
for cmd, args in (("basis_destroy", (b,)), ("snapshots_destroy", (s,)), ("basis_dims", (b,)), ("mpc_destroy", (ms.Handle(int(b) + 16),))):
    try:
        ms.kp_mex(cmd, *args, nargout=0)
        raise SystemExit("no error for " + cmd)
    except ms.MexError as ex:
        assert ex.identifier == "kp:handle", ex.identifier
ms.kp_mex("destroy", h, nargout=0)
print("driver ok", errors)
'''


@pytest.mark.timeout(600)
def test_gateway_and_cpu_backend_under_address_and_undefined_behaviour_sanitizers():
    asan, ubsan = _gcc_lib("libasan.so"), _gcc_lib("libubsan.so")
    if not asan or not ubsan:
        pytest.skip("gcc's sanitizer runtimes are not installed")
    with tempfile.TemporaryDirectory() as td:
        stub = os.path.join(td, "kp_stubs.c")
        open(stub, "w").write(_stub_source())
        so = os.path.join(td, "kp_mex_asan.so")
        shim = os.path.join(ROOT, "tests", "mex_shim")
        cmd = ["gcc", "-O1", "-g", "-fno-omit-frame-pointer", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined", "-fopenmp", "-Wall", "-Wextra",
               "-Wno-unused-parameter", "-fPIC", "-shared", "-I", shim, "-I", os.path.join(ROOT, "include"),
               os.path.join(ROOT, "matlab", "kp_mex.c"), os.path.join(shim, "mex_shim.c"), os.path.join(ROOT, "oracle", "koopman_cpu_abi.c"),
               os.path.join(ROOT, "oracle", "koopman_oracle_c.c"), stub, "-o", so, "-lm", "-Wl,--no-undefined"]
        r = subprocess.run(cmd, capture_output=True, text=True)
        assert r.returncode == 0, r.stdout + r.stderr
        env = dict(os.environ, LD_PRELOAD=asan + ":" + ubsan, ASAN_OPTIONS="detect_leaks=0:abort_on_error=0:exitcode=97",
                   UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=1:exitcode=98", OMP_NUM_THREADS="2")
        r = subprocess.run([sys.executable, "-c", _DRIVER, ROOT, so], capture_output=True, text=True, env=env, timeout=550)
        assert r.returncode == 0, (r.returncode, r.stdout[-2000:], r.stderr[-4000:])
        assert "driver ok" in r.stdout
        assert "AddressSanitizer" not in r.stderr and "runtime error:" not in r.stderr, r.stderr[-4000:]


_TSAN_DRIVER = r'''
import sys, importlib.util
import numpy as np
spec = importlib.util.spec_from_file_location("_kp_gather", sys.argv[1])
g = importlib.util.module_from_spec(spec); spec.loader.exec_module(g)
rng = np.random.default_rng(0)
nsys, k, T = 96, 11, 101
systems = [[{"t": np.arange(T) * 0.01, "y": rng.standard_normal((T, 1)), "u": rng.standard_normal((T, 1))} for _ in range(k)] for _ in range(nsys)]
for rep in range(5):
    for key in ("y", "u"):
        out = np.empty((nsys, k * T, 1))
        nbytes, same = g.gather(systems, out.ctypes.data, out.nbytes, 12, key)
        assert nbytes == out.nbytes and same
        ref = np.concatenate([tr[key] for s in systems for tr in s]).reshape(out.shape)
        assert np.array_equal(out, ref)
    ok, same = g.trials_increasing(systems, k, 12, "t")
    assert ok and same
systems[7][3]["t"] = systems[7][3]["t"][::-1].copy()             # a trial whose time runs backwards: the seam test must see it
ok, same = g.trials_increasing(systems, k, 12, "t")
assert not ok
flat = [tr["y"] for s in systems for tr in s]
out = np.empty((len(flat) * T, 1))
nbytes, same = g.gather(flat, out.ctypes.data, out.nbytes, 12)
assert nbytes == out.nbytes and np.array_equal(out, np.concatenate(flat))
print("tsan driver ok")
'''


@pytest.mark.timeout(600)
def test_threaded_gather_helper_under_thread_sanitizer():
    tsan = _gcc_lib("libtsan.so")
    if not tsan:
        pytest.skip("gcc's thread sanitizer runtime is not installed")
    with tempfile.TemporaryDirectory() as td:
        so = os.path.join(td, "_kp_gather" + (sysconfig.get_config_var("EXT_SUFFIX") or ".so"))
        inc = sysconfig.get_paths()["include"]
        cmd = ["gcc", "-O1", "-g", "-fsanitize=thread", "-fPIC", "-shared", "-I", inc, "-DKP_HAVE_NUMPY", "-I", np.get_include(),
               os.path.join(ROOT, "koopman-realizations_amd", "csrc", "kp_pygather.c"), "-o", so, "-lpthread"]
        r = subprocess.run(cmd, capture_output=True, text=True)
        assert r.returncode == 0, r.stdout + r.stderr
        env = dict(os.environ, LD_PRELOAD=tsan, TSAN_OPTIONS="exitcode=96:halt_on_error=0:report_signal_unsafe=0", OMP_NUM_THREADS="1",
                   OPENBLAS_NUM_THREADS="1")
        r = subprocess.run([sys.executable, "-c", _TSAN_DRIVER, so], capture_output=True, text=True, env=env, timeout=550)
        if r.returncode != 0 and "unexpected memory mapping" in r.stderr:
            pytest.skip("ThreadSanitizer cannot map its shadow memory in this container (ASLR layout)")
        assert r.returncode == 0, (r.returncode, r.stdout[-2000:], r.stderr[-4000:])
        assert "tsan driver ok" in r.stdout
        assert "ThreadSanitizer" not in r.stderr, r.stderr[-4000:]
