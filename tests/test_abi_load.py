"""CPU tests: the C-ABI library loads, exports every symbol of include/koopman_hip.h and
fails loudly (no silent fallback) when no HIP device is present."""
import os
import re

import pytest

import __graft_entry__ as ge

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def built():
    ge.build()
    from koopman_realizations_amd import _ffi
    return _ffi


def test_every_header_symbol_is_exported_and_bound(built):
    hdr = open(os.path.join(ROOT, "include", "koopman_hip.h")).read()
    declared = set(re.findall(r"\b(kp_[a-z_A-Z0-9]+)\s*\(", hdr))
    declared -= {"kp_status"}
    assert declared, "no declarations parsed"
    assert declared == set(built.SIGNATURES), declared ^ set(built.SIGNATURES)
    lib = built.lib()
    for name in declared:
        assert getattr(lib, name) is not None


def test_product_path_does_not_import_the_oracle():
    pkg = os.path.join(ROOT, "koopman-realizations_amd")
    pat = re.compile(r"^\s*(from\s+oracle|import\s+oracle|from\s+\.+oracle)|oracle/_ref|koopman_oracle", re.M)
    for dp, _, fns in os.walk(pkg):
        for fn in fns:
            if fn.endswith((".py", ".hip", ".h", ".cpp", ".c")):
                assert not pat.search(open(os.path.join(dp, fn)).read()), (dp, fn)


def test_fails_loudly_without_a_gpu(built):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    import koopman_realizations_amd as kra
    with pytest.raises(kra.KoopmanHipError):
        kra.Context(0)


def test_entry_points_of_the_driver_compile_and_parse_their_arguments():
    """bench.py and __graft_entry__.py are run by the driver, not imported by any other test: a syntax error there would
    only show at round end.  Byte-compile both and let bench.py parse the driver's flags (no GPU work: --help exits first)."""
    import py_compile
    import subprocess
    import sys
    for f in ("bench.py", "__graft_entry__.py"):
        py_compile.compile(os.path.join(ROOT, f), doraise=True)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--help"], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and "--gpus" in r.stdout and "--steps" in r.stdout and "--warmup" in r.stdout
