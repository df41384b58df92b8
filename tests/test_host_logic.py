"""CPU tests of the host-side mirror's own logic (no device calls): monomial tables and
snapshot selection agree with the oracle's independent restatement."""
import numpy as np

from koopman_realizations_amd.ksysid import poly_exponent_table
from oracle import koopman_oracle as ko


def test_exponent_tables_agree_with_oracle():
    for nv, d in [(6, 2), (6, 3), (9, 3), (1, 13), (2, 4), (3, 5)]:
        assert (poly_exponent_table(nv, d) == ko.poly_exponents(nv, d)).all()


def test_mex_gateway_type_checks_against_the_c_abi():
    """matlab/kp_mex.c (the MATLAB-side binding, not buildable here: no MATLAB) compiles against include/koopman_hip.h
    with a declarations-only mex.h: every call in the gateway matches the exported signatures."""
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run(["gcc", "-fsyntax-only", "-Wall", "-Werror", "-I" + os.path.join(root, "include"),
                        "-I" + os.path.join(root, "tests", "mex_stub"), os.path.join(root, "matlab", "kp_mex.c")],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr


def test_matlab_wrappers_call_only_commands_the_gateway_implements():
    import os
    import re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    impl = set(re.findall(r'!strcmp\(cmd, "(\w+)"\)', open(os.path.join(root, "matlab", "kp_mex.c")).read()))
    used = set()
    for fn in ("KsysidHip.m", "KmpcHip.m", "quadprog_hip.m"):
        used |= set(re.findall(r"kp_mex\(\s*'(\w+)'", open(os.path.join(root, "matlab", fn)).read()))
    assert used and used <= impl, used - impl
