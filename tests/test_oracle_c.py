"""The plain-C restatement of the fit (oracle/koopman_oracle_c.c, the compiled CPU baseline) against the numpy oracle,
which is pinned to the reference's stored artefacts (tests/test_oracle_golden.py)."""
import os
import subprocess

import numpy as np
import pytest

from oracle import koopman_oracle as ko
from conftest import synth_pairs

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def co():
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle")], stdout=subprocess.DEVNULL)
    from oracle import c_oracle
    c_oracle.lib()
    return c_oracle


@pytest.mark.parametrize("mt,deg,nz,m", [("bilinear", 3, 6, 3), ("linear", 3, 4, 2), ("nonlinear", 2, 3, 2), ("bilinear", 2, 2, 1)])
def test_c_lift_and_qr_solve_match_the_numpy_oracle(co, mt, deg, nz, m):
    p = synth_pairs(1500, nz, m, seed=8)
    dic = ko.build_dictionary(mt, nz, m, ["poly"], [deg])
    nv = nz + (m if mt == "nonlinear" else 0)
    exps = ko.poly_exponents(nv, deg)[nv:]
    Px, Py = ko.px_py(dic, p)
    assert np.abs(co.lift_rows(mt, nz, m, exps, p["alpha"], p["u"]) - Px).max() < 1e-14
    K = co.get_koopman(mt, nz, m, exps, p["alpha"], p["beta"], p["u"])
    Kref = ko.koopman_ls(Px, Py)
    assert np.abs(K - Kref).max() <= 1e-10 * np.abs(Kref).max()


def test_c_oracle_on_the_arm_pairs_subset(co, arm):
    """The reference's own data (first 3000 pairs of the arm file, poly-2 on the raw markers is rank deficient, so the
    full-rank linear poly-1 dictionary is used): C QR solve = numpy lstsq."""
    p = {k: v[:3000] for k, v in arm["pairs"].items()}
    dic = ko.build_dictionary("linear", 6, 3, ["poly"], [1])
    Px, Py = ko.px_py(dic, p)
    K = co.get_koopman("linear", 6, 3, np.zeros((0, 6), np.uint8), p["alpha"], p["beta"], p["u"])
    Kref = ko.koopman_ls(Px, Py)
    assert np.abs(K - Kref).max() <= 1e-9 * np.abs(Kref).max()
