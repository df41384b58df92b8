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


def test_cpu_backend_of_the_same_c_abi_reproduces_the_oracle(co):
    """oracle/libkoopman_cpu.so exports the fit path of include/koopman_hip.h on the host (its source includes the header: the
    signatures are the compiler-checked ones) and is driven through the product binding's own argument types: kp_create ->
    kp_basis_create -> kp_snapshots_upload -> kp_fit / kp_fit_gram / kp_lift.  This is what bench.py times as cpu_baseline."""
    from oracle import cpu_abi
    from koopman_realizations_amd import _ffi as F
    hdr = open(os.path.join(ROOT, "include", "koopman_hip.h")).read()
    out = subprocess.run(["nm", "-D", "--defined-only", os.path.join(ROOT, "oracle", "libkoopman_cpu.so")], capture_output=True, text=True).stdout
    exported = {ln.split()[-1] for ln in out.splitlines() if " T kp_" in ln}
    assert exported == set(cpu_abi.EXPORTS) and all(f"{n}(" in hdr and n in F.SIGNATURES for n in exported)
    for mt, deg, nz, m in [("bilinear", 3, 6, 3), ("linear", 2, 4, 2), ("nonlinear", 2, 3, 2)]:
        p = synth_pairs(1200, nz, m, seed=9)
        dic = ko.build_dictionary(mt, nz, m, ["poly"], [deg])
        nv = nz + (m if mt == "nonlinear" else 0)
        fit = cpu_abi.CpuFit(mt, nz, m, ko.poly_exponents(nv, deg)[nv:], p["alpha"], p["beta"], p["u"])
        try:
            Px, Py = ko.px_py(dic, p)
            assert (fit.N, fit.W) == (dic.N, dic.W)
            Kref = ko.koopman_ls(Px, Py)
            assert np.abs(fit.fit() - Kref).max() <= 1e-10 * np.abs(Kref).max()
            G, Cm = fit.gram()
            assert np.abs(G - Px.T @ Px).max() <= 1e-12 * np.abs(G).max() and np.abs(Cm - Px.T @ Py).max() <= 1e-12 * np.abs(G).max()
            assert np.abs(fit.lift(F.LIFT_ROW, p["alpha"][:50], p["u"][:50]) - Px[:50]).max() < 1e-14
            full = fit.lift(F.LIFT_FULL, p["alpha"][:50], p["u"][:50] if mt == "nonlinear" else None)
            assert np.abs(full - Px[:50, :dic.N]).max() < 1e-14
        finally:
            fit.close()
