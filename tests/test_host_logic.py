"""CPU tests of the host-side mirror's own logic (no device calls): monomial tables and
snapshot selection agree with the oracle's independent restatement."""
import numpy as np
import pytest

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


def test_host_gather_helper_matches_numpy_and_rejects_what_it_cannot_take():
    """_kp_gather (csrc/kp_pygather.c, built by the csrc Makefile): the buffer-protocol gather behind sweep._stack_raw gives
    what np.concatenate gives, reports unequal piece lengths, refuses non-float64 pieces (the caller then converts through
    numpy) and pieces that overflow the destination; _stack_raw falls back to None on ragged trials either way."""
    import numpy as np
    import __graft_entry__ as ge
    ge.build()
    from koopman_realizations_amd import sweep, _kp_gather
    rng = np.random.default_rng(0)
    pieces = [rng.standard_normal((17, 2)) for _ in range(300)]
    out = np.zeros((300 * 17, 2))
    n, same = _kp_gather.gather(pieces, out.ctypes.data, out.nbytes, 3)
    assert n == out.nbytes and same is True and np.array_equal(out, np.concatenate(pieces, axis=0))
    big = [rng.standard_normal(40000) for _ in range(40)]          # > 8 MB: the threaded path
    outb = np.zeros(40 * 40000)
    assert _kp_gather.gather(big, outb.ctypes.data, outb.nbytes, 4)[0] == outb.nbytes and np.array_equal(outb, np.concatenate(big))
    assert _kp_gather.gather([np.zeros(3), np.zeros(4)], out.ctypes.data, out.nbytes)[1] is False
    # the nested form: systems -> trials -> trial[key], walked by the helper itself
    nested = [[{"y": pieces[3 * i + j], "u": None} for j in range(3)] for i in range(100)]
    out2 = np.zeros_like(out)
    assert _kp_gather.gather(nested, out2.ctypes.data, out2.nbytes, 2, "y") == (out.nbytes, True)
    assert np.array_equal(out2, np.concatenate(pieces, axis=0))
    with pytest.raises(KeyError):
        _kp_gather.gather(nested, out2.ctypes.data, out2.nbytes, 2, "t")
    with pytest.raises(TypeError):
        _kp_gather.gather(nested, out2.ctypes.data, out2.nbytes, 2, "u")                     # None is no buffer
    with pytest.raises(TypeError):
        _kp_gather.gather([[pieces[0]]], out2.ctypes.data, out2.nbytes, 2, "y")              # trials must be dicts
    with pytest.raises(TypeError):
        _kp_gather.gather([np.arange(4)], out.ctypes.data, out.nbytes)                      # int64
    with pytest.raises(ValueError):
        _kp_gather.gather([np.zeros(10)], out.ctypes.data, 8)
    with pytest.raises((TypeError, BufferError, ValueError)):
        _kp_gather.gather([np.zeros((4, 4))[:, ::2]], out.ctypes.data, out.nbytes)          # not contiguous
    tq = np.arange(11) * 0.1
    systems = [{"train": [{"t": tq.copy(), "y": rng.standard_normal((11, 1)), "u": rng.standard_normal((11, 1))} for _ in range(3)],
                "val": [{"t": tq.copy(), "y": rng.standard_normal((11, 1)), "u": rng.standard_normal((11, 1))}]} for _ in range(5)]
    a = sweep._stack_raw(systems)
    helper, sweep._kp_gather = sweep._kp_gather, None
    try:
        b = sweep._stack_raw(systems)
    finally:
        sweep._kp_gather = helper
    assert all(np.array_equal(x, y) if isinstance(x, np.ndarray) else x == y for x, y in zip(a, b))
    systems[2]["train"][1]["y"] = [[float(v)] for v in systems[2]["train"][1]["y"][:, 0]]    # a list among the arrays: numpy path
    c = sweep._stack_raw(systems)
    assert np.array_equal(c[0], a[0])
    systems[1]["train"][0]["y"] = systems[1]["train"][0]["y"][:10]                          # ragged: no stacked form
    assert sweep._stack_raw(systems) is None


def test_seam_check_of_the_stacked_sweep_agrees_with_the_numpy_form():
    """_kp_gather.trials_increasing (the seam test of Ksysid.m:948 asked of all trials at once, without copying the time
    vectors): a clock that runs on across a trial join, a repeated or decreasing time stamp inside a trial and a NaN all
    refuse the stacked form, exactly as the numpy comparison does; the big (threaded) case agrees too."""
    import numpy as np
    import __graft_entry__ as ge
    ge.build()
    from koopman_realizations_amd import sweep, _kp_gather
    rng = np.random.default_rng(1)
    tq = np.arange(11) * 0.1

    def systems(nsys=4):
        return [{"train": [{"t": tq.copy(), "y": rng.standard_normal((11, 1)), "u": rng.standard_normal((11, 1))} for _ in range(3)],
                 "val": [{"t": tq.copy(), "y": rng.standard_normal((11, 1)), "u": rng.standard_normal((11, 1))}]} for _ in range(nsys)]

    def both(sy):
        a = sweep._stack_raw(sy)
        helper, sweep._kp_gather = sweep._kp_gather, None
        try:
            b = sweep._stack_raw(sy)
        finally:
            sweep._kp_gather = helper
        assert (a is None) == (b is None)
        return a

    assert both(systems()) is not None
    s1 = systems(); s1[1]["train"][1]["t"] = tq + 1.1            # clock runs on over the join 0 -> 1: the pair across it would count
    assert both(s1) is None
    s2 = systems(); s2[3]["train"][2]["t"][5] = s2[3]["train"][2]["t"][4]      # repeated stamp inside a trial
    assert both(s2) is None
    s3 = systems(); s3[0]["train"][0]["t"][7] = np.nan
    assert both(s3) is None
    s4 = systems(); s4[2]["train"][2]["t"] = tq + 5.0            # later start of the LAST trial of a system still restarts vs the next system's
    assert both(s4) is None                                       # ... but not vs its own predecessor: 1.0 < 5.0 runs on
    big = [np.arange(3000) * 0.01 for _ in range(600)]            # > 8 MB: threads
    assert _kp_gather.trials_increasing(big, 10, 4) == (True, True)
    big[431] = big[431].copy(); big[431][2999] = big[431][2998]
    assert _kp_gather.trials_increasing(big, 10, 4)[0] is False
    assert _kp_gather.trials_increasing([tq, tq[:5]], 2, 1) == (True, False)
    with pytest.raises(TypeError):
        _kp_gather.trials_increasing([np.arange(4)], 1, 1)
    with pytest.raises(ValueError):
        _kp_gather.trials_increasing([tq, tq, tq], 2, 1)
