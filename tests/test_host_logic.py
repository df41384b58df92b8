"""CPU tests of the host-side mirror's own logic (no device calls): monomial tables and
snapshot selection agree with the oracle's independent restatement."""
import os

import numpy as np
import pytest

from koopman_realizations_amd.ksysid import poly_exponent_table
from oracle import koopman_oracle as ko


def test_exponent_tables_agree_with_oracle():
    for nv, d in [(6, 2), (6, 3), (9, 3), (1, 13), (2, 4), (3, 5)]:
        assert (poly_exponent_table(nv, d) == ko.poly_exponents(nv, d)).all()


def _matlab_calls():
    """Every kp_mex( 'cmd' , args... ) call in matlab/*.m: (file, line, command, number of arguments after the command, number of
    outputs requested).  A small scanner, not a MATLAB parser: it balances () [] {} and skips strings, which is all these
    files need."""
    import os
    import re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    calls = []
    for fn in sorted(os.listdir(os.path.join(root, "matlab"))):
        if not fn.endswith(".m"):
            continue
        text = open(os.path.join(root, "matlab", fn)).read()
        text = re.sub(r"\.\.\.[^\n]*\n", " ", text)                     # line continuations
        lines = text.split("\n")
        for ln, line in enumerate(lines, 1):
            code = line
            # strip a trailing comment (a % that is not inside a string)
            out, in_str, i = [], False, 0
            while i < len(code):
                ch = code[i]
                if ch == "'" and (in_str or i == 0 or not (code[i - 1].isalnum() or code[i - 1] in ")]}'._")):
                    in_str = not in_str
                elif ch == "%" and not in_str:
                    break
                out.append(ch)
                i += 1
            code = "".join(out)
            for mm in re.finditer(r"kp_mex\(\s*'(\w+)'", code):
                j = mm.end()
                depth, nargs, in_str, k = 1, 0, False, j
                while k < len(code) and depth:
                    ch = code[k]
                    if in_str:
                        in_str = ch != "'"
                    elif ch == "'" and not (code[k - 1].isalnum() or code[k - 1] in ")]}'._"):
                        in_str = True
                    elif ch in "([{":
                        depth += 1
                    elif ch in ")]}":
                        depth -= 1
                    elif ch == "," and depth == 1:
                        nargs += 1
                    k += 1
                assert depth == 0, (fn, ln, line)
                head = code[:mm.start()]
                mo = re.search(r"\[([^\]]*)\]\s*=\s*$", head)
                if mo:
                    nout = len([x for x in re.split(r"[,\s]+", mo.group(1).strip()) if x])
                elif re.search(r"[\w\)\}\.]\s*=\s*$", head):
                    nout = 1
                elif head.strip() == "" and code[k:].strip() in ("", ";"):
                    nout = 0
                else:
                    nout = 1                                              # used inside an expression
                calls.append((fn, ln, mm.group(1), nargs, nout))
    return calls


def test_mex_gateway_builds_against_the_functional_shim_and_reports_its_command_table():
    """matlab/kp_mex.c compiles (gcc -Wall -Wextra -Werror) against include/koopman_hip.h and the functional mex.h of
    tests/mex_shim, links against libkoopman_hip.so with --no-undefined (every entry point it calls exists), and its
    mexFunction RUNS here: the command table comes from the gateway itself, usage errors surface as MATLAB errors."""
    import __graft_entry__ as ge
    ge.build()
    import mexshim as ms
    ms.build(force=True)
    cmds = ms.commands()
    assert len(cmds) >= 70 and cmds["fit"] == (4, 4, 1) and cmds["mpc_step_zeta"] == (5, 6, 2)
    with pytest.raises(ms.MexError) as e:
        ms.kp_mex("no_such_command")
    assert e.value.identifier == "kp:usage"
    with pytest.raises(ms.MexError) as e:
        ms.kp_mex("fit", 1.0)                                  # too few arguments: refused before any handle is touched
    assert e.value.identifier == "kp:usage" and "4 to 4" in e.value.message
    with pytest.raises(ms.MexError) as e:
        ms.kp_mex("destroy", ms.Handle(12345), nargout=1)      # no outputs
    assert e.value.identifier == "kp:usage"
    with pytest.raises(ms.MexError) as e:
        ms.kp_mex("lift", 3.0, 4.0, 1.0, np.zeros((2, 2)))     # handles must be uint64 scalars
    assert e.value.identifier == "kp:handle"
    d = dict(model_type=np.int32(1), nzeta=np.int32(6), m=np.int32(3), block_type=np.array([[0]], dtype=np.int32),
             block_count=np.array([[77]], dtype=np.int32), poly_exps=np.zeros((6, 77), dtype=np.uint8), gauss_centres=None, pcs=None)
    assert ms.kp_mex("basis_desc_dims", d).ravel().tolist() == [6, 84, 84, 336]
    d["poly_exps"] = np.zeros((6, 50), dtype=np.uint8)         # a table shorter than block_count announces must not be read
    with pytest.raises(ms.MexError) as e:
        ms.kp_mex("basis_desc_dims", d)
    assert e.value.identifier == "kp:desc"
    del d["pcs"]
    with pytest.raises(ms.MexError) as e:
        ms.kp_mex("basis_desc_dims", d)
    assert "pcs" in e.value.message
    assert ms.lib().shim_live_arrays() == 0                    # nothing leaked, error paths included
    # every entry point of the C ABI is reachable from MATLAB, except the four that traffic in raw pointers
    import re
    hdr = open(os.path.join(ms.ROOT, "include", "koopman_hip.h")).read()
    declared = set(re.findall(r"\b(kp_[a-z_A-Z0-9]+)\s*\(", hdr)) - {"kp_status"}
    used = set(re.findall(r"\b(kp_[a-z_A-Z0-9]+)\s*\(", re.sub(r"/\*.*?\*/", " ", open(ms.GATEWAY).read(), flags=re.S)))
    assert declared - used == {"kp_host_alloc", "kp_host_free", "kp_stream", "kp_multi_ctx", "kp_multi_host_alloc", "kp_multi_host_free"}, declared - used


def test_matlab_wrappers_call_the_gateway_with_argument_counts_it_accepts():
    """Every kp_mex(...) call of matlab/*.m names a command of the gateway's table and passes a number of arguments and
    requests a number of outputs that the table allows (the table is the one mexFunction enforces at run time)."""
    import mexshim as ms
    cmds = ms.commands()
    calls = _matlab_calls()
    assert len(calls) > 40
    seen = set()
    for fn, ln, cmd, nargs, nout in calls:
        assert cmd in cmds, (fn, ln, cmd)
        lo, hi, nl = cmds[cmd]
        assert lo <= nargs <= hi, (fn, ln, cmd, nargs, (lo, hi))
        assert nout <= max(nl, 0) or (nl >= 1 and nout <= nl), (fn, ln, cmd, nout, nl)
        if nl == 0:
            assert nout == 0, (fn, ln, cmd)
        seen.add(cmd)
    # the wrappers reach the paths the review asked for from MATLAB: refinement, rank, batch MPC, sweeps, several GPUs
    for need in ("fit_refine", "last_pivot_ratio", "last_rank", "mpc_step_batch", "sweep_eval_nested", "traj_upload",
                 "multi_create", "multi_sweep_eval_nested", "fit_lasso", "fit_gram", "model_project", "qp_solve"):
        assert need in seen, need


def test_ksysid_hip_train_models_keeps_the_callers_view_of_the_parent():
    """Static checks of matlab/KsysidHip.m against /root/reference/Ksysid.m:1344-1389, 987-1092 (MATLAB cannot run here):
    the vector branch of train_models leaves koopData a CELL with one struct per value, candidates{i}.lasso, and
    obj.model = obj.candidates{1}; get_Koopman returns [koopData, K] and treats an empty lasso argument as the default."""
    import re
    src = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "matlab", "KsysidHip.m")).read()
    tm = src[src.index("function obj = train_models"):src.index("function delete_hip")]
    assert re.search(r"obj\.koopData\s*=\s*cell\(\s*length\(lasso\)\s*,\s*1\s*\)", tm)
    assert re.search(r"obj\.candidates\s*=\s*cell\(\s*length\(lasso\)\s*,\s*1\s*\)", tm)
    assert re.search(r"obj\.koopData\{i\}\s*=", tm) and re.search(r"obj\.candidates\{i\}\.lasso\s*=\s*lasso\(i\)", tm)
    assert re.search(r"obj\.model\s*=\s*obj\.candidates\{1\}", tm)
    assert "train_models@Ksysid" in tm                         # scalar / loaded / delay-row cases stay the parent's loop
    gk = src[src.index("function [ koopData , K ] = get_Koopman"):src.index("function koopData = hip_koopData")]
    assert "isempty( varargin{1} )" in gk and "lasso = 1e4" in gk
    assert "all( obj.lasso >= 1e6 )" in gk                     # the PROPERTY decides the branch (Ksysid.m:1068)
    assert "fit_refine" in gk and "last_pivot_ratio" in gk and "1e-5" in gk
    assert "hip_lasso_delay_rows" in gk and "obj.params.nd >= 1" in gk
    gm = src[src.index("function [ out , obj ] = get_model"):src.index("function obj = train_models")]
    assert "out.A = MA;" in gm and "out.B = MB;" in gm and "M * " not in gm      # kp_model_project returns M*A, M*B already (executed check:
    kd = src[src.index("function koopData = hip_koopData"):src.index("function K = hip_lasso_delay_rows")]
    order = [kd.index("koopData.K = "), kd.index("koopData.Px = "), kd.index("koopData.Py = "), kd.index("koopData.u = "), kd.index("koopData.alpha = ")]
    assert order == sorted(order)                              # field order of Ksysid.m:1084-1091


def test_integration_excerpt_is_the_shipped_code():
    """INTEGRATION.md section 3 shows get_Koopman of matlab/KsysidHip.m: every line of the excerpt (but the `...` that stands for
    the loaded branch) is a line of the shipped file, in the same order."""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    doc = open(os.path.join(root, "INTEGRATION.md")).read()
    sec = doc[doc.index("## 3. `Ksysid.get_Koopman`"):]
    block = sec[sec.index("```matlab") + len("```matlab"):]
    block = block[:block.index("```")]
    src = [l.strip() for l in open(os.path.join(root, "matlab", "KsysidHip.m")).read().splitlines()]
    pos = 0
    n = 0
    for line in block.splitlines():
        t = line.strip()
        if not t or t == "...":
            continue
        assert t in src[pos:], t
        pos = src.index(t, pos) + 1
        n += 1
    assert n >= 20


def test_matlab_delay_row_indices_equal_the_oracles():
    """KsysidHip.hip_lasso_delay_rows decodes the index formulas of Ksysid.m:1146-1157 with vector expressions; the same
    expressions in numpy give the oracle's literal decoding (oracle.delay_pins) for several (n, m, nd, N)."""
    from oracle import koopman_oracle as ko
    for n, m, nd, N in [(2, 1, 1, 9), (2, 2, 2, 14), (3, 1, 3, 20), (1, 3, 2, 11)]:
        Nm, nnd = N + m, n * nd
        idx = np.concatenate([(Nm + 1) * np.arange(nnd), Nm * nnd + N + (Nm + 1) * np.arange(m),
                              Nm * (nnd + m) + nnd + (Nm + 1) * np.arange(m * (nd - 1))]).astype(int)
        rows, cols = idx % Nm, n + idx // Nm                      # the .m file adds 1 to both (1-based)
        c0, c1, ones = ko.delay_pins(n, m, nd, N)
        assert sorted(zip(rows.tolist(), cols.tolist())) == sorted(ones)
        assert (c0, c1) == (n, n * (nd + 1) + m * nd)             # pinned = n+1 : n*(nd+1)+mnd in the .m file


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
