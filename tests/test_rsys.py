"""Rsys generator: structure of the random vector fields, the step inputs and
the data4sysid layout (Rsys.m:34-216).  The reference draws from MATLAB's global stream, so there is
no bit-level fixture; these tests pin the restated formulas."""
import numpy as np

import koopman_realizations_amd as kra
from koopman_realizations_amd.rsys import Rsys


def test_vector_field_formula_and_reproducibility():
    r = Rsys(4, 3, 2, 2, seed=7)
    r2 = Rsys(4, 3, 2, 2, seed=7)
    assert len(r.systems) == 4
    for s, s2 in zip(r.systems, r2.systems):
        assert (s["coeffs"] == s2["coeffs"]).all() and (s["selectors"] == s2["selectors"]).all()
        assert s["selectors"].shape == (3, 4) and set(np.unique(s["selectors"])) <= {0, 1}
        assert (np.abs(s["coeffs"]) <= 1).all() and abs(s["input_gain"]) <= 2
        x, u = 0.37, -0.81
        # literal Rsys.m:72: prod([x x u u].^selectors(j,:))
        funcs = np.array([x, x, u, u])
        terms = sum(c * np.prod(funcs ** sel) for c, sel in zip(s["coeffs"], s["selectors"]))
        want = np.exp(-x ** 4) * (terms + s["input_gain"] * u) - np.arctan(x)
        assert abs(s["vf_func"](0.0, x, u) - want) < 1e-15
    assert any((a["coeffs"] != b["coeffs"]).any() for a, b in zip(r.systems, Rsys(4, 3, 2, 2, seed=8).systems))


def test_step_inputs_hold_and_zero_tail():
    r = Rsys(1, 2, 2, 2, seed=1)
    tq = np.arange(0, 2.0 + 1e-9, 0.01)             # 201 samples -> switch indices 0,50,100,150,200
    U = r.generate_input_steps(tq, 50)
    assert U.shape == (201,) and np.abs(U).max() <= 1
    for a in (0, 50, 100, 150):
        assert (U[a:a + 50] == U[a]).all()
    assert U[200] == 0.0                            # Rsys.m:146-148 never fills the last segment
    assert len({U[0], U[50], U[100], U[150]}) == 4


def test_simulation_and_data4sysid_layout():
    r = Rsys(2, 3, 2, 2, seed=3)
    data = r.simulate_systems(1.0, 0.01, 3, np.array([[0.5]]))
    assert len(data) == 3 and len(data[0]) == 2
    d = data[1][0]
    assert d["t"].shape == (101,) and d["y"].shape == (101, 1) and d["u"].shape == (101, 1)
    assert d["y"][0, 0] == 0.5 and np.isfinite(d["y"]).all() and np.abs(d["y"]).max() < 3
    # one sample of the trajectory against a fine fixed-step RK4 of the same vector field
    f = r.systems[0]["vf_func"]; x = d["y"][10, 0]; u = d["u"][10, 0]; h = 0.01 / 50
    for _ in range(50):
        k1 = f(0, x, u); k2 = f(0, x + h / 2 * k1, u); k3 = f(0, x + h / 2 * k2, u); k4 = f(0, x + h * k3, u)
        x = x + h / 6 * (k1 + 2 * k2 + 2 * k3 + k4)
    assert abs(x - d["y"][11, 0]) < 1e-5
    sets = Rsys.save_data(data)
    assert len(sets) == 2 and len(sets[0]["train"]) == 2 and len(sets[0]["val"]) == 1
    assert sets[1]["val"][0] is data[2][1]


def test_fast_generator_matches_the_scalar_one():
    """simulate_systems_fast (all trajectories as numpy lanes, per-lane ode45 step control) = simulate_systems to rounding,
    with the same random stream."""
    a = Rsys(3, 5, 3, 2, seed=4); b = Rsys(3, 5, 3, 2, seed=4)
    da = a.simulate_systems(0.6, 0.01, 3, np.array([[0.3]]))
    db = b.simulate_systems_fast(0.6, 0.01, 3, np.array([[0.3]]))
    for j in range(3):
        for i in range(3):
            assert np.array_equal(da[j][i]["u"], db[j][i]["u"])
            assert np.abs(da[j][i]["y"] - db[j][i]["y"]).max() < 1e-13


def test_sweep_stacking_of_data4sysid_structs():
    """The sweep's host gathering: equally shaped trials become one block per quantity; ragged layouts are refused (they
    take the per-system path)."""
    from koopman_realizations_amd import sweep
    r = Rsys(4, 3, 2, 2, seed=2)
    sets = Rsys.save_data(r.simulate_systems_fast(0.5, 0.01, 4, np.zeros((1, 1))))
    Y, U, k, Yv, Uv = sweep._stack_raw(sets)
    assert Y.shape == (4, 3 * 51, 1) and U.shape == (4, 3 * 51, 1) and k == 3 and Yv.shape == (4, 51, 1)
    assert np.array_equal(Y[2, 51:102, 0], sets[2]["train"][1]["y"][:, 0]) and np.array_equal(Uv[3, :, 0], sets[3]["val"][0]["u"][:, 0])
    sets[1]["train"][2] = {k_: v[:40] for k_, v in sets[1]["train"][2].items()}
    assert sweep._stack_raw(sets) is None
    assert sweep._stack_raw([{"train": sets[0]["train"][:2], "val": sets[0]["val"]}, sets[2]]) is None
    # the seam test of Ksysid.m:948 (time restarts exactly at the trial joins): a glitch inside a trial, or trials whose
    # clocks run on, are refused; lists in place of arrays are accepted
    import copy
    sets = Rsys.save_data(r.simulate_systems_fast(0.5, 0.01, 4, np.zeros((1, 1))))
    bad = copy.deepcopy(sets); bad[1]["train"][2]["t"][20] = bad[1]["train"][2]["t"][19]
    assert sweep._stack_raw(bad) is None
    bad = copy.deepcopy(sets); bad[0]["train"][1]["t"] = bad[0]["train"][1]["t"] + 100.0
    assert sweep._stack_raw(bad) is None
    lst = copy.deepcopy(sets); lst[0]["train"][0] = {k_: v.tolist() for k_, v in lst[0]["train"][0].items()}
    Yl = sweep._stack_raw(lst)[0]
    assert np.array_equal(Yl, sweep._stack_raw(sets)[0])
