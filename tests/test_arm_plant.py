"""Arm plant against the transitions stored by the reference's own runs:
the closed-loop result files hold X(k+1) = Arm.simulate_Ts(X(k), U(k)) (Ksim.m:239-245) and
Y = Arm.get_y(X), so every stored step pins the equations of motion, the integrator and the
marker output at once.  CPU only (the plant is host code)."""
import numpy as np
import pytest

import koopman_realizations_amd as kra


@pytest.fixture(scope="module")
def plant(golden):
    g = golden["arm_plant"]
    params = {k[2:]: (float(g[k]) if g[k].ndim == 0 else g[k]) for k in g.files if k.startswith("p_")}
    return kra.Arm(params, output_type="markers"), g


def test_marker_output_matches_stored_outputs(plant):
    arm, g = plant
    assert np.abs(arm.get_y(g["bilin_X"]) - g["bilin_Y"]).max() < 1e-13
    assert np.abs(arm.get_y(g["train_x"]) - g["train_y"]).max() < 1e-13
    assert arm.get_y(g["bilin_X"][3]).shape == (6,)
    with pytest.raises(ValueError):
        arm.get_y(np.zeros((2, 5)))
    arm2 = kra.Arm(arm.params, output_type="endeff")
    assert np.abs(arm2.get_y(g["bilin_X"]) - g["bilin_Y"][:, -2:]).max() < 1e-13
    assert (kra.Arm(arm.params).get_y(g["bilin_X"]) == g["bilin_X"][:, :3]).all()       # default 'angles'


def test_one_period_step_matches_stored_closed_loop(plant):
    """ode45 restatement + closed-form Lagrangian dynamics reproduce the stored next states."""
    arm, g = plant
    X, U = g["bilin_X"], g["bilin_U"]
    for k in list(range(0, 300, 7)) + [298, 299]:
        x1 = arm.simulate_Ts(X[k], U[k], None)
        assert np.abs(x1 - X[k + 1]).max() < 1e-10, k


def test_open_loop_trial_within_integrator_tolerance(plant):
    """Training trials were integrated across samples (Arm.m:897-898) with the input row of the
    arriving sample held; restarting per sample agrees to the ode45 tolerance on the joint angles."""
    arm, g = plant
    x, u = g["train_x"], g["train_u"]
    for k in range(40, 60):
        x1 = arm.simulate_Ts(x[k], u[k + 1])
        assert np.abs(x1[:3] - x[k + 1, :3]).max() < 2e-4


def test_mass_matrix_and_energy_consistency(plant):
    """Dq is symmetric positive definite; with no damping/spring/input torque the total energy
    (kinetic + potential, Arm.m:153-170) is conserved by the closed-form dynamics, with and without
    an end-effector load."""
    arm, g = plant
    p = dict(arm.params); p.update(d=0.0, ku=0.0, k=0.0)
    free = kra.Arm(p, output_type="markers")
    for w in ((0.0, 0.0), (0.3, 0.4)):
        x = np.array([0.3, -0.5, 0.8, 0.2, -0.1, 0.4])

        def energy(x):
            a, ad = x[:3], x[3:]
            D = free.get_massMatrix(a, w)
            assert np.allclose(D, D.T) and np.linalg.eigvalsh(D).min() > 0
            xj, xcm = free.alpha2x(a)
            grav = np.array([-np.sin(w[1]), np.cos(w[1])])
            return 0.5 * ad @ D @ ad - p["m"] * p["g"] * (xcm @ grav).sum() - w[0] * p["g"] * (xj[-1] @ grav)

        e0 = energy(x)
        from koopman_realizations_amd.arm import dopri45
        x1 = dopri45(lambda t, y: free.vf(y, np.zeros(3), w), 0.0, 0.3, x, rtol=1e-10, atol=1e-12)
        assert abs(energy(x1) - e0) < 1e-7 * max(1.0, abs(e0))
        assert np.abs(x1 - x).max() > 0.05          # it did move


def test_simulate_zoh_and_argument_checks(plant):
    arm, g = plant
    t = np.arange(6) * arm.params["Ts"]
    u = np.tile([0.2, -0.1, 0.3], (6, 1))
    sim = arm.simulate(t, u)
    assert sim["x"].shape == (6, 6) and sim["y"].shape == (6, 6) and (sim["x"][0] == 0).all()
    x = np.zeros(6)
    for k in range(5):
        x = arm.simulate_Ts(x, u[k])
    assert np.abs(x - sim["x"][-1]).max() < 1e-12
    with pytest.raises(ValueError):
        arm.simulate(t, u[:4])
    with pytest.raises(ValueError):
        arm.simulate(t, u[:, :2])
