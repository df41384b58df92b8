"""Synthetic loaded system for the `loaded = true` tests: a damped pendulum-like map whose stiffness and input gain
depend on a load w (two loads in the second variant), several trials with a constant load each - the shape of data
the reference's loaded experiments have (fields t, y, u, w per trial)."""
import numpy as np


def _step(x, u, w):
    w2 = w[1] if len(w) > 1 else 0.0
    return np.array([x[0] + 0.1 * x[1] + 0.02 * w2 * u[0],
                     x[1] + 0.1 * (-(1.0 + 0.5 * w[0]) * np.sin(x[0]) - (0.2 + 0.1 * w2) * x[1] + (1.0 - 0.3 * w[0]) * u[0])])


def make_trials(ntrials=8, T=150, nw=1, seed=0):
    rng = np.random.default_rng(seed)
    trials = []
    for _ in range(ntrials):
        w = rng.uniform(-1.0, 1.0, nw)
        x = rng.uniform(-1.0, 1.0, 2)
        ys, us = [], []
        for _k in range(T):
            u = rng.uniform(-1.0, 1.0, 1)
            ys.append(x); us.append(u)
            x = _step(x, u, w)
        trials.append({"t": np.arange(T) * 0.1, "y": np.array(ys), "u": np.array(us), "w": np.tile(w, (T, 1))})
    return trials
