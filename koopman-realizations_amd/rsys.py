"""Random 1-D nonlinear systems: the data generator behind the batched sweep
(config 5, evaluate_rand_models.m).  Host-side numpy; mirror of classdef Rsys (Rsys.m:1).

  construct_systems     Rsys.m:34-91    xdot = exp(-x^4) (sum_j c_j x^(a_j) u^(b_j) + c_u u) - atan(x)
                                         a_j / b_j = number of selected copies of x / u (selectors are
                                         0/1 over `degree_x` copies of x and `degree_u` copies of u)
  simulate_systems      Rsys.m:96-125   step inputs held `num_steps` samples, ode45 between samples
  generate_input_steps  Rsys.m:136-150
  save_data             Rsys.m:182-216  -> list of data4sysid dicts {train: [...], val: [...]} (the last
                                         trial validates), the input format of `Ksysid` / `sweep.eval_system`
MATLAB's global random stream is replaced by a seeded numpy Generator (draw order kept: coefficients,
selectors, isolated input gain per system), so systems are reproducible but not the reference's own
draws; the shipped .mat data sets remain the fixtures for parity.
"""
from __future__ import annotations

import numpy as np

from .arm import dopri45


class Rsys:
    def __init__(self, num_sys, num_terms, degree_x, degree_u, seed=0):
        self.num_sys, self.num_terms = int(num_sys), int(num_terms)
        self.degree_x, self.degree_u = int(degree_x), int(degree_u)
        self.rng = np.random.default_rng(seed)
        self.systems = []
        self.construct_systems()

    def construct_systems(self):
        """Rsys.m:60-90."""
        self.systems = []
        for _ in range(self.num_sys):
            coeffs = 2.0 * self.rng.random(self.num_terms) - 1.0                                   # :62
            sel = self.rng.integers(0, 2, size=(self.num_terms, self.degree_x + self.degree_u))   # :65
            cu = 2.0 * (2.0 * self.rng.random() - 1.0)                                             # :84
            px = sel[:, :self.degree_x].sum(axis=1)          # prod(funcs.^selectors): x^(#selected x copies) ...
            pu = sel[:, self.degree_x:].sum(axis=1)          # ... times u^(#selected u copies)   (:72)
            self.systems.append({"coeffs": coeffs, "selectors": sel, "pow_x": px, "pow_u": pu, "input_gain": cu,
                                 "vf_func": self._make_vf(coeffs, px, pu, cu)})
        return self

    @staticmethod
    def _make_vf(coeffs, px, pu, cu):
        def vf(t, x, u):
            x = np.asarray(x, dtype=np.float64); u = np.asarray(u, dtype=np.float64)
            terms = sum(c * x ** a * u ** b for c, a, b in zip(coeffs, px, pu))
            return np.exp(-x ** 4) * (terms + cu * u) - np.arctan(x)        # :82-86
        return vf

    def generate_input_steps(self, tq, num_steps):
        """Rsys.m:136-150: uniform [-1,1] levels held `num_steps` samples.  As in the reference the
        tail after the last switching index stays zero."""
        n = len(tq)
        ind = np.arange(0, n, num_steps)
        inputs = 2.0 * self.rng.random(len(ind)) - 1.0
        U = np.zeros(n)
        for i in range(len(ind) - 1):
            U[ind[i]:ind[i + 1]] = inputs[i]
        return U

    def simulate_systems(self, t_end, Ts, num_trials, x0):
        """Rsys.m:96-125.  Returns data[j][i] = {t, y, u} for trial j of system i.  The state is
        integrated sample to sample under the held input (get_u, :128-133)."""
        x0 = np.atleast_2d(np.asarray(x0, dtype=np.float64))
        if x0.shape[0] == 1:
            x0 = np.repeat(x0, num_trials, axis=0)                       # :104-106
        tq = np.arange(0.0, t_end + 0.5 * Ts, Ts)
        data = [[None] * self.num_sys for _ in range(num_trials)]
        for i, s in enumerate(self.systems):
            f = s["vf_func"]
            for j in range(num_trials):
                uq = self.generate_input_steps(tq, 50)                   # :116
                y = np.zeros((len(tq), 1)); y[0] = x0[j]
                for k in range(len(tq) - 1):
                    uk = uq[k]
                    y[k + 1] = dopri45(lambda t, x: f(t, x, uk), tq[k], tq[k + 1], y[k])
                data[j][i] = {"t": tq.copy(), "y": y, "u": uq[:, None]}
        return data

    @staticmethod
    def save_data(data):
        """Rsys.m:182-216 without the file system: one data4sysid dict per system."""
        out = []
        ntr = len(data)
        for i in range(len(data[0])):
            out.append({"train": [data[j][i] for j in range(ntr - 1)], "val": [data[ntr - 1][i]]})
        return out
