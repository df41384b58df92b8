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

    def simulate_systems_fast(self, t_end, Ts, num_trials, x0):
        """simulate_systems with every (system, trial) integrated at once: the same Dormand-Prince steps and ode45 step
        control as `arm.dopri45`, sample to sample under the held input, carried out on all trajectories as numpy
        lanes with per-lane step sizes (a lane whose step fails retries alone).  Same random stream, same layout;
        the trajectories agree with `simulate_systems` to rounding.  1024 systems x 11 trials x 1001 samples take
        seconds instead of hours, which is what lets the batched sweep run on generated systems."""
        from .arm import _A, _B5, _C, _E
        x0 = np.atleast_2d(np.asarray(x0, dtype=np.float64))
        if x0.shape[0] == 1:
            x0 = np.repeat(x0, num_trials, axis=0)
        tq = np.arange(0.0, t_end + 0.5 * Ts, Ts)
        nt = len(tq)
        ns = self.num_sys
        L = ns * num_trials                                              # lane = system-major, trial-minor (the draw order)
        ind = np.arange(0, nt, 50)
        levels = 2.0 * self.rng.random((ns, num_trials, len(ind))) - 1.0
        U = np.zeros((ns, num_trials, nt))
        for i in range(len(ind) - 1):
            U[:, :, ind[i]:ind[i + 1]] = levels[:, :, i:i + 1]
        U = U.reshape(L, nt)
        co = np.repeat(np.stack([s["coeffs"] for s in self.systems]), num_trials, axis=0)      # L x terms
        px = np.repeat(np.stack([s["pow_x"] for s in self.systems]), num_trials, axis=0).astype(np.float64)
        pu = np.repeat(np.stack([s["pow_u"] for s in self.systems]), num_trials, axis=0).astype(np.float64)
        cu = np.repeat(np.array([s["input_gain"] for s in self.systems]), num_trials)
        Y = np.zeros((L, nt)); Y[:, 0] = np.tile(x0[:, 0], ns)
        pui = pu.astype(np.int64)
        sel_p = [(px == p).astype(np.float64) for p in range(self.degree_x + 1)]
        rtol, atol = 1e-3, 1e-6
        thr = atol / rtol
        eps = np.finfo(float).eps
        span = Ts
        hmax = 0.1 * span
        for k in range(nt - 1):
            uk = U[:, k]
            if k == 0 or not np.array_equal(uk, U[:, k - 1]):              # inputs are held 50 samples
                upow = np.ones((L, self.degree_u + 1))
                for b_ in range(1, self.degree_u + 1):
                    upow[:, b_] = upow[:, b_ - 1] * uk
                cub = co * np.take_along_axis(upow, pui, axis=1)           # c_j u^b_j, constant while the input is held
                # sum_j c_j u^b_j x^a_j collected by power of x: a polynomial per lane (a_j <= degree_x), Horner below
                wp = np.stack([(cub * sel_p[p]).sum(axis=1) for p in range(self.degree_x + 1)])
                wp[0] += cu * uk

            def f(x, ln):
                w = wp[:, ln]
                acc = w[-1]
                for p in range(self.degree_x - 1, -1, -1):
                    acc = acc * x + w[p]
                x2 = x * x
                return np.exp(-x2 * x2) * acc - np.arctan(x)
            allr = np.arange(L)
            t0, tf = tq[k], tq[k + 1]
            y = Y[:, k].copy()
            t = np.full(L, t0)
            k0 = f(y, allr)
            h = np.full(L, min(hmax, span))
            rh = np.abs(k0 / np.maximum(np.abs(y), thr)) / (0.8 * rtol ** 0.2)
            h = np.where(h * rh > 1, 1.0 / np.where(rh > 0, rh, 1.0), h)
            h = np.maximum(h, 16 * eps * max(abs(t0), 1e-300))
            active = t < tf
            while active.any():
                ln = np.nonzero(active)[0]
                hmin = 16 * eps * np.maximum(np.abs(t[ln]), 1e-300)
                hl = np.minimum(hmax, np.maximum(hmin, h[ln]))
                hl = np.where(1.1 * hl >= tf - t[ln], tf - t[ln], hl)
                yl, kl0 = y[ln], k0[ln]
                nofail = np.ones(len(ln), dtype=bool)
                ynew = np.empty(len(ln)); k6 = np.empty(len(ln)); err = np.empty(len(ln))
                todo = np.arange(len(ln))
                while todo.size:
                    sub = ln[todo]
                    ks = np.zeros((7, todo.size)); ks[0] = kl0[todo]
                    hh, yy = hl[todo], yl[todo]
                    for s_ in range(1, 6):
                        ks[s_] = f(yy + hh * (_A[s_, :s_] @ ks[:s_]), sub)
                    yn = yy + hh * (_B5[:6] @ ks[:6])
                    ks[6] = f(yn, sub)
                    er = hh * np.abs(_E @ ks) / np.maximum(np.maximum(np.abs(yy), np.abs(yn)), thr)
                    ynew[todo], k6[todo], err[todo] = yn, ks[6], er
                    bad = er > rtol
                    if bad.any():
                        b = todo[bad]
                        if (hl[b] <= hmin[b]).any():
                            raise RuntimeError("Rsys.simulate_systems_fast: step size underflow")
                        shrink = np.where(nofail[b], hl[b] * np.maximum(0.1, 0.8 * (rtol / err[b]) ** 0.2), 0.5 * hl[b])
                        hl[b] = np.maximum(hmin[b], shrink)
                        nofail[b] = False
                    todo = todo[bad]
                temp = 1.25 * (err / rtol) ** 0.2
                hnext = np.where(nofail, np.where(temp > 0.2, hl / np.where(temp > 0, temp, 1.0), 5.0 * hl), hl)
                t[ln] = t[ln] + hl
                t[ln] = np.where(hl >= tf - (t[ln] - hl), tf, t[ln])       # the closing step lands exactly on the sample
                y[ln] = ynew; k0[ln] = k6; h[ln] = hnext
                active = t < tf
            Y[:, k + 1] = y
        data = [[None] * ns for _ in range(num_trials)]
        for i in range(ns):
            for j in range(num_trials):
                l_ = i * num_trials + j
                data[j][i] = {"t": tq.copy(), "y": Y[l_][:, None].copy(), "u": U[l_][:, None].copy()}
        return data

    @staticmethod
    def save_data(data):
        """Rsys.m:182-216 without the file system: one data4sysid dict per system."""
        out = []
        ntr = len(data)
        for i in range(len(data[0])):
            out.append({"train": [data[j][i] for j in range(ntr - 1)], "val": [data[ntr - 1][i]]})
        return out
