"""Host-side mirrors of the reference's `Kmpc` (Kmpc.m) and `Ksim` (Ksim.m) classes.

Kmpc keeps the reference's constructor options and method signatures for the QP-based
controllers (model_type 'linear' and 'bilinear' with mpc_type 'linear'); the per-step work —
lift, QP assembly and QP solve — is one kernel launch in libkoopman_hip.so.
Loaded models (sysid_class.loaded): the lifted state is the loaded lift with the current load estimate
traj['what'] (Kmpc.m:347-348, :771-772, :839-840); estimate_load_linear / estimate_load_bilinear (Kmpc.m:1298-1445)
assemble the regression on the host and solve the constrained least squares with the library's QP kernel.
state_bounds (Kmpc.m:300-318) are scaled down like the reference does and handed to kp_mpc_set_state_bounds.
Out of scope: mpc_type 'nonlinear' (fmincon SQP).
"""
from __future__ import annotations

import time

import numpy as np

from .device import Mpc


class Kmpc:
    """Model predictive controller (mirror of classdef Kmpc, Kmpc.m:1)."""

    def __init__(self, sysid_class, **kwargs):
        s = sysid_class
        self.sysid = s
        self.ctx = s.ctx
        self.params = s.params                      # Kmpc.m:44
        self.model = s.model
        self.lift = s.lift
        self.basis = s.basis
        self.model_type = s.model_type              # :51
        self.loaded = s.loaded
        # defaults (:55-72)
        self.horizon = int(np.floor(1.0 / self.params["Ts"]))
        self.input_bounds = None
        self.input_slopeConst = None
        self.input_smoothConst = None
        self.state_bounds = None
        self.cost_running = 0.1
        self.cost_terminal = 100.0
        self.cost_input = 0.0
        self.projmtx = self.model["C"]
        self.mpc_type = "nonlinear" if self.model_type == "nonlinear" else "linear"
        for k, v in kwargs.items():                 # parse_args :107-113
            if not hasattr(self, k):
                raise AttributeError(f"unknown Kmpc property {k}")
            setattr(self, k, v)
        if isinstance(self.input_bounds, (list, tuple, np.ndarray)) and np.size(self.input_bounds) == 0:
            self.input_bounds = None
        if isinstance(self.state_bounds, (list, tuple, np.ndarray)) and np.size(self.state_bounds) == 0:
            self.state_bounds = None
        if self.mpc_type != "linear" or self.model_type == "nonlinear":
            raise NotImplementedError("nonlinear MPC (fmincon SQP) is out of scope (SURVEY section 8)")
        self.projmtx = np.atleast_2d(np.asarray(self.projmtx, dtype=np.float64))
        self.expand_props()                         # :82
        m = self.params["m"]
        sc = self.params["scale"]
        ci = np.asarray(self.cost_input, dtype=np.float64)
        r = np.full(m, float(ci)) if ci.ndim == 0 else ci.reshape(-1)   # eye(m).*cost_input  :201,548
        lo = hi = None
        if self.input_bounds is not None:           # :247,659  scaled-down bounds
            lo = (self.input_bounds[:, 0] - sc["u_offset"]) / sc["u_factor"]
            hi = (self.input_bounds[:, 1] - sc["u_offset"]) / sc["u_factor"]
        slope = None if self.input_slopeConst is None else float(self.input_slopeConst) * float(np.mean(sc["u_factor"]))  # :272,684
        smooth = None if self.input_smoothConst is None else \
            self.params["Ts"] ** 2 * float(self.input_smoothConst) * float(np.mean(sc["u_factor"]))                       # :294,706
        self.dev = Mpc(self.ctx, self.model_type, self.model["A"], self.model["B"], self.horizon, self.projmtx,
                       self.cost_running, self.cost_terminal, r, lo, hi, slope, smooth)
        if self.state_bounds is not None:                                   # :313 state_bounds_sc = scaledown.y(state_bounds')'
            self.dev.set_state_bounds((self.state_bounds[:, 0] - sc["y_offset"]) / sc["y_factor"],
                                      (self.state_bounds[:, 1] - sc["y_offset"]) / sc["y_factor"])

    # ---- Kmpc.m:116-130 -------------------------------------------------------------------
    def expand_props(self):
        m = self.params["m"]
        if self.input_bounds is not None:
            b = np.atleast_2d(np.asarray(self.input_bounds, dtype=np.float64))
            if b.shape[0] != m:
                b = np.kron(np.ones((m, 1)), b)
            self.input_bounds = b
        if self.state_bounds is not None:                                   # :126-129
            n = self.params["n"]
            sb = np.atleast_2d(np.asarray(self.state_bounds, dtype=np.float64))
            if sb.shape[0] != n:
                sb = np.kron(np.ones((n, 1)), sb)
            self.state_bounds = sb

    # ---- Kmpc.m:135-152 --------------------------------------------------------------------
    def _ref_index(self):
        n = self.params["n"]
        return np.nonzero(self.projmtx[:, :n].sum(axis=0))[0]

    def scaledown_ref(self, ref):
        idx = self._ref_index(); sc = self.params["scale"]
        return (np.atleast_2d(ref) - sc["y_offset"][idx]) / sc["y_factor"][idx]

    def scaleup_ref(self, ref_sc):
        idx = self._ref_index(); sc = self.params["scale"]
        return np.atleast_2d(ref_sc) * sc["y_factor"][idx] + sc["y_offset"][idx]

    # ---- per-step entry points ----------------------------------------------------------------
    def _zeta(self, traj):
        _, zeta = self.sysid.get_zeta(traj)
        return zeta[-1]                                               # Kmpc.m:343-344

    def _pad_ref(self, ref):
        """Kmpc.m:354-365."""
        Np = self.horizon
        ref = np.atleast_2d(np.asarray(ref, dtype=np.float64))
        if ref.shape[1] != self.projmtx.shape[0]:
            raise ValueError("Reference trajectory is not the correct dimension")
        if ref.shape[0] > Np + 1:
            ref = ref[:Np + 1]
        elif ref.shape[0] < Np + 1:
            ref = np.vstack([ref, np.tile(ref[-1], (Np + 1 - ref.shape[0], 1))])
        return ref.reshape(-1)

    def _step(self, traj, ref, iters):
        zeta = self._zeta(traj)
        u_prev = np.atleast_2d(traj["u"])[-1]
        if self.loaded:                                                # Kmpc.m:347-348: lift with the load estimate
            z = self.lift.econ_full_loaded(zeta, np.atleast_2d(traj["what"])[-1])
            U, st = self.dev.step(z, u_prev, self._pad_ref(ref), iters)
            return U, z
        U, z, st = self.dev.step_zeta(self.sysid.basis_dev, zeta, u_prev, self._pad_ref(ref), iters)
        return U, z                                                    # U is NaN when the QP failed

    # ---- load estimation (Kmpc.m:1298-1445) ------------------------------------------------------------
    def _lsqlin_load(self, Cl, dl, whatpast, pin_last_zero):
        """The lsqlin call (Kmpc.m:1354, :1442): min ||C x - d||^2, x = [1; w], x_1 = 1, -1 <= w <= 1 and, with a
        previous estimate, |w_i - whatpast_i| <= 0.01.  Strictly convex QP in the free loads, solved by kp_qp_solve."""
        nw = self.params["nw"]
        free = list(range(nw - 1)) if (pin_last_zero and nw >= 1) else list(range(nw))
        what = np.zeros(nw)
        if free:
            Cf = Cl[:, 1:][:, free]
            r = dl - Cl[:, 0]
            rows, rhs = [], []
            for k, i in enumerate(free):
                lo, hi = -1.0, 1.0
                if whatpast is not None:
                    wp = np.atleast_2d(whatpast)[-1]
                    lo, hi = max(lo, wp[i] - 0.01), min(hi, wp[i] + 0.01)
                e = np.zeros(len(free)); e[k] = 1.0
                rows += [e, -e]; rhs += [hi, -lo]
            what[free] = self.ctx.qp_solve(2.0 * Cf.T @ Cf, -2.0 * Cf.T @ r, np.array(rows), np.array(rhs))[0]
        res = Cl @ np.concatenate([[1.0], what]) - dl
        return what, float(res @ res)

    def estimate_load_linear(self, ypast, upast, whatpast=None):
        """Kmpc.m:1298-1356.  The shipped code pins the LAST load to zero through the debugging equality
        Aeq = blkdiag(1, 0, 1) (:1350), which only has the right size for nw = 2; reproduced for nw = 2."""
        p = self.params; N, nz, nw, nd = p["N"], p["nzeta"], p["nw"], p["nd"]
        ypast = np.atleast_2d(ypast); upast = np.atleast_2d(upast)
        if ypast.shape[0] != upast.shape[0]:
            raise ValueError("Input arguments must have the same number of rows")
        _, zp = self.sysid.get_zeta({"y": ypast, "u": upast})
        G = self.lift.econ_full(zp[:-1])                                # psi of every past state, one device call
        CA, CB = self.model["A"][:nz, :], self.model["B"][:nz, :]
        rows = [CA @ np.kron(np.eye(nw + 1), g[:, None]) for g in G]      # :1320-1326
        rhs = [zp[i + 1, :nz] - CB @ upast[nd + i] for i in range(len(G))]
        return self._lsqlin_load(np.vstack(rows), np.concatenate(rhs), whatpast, nw == 2)

    def estimate_load_bilinear(self, ypast, upast, whatpast=None):
        """Kmpc.m:1360-1444."""
        p = self.params; N, nz, nw, m = p["N"], p["nzeta"], p["nw"], p["m"]
        NL = N * (nw + 1)
        ypast = np.atleast_2d(ypast); upast = np.atleast_2d(upast)
        if ypast.shape[0] != upast.shape[0]:
            raise ValueError("Input arguments must have the same number of rows")
        _, zp = self.sysid.get_zeta({"y": ypast, "u": upast})
        G = self.lift.econ_full(zp[:-1])
        A, B = self.model["A"], self.model["B"]
        rows = []
        for i, g in enumerate(G):
            Om = np.kron(np.eye(nw + 1), g[:, None])
            rows.append((A[:nz, :] + sum(upast[i, j] * B[:nz, j * NL:(j + 1) * NL] for j in range(m))) @ Om)   # :1384-1394
        rhs = [zp[i + 1, :nz] for i in range(len(G))]
        return self._lsqlin_load(np.vstack(rows), np.concatenate(rhs), whatpast, False)

    def get_mpcInput(self, traj, ref):
        """Kmpc.m:329-387 (linear model)."""
        return self._step(traj, ref, 1)

    def get_mpcInput_bilinear(self, traj, ref):
        """Kmpc.m:750-814."""
        return self._step(traj, ref, 1)

    def get_mpcInput_bilinear_iter(self, traj, ref, iter=1):
        """Kmpc.m:817-904."""
        return self._step(traj, ref, int(iter))

    # ---- Kmpc.m:403-512 ---------------------------------------------------------------------------
    def run_simulation(self, ref_y, y0=None, u0=None):
        """Closed loop with the identified model as the plant (delays = 0)."""
        p = self.params
        n, m = p["n"], p["m"]
        s = self.sysid
        y0 = np.zeros(n) if y0 is None else np.asarray(y0, dtype=np.float64)
        u0 = np.zeros(m) if u0 is None else np.asarray(u0, dtype=np.float64)
        ref_sc = self.scaledown_ref(ref_y)
        res = {"T": [0.0], "U": [u0], "Y": [y0], "K": [0], "R": [np.atleast_2d(ref_y)[0]], "Z": [], "comp_time": []}
        k = 1
        while k < ref_sc.shape[0]:
            cur = {"y": s.scaledown_y(res["Y"][-1])[None, :], "u": s.scaledown_u(res["U"][-1])[None, :]}
            refhor = ref_sc[k - 1:k + self.horizon]
            t0 = time.perf_counter()
            if self.model_type == "linear":
                U, z = self.get_mpcInput(cur, refhor)
            else:
                U, z = self.get_mpcInput_bilinear_iter(cur, refhor, 1)
            res["comp_time"].append(time.perf_counter() - t0)
            if np.isnan(U).any():
                break
            u_k_sc = s.scaledown_u(res["U"][-1])                    # Kmpc.m:495 one-step input delay
            if self.model_type == "linear":
                z1 = self.model["A"] @ z + self.model["B"] @ u_k_sc
            else:
                z1 = self.model["A"] @ z + self.model["Beta"](z) @ u_k_sc
            res["T"].append(k * p["Ts"]); res["U"].append(s.scaleup_u(U[1])); res["Y"].append(s.scaleup_y(self.model["C"] @ z1))
            res["K"].append(k); res["R"].append(self.scaleup_ref(ref_sc[k - 1])[0]); res["Z"].append(z)
            k += 1
        return {k_: np.array(v) for k_, v in res.items()}


class Ksim:
    """Closed-loop simulator (mirror of classdef Ksim, Ksim.m:1).  `system_class` must offer
    simulate_Ts(x, u, w) -> x+ and get_y(x) -> y, like the reference's Arm class, plus
    params['nx'], params['nu']."""

    def __init__(self, system_class, mpc_class):
        self.sys = system_class
        self.mpc = mpc_class

    def run_trial_mpc(self, ref, x0=None, u0=None):
        """Ksim.m:47-262 (delays = 0, unloaded).  Result fields as in the reference."""
        mpc, s = self.mpc, self.mpc.sysid
        Np = mpc.horizon
        nx, nu = int(self.sys.params["nx"]), int(self.sys.params["nu"])
        x0 = np.zeros(nx) if x0 is None else np.asarray(x0, dtype=np.float64)
        u0 = np.zeros(nu) if u0 is None else np.asarray(u0, dtype=np.float64)
        y0 = np.asarray(self.sys.get_y(x0), dtype=np.float64)
        ref = np.atleast_2d(np.asarray(ref, dtype=np.float64))
        ref_sc = mpc.scaledown_ref(ref)                                       # Ksim.m:113
        res = {"T": [0.0], "U": [u0], "Y": [y0], "K": [0], "R": [ref[0]], "X": [x0], "Z": [], "comp_time": [], "err": []}
        proj = mpc.projmtx[:, :mpc.params["n"]]
        k = 1
        while k < ref_sc.shape[0]:                                            # :147
            cur = {"y": s.scaledown_y(res["Y"][-1])[None, :], "u": s.scaledown_u(res["U"][-1])[None, :]}   # :153-166
            refhor = ref_sc[k - 1:k + Np]                                     # :198-202 (1-based k : k+Np)
            t0 = time.perf_counter()                                          # :205
            if mpc.model_type == "linear":
                U, z = mpc.get_mpcInput(cur, refhor)
            else:
                U, z = mpc.get_mpcInput_bilinear_iter(cur, refhor, 1)         # :210
            comp = time.perf_counter() - t0
            if np.isnan(U).any():                                             # :220-222
                break
            u_kp1 = s.scaleup_u(U[1])                                         # :225-228
            x_kp1 = np.asarray(self.sys.simulate_Ts(res["X"][-1], res["U"][-1], None), dtype=np.float64)   # :239-245
            y_kp1 = np.asarray(self.sys.get_y(x_kp1), dtype=np.float64)
            res["T"].append(k * mpc.params["Ts"]); res["U"].append(u_kp1); res["Y"].append(y_kp1); res["K"].append(k)
            res["R"].append(mpc.scaleup_ref(ref_sc[k - 1])[0]); res["X"].append(x_kp1); res["Z"].append(z)
            res["comp_time"].append(comp)
            res["err"].append(float(np.sqrt(((res["R"][-1] - proj @ y_kp1) ** 2).sum())))   # :258
            k += 1
        return {k_: np.array(v) for k_, v in res.items()}


class ModelPlant:
    """Plant adapter that steps the identified Koopman model itself (the role Arm plays in
    example_control.m; the true arm dynamics are out of scope)."""

    def __init__(self, sysid_class):
        self.s = sysid_class
        self.params = {"nx": sysid_class.params["n"], "nu": sysid_class.params["m"]}

    def get_y(self, x):
        return np.asarray(x, dtype=np.float64)

    def simulate_Ts(self, x, u, w=None):
        s = self.s
        z = s.lift.econ_full(s.scaledown_y(x))
        us = s.scaledown_u(u)
        if s.model_type == "linear":
            z1 = s.model["A"] @ z + s.model["B"] @ us
        else:
            z1 = s.model["A"] @ z + s.model["Beta"](z) @ us
        return s.scaleup_y(s.model["C"] @ z1)
