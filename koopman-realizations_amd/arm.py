"""Planar n-link manipulator plant: the true system that `Ksim` steps in
example_control.m.  Host-side numpy; this is the plant beside the accelerated path, not part of it.

Mirror of classdef Arm (Arm.m:1):
  * equations of motion   Arm.m:111-222 derives them symbolically from the Lagrangian; here the same
                          terms are evaluated in closed form (the chain's Jacobians and Hessians are
                          sums of unit vectors), no symbolic toolbox
  * simulate_Ts           Arm.m:932-957 (ode45 with a state-dependent mass matrix over one period)
  * get_y / get_markers   Arm.m:306-309, 364-413
`ode45` itself is not part of the reference; `dopri45` restates the published algorithm
(Dormand-Prince 5(4) pair with the step control of Shampine & Reichelt, "The MATLAB ODE Suite",
default RelTol 1e-3 / AbsTol 1e-6).
"""
from __future__ import annotations

import numpy as np

# Dormand-Prince 5(4) tableau
_A = np.array([
    [0, 0, 0, 0, 0, 0],
    [1 / 5, 0, 0, 0, 0, 0],
    [3 / 40, 9 / 40, 0, 0, 0, 0],
    [44 / 45, -56 / 15, 32 / 9, 0, 0, 0],
    [19372 / 6561, -25360 / 2187, 64448 / 6561, -212 / 729, 0, 0],
    [9017 / 3168, -355 / 33, 46732 / 5247, 49 / 176, -5103 / 18656, 0],
])
_B5 = np.array([35 / 384, 0, 500 / 1113, 125 / 192, -2187 / 6784, 11 / 84, 0])
_E = np.array([71 / 57600, 0, -71 / 16695, 71 / 1920, -17253 / 339200, 22 / 525, -1 / 40])
_C = np.array([0, 1 / 5, 3 / 10, 4 / 5, 8 / 9, 1, 1])


def dopri45(f, t0, tf, y0, rtol=1e-3, atol=1e-6):
    """Integrate y' = f(t, y) from t0 to tf; returns y(tf).  Step control as ode45's:
    error norm = max |e_i| / max(|y_i|, |ynew_i|, atol/rtol); shrink by max(0.1, 0.8 (rtol/err)^(1/5))
    on the first failure of a step and by 1/2 afterwards; grow by at most 5x after a clean step."""
    y = np.asarray(y0, dtype=np.float64).copy()
    t = float(t0)
    span = tf - t0
    thr = atol / rtol
    hmax = 0.1 * abs(span)
    f0 = f(t, y)
    h = min(hmax, abs(span))
    rh = np.max(np.abs(f0 / np.maximum(np.abs(y), thr))) / (0.8 * rtol ** 0.2)
    if h * rh > 1:
        h = 1.0 / rh
    hmin = 16 * np.finfo(float).eps * max(abs(t), 1e-300)
    h = max(h, hmin)
    k = np.zeros((7, y.size))
    k[0] = f0
    while t < tf:
        hmin = 16 * np.finfo(float).eps * max(abs(t), 1e-300)
        h = min(hmax, max(hmin, h))
        if 1.1 * h >= tf - t:
            h = tf - t
        nofail = True
        while True:
            for s in range(1, 6):
                k[s] = f(t + _C[s] * h, y + h * (_A[s, :s] @ k[:s]))
            ynew = y + h * (_B5[:6] @ k[:6])
            tnew = t + h
            k[6] = f(tnew, ynew)
            err = h * np.max(np.abs(_E @ k) / np.maximum(np.maximum(np.abs(y), np.abs(ynew)), thr))
            if err > rtol:
                if h <= hmin:
                    raise RuntimeError("dopri45: step size underflow")
                if nofail:
                    nofail = False
                    h = max(hmin, h * max(0.1, 0.8 * (rtol / err) ** 0.2))
                else:
                    h = max(hmin, 0.5 * h)
                continue
            break
        if nofail:
            temp = 1.25 * (err / rtol) ** 0.2
            hnext = h / temp if temp > 0.2 else 5.0 * h
        else:
            hnext = h
        t, y = tnew, ynew
        k[0] = k[6]
        h = hnext
    return y


class Arm:
    """params: dict with the fields of the reference's `params` struct (Nmods, nlinks, Nlinks, l, k,
    d, m, i, g, ku, Ts, nx, ny, nu, nw ...), as stored in the data files."""

    def __init__(self, params, output_type="angles"):
        self.params = dict(params)
        for f in ("Nmods", "nlinks", "Nlinks", "nx", "ny", "nu", "nw"):     # counts arrive as doubles from MAT files
            if f in self.params:
                self.params[f] = int(self.params[f])
        self.output_type = output_type
        n = int(self.params["Nlinks"])
        self._Jth = np.tril(np.ones((n, n)))                 # d theta / d alpha  (Arm.m:35-49)
        self._W = self._weights()

    # ---- kinematics -------------------------------------------------------------------------
    @staticmethod
    def _e(th):      # unit vector along a link whose absolute angle (from the y axis) is th  (Arm.m:72-73)
        return np.stack([-np.sin(th), np.cos(th)])

    def alpha2theta(self, alpha):
        return np.cumsum(np.asarray(alpha, dtype=np.float64))

    def alpha2x(self, alpha):
        """Joint coordinates (Nlinks+1 x 2) and link centres of mass (Nlinks x 2), Arm.m:52-83."""
        l = self.params["l"]
        e = self._e(self.alpha2theta(alpha)).T
        x = np.vstack([np.zeros((1, 2)), np.cumsum(l * e, axis=0)])
        return x, x[:-1] + 0.5 * l * e

    def get_markers(self, alpha):
        x, _ = self.alpha2x(alpha)
        return x[::int(self.params["nlinks"])]

    def get_y(self, x):
        """Arm.m:364-413; x: one state or rows of states [alpha, alphadot]."""
        x = np.asarray(x, dtype=np.float64)
        single = x.ndim == 1
        X = np.atleast_2d(x)
        n = int(self.params["Nlinks"])
        if X.shape[1] != 2 * n:
            raise ValueError(f"Input state matrix has wrong dimension. Its width should be {2 * n}")
        if self.output_type == "markers":
            Y = np.stack([self.get_markers(r[:n])[1:].ravel() for r in X])
        elif self.output_type == "angles":
            Y = X[:, :n].copy()
        elif self.output_type == "endeff":
            Y = np.stack([self.get_markers(r[:n])[-1] for r in X])
        else:
            raise ValueError(f"output_type {self.output_type!r} is not supported")
        return Y[0] if single else Y

    # ---- dynamics ---------------------------------------------------------------------------
    def _weights(self):
        """Constant selection tensors: Jx[i,:,k] = sum_j Wx[i,k,j] de_j and
        d Jx[i,:,k] / d alpha_p = -sum_j Vx[i,p,k,j] e_j (same with Wc, Vc for the centres of mass)."""
        n, l = int(self.params["Nlinks"]), self.params["l"]
        Wx = np.zeros((n, n, n)); Wc = np.zeros((n, n, n))
        Vx = np.zeros((n, n, n, n)); Vc = np.zeros((n, n, n, n))
        for i in range(n):
            wx = np.full(i + 1, l); wc = wx.copy(); wc[i] = 0.5 * l      # weight of link j in x_i / xcm_i
            for k in range(i + 1):
                Wx[i, k, k:i + 1] = wx[k:]; Wc[i, k, k:i + 1] = wc[k:]
                for p in range(i + 1):
                    a = max(k, p)
                    Vx[i, p, k, a:i + 1] = wx[a:]; Vc[i, p, k, a:i + 1] = wc[a:]
        return Wx, Wc, Vx, Vc

    def _jac(self, alpha):
        """Jx[i] (2 x n): d x_i / d alpha for joints i = 1..n;  Jc[i]: same for the centres of mass;
        Hx[i][p], Hc[i][p]: their derivatives with respect to alpha_p."""
        th = self.alpha2theta(alpha)
        e = self._e(th)                                       # 2 x n
        de = np.stack([-np.cos(th), -np.sin(th)])             # d e / d theta
        Wx, Wc, Vx, Vc = self._W
        return (np.einsum("ikj,cj->ick", Wx, de), np.einsum("ikj,cj->ick", Wc, de),
                -np.einsum("ipkj,cj->ipck", Vx, e), -np.einsum("ipkj,cj->ipck", Vc, e))

    def _mass(self, Jx, Jc, w):
        p = self.params
        return (p["i"] * self._Jth.T @ self._Jth + w[0] * Jx[-1].T @ Jx[-1]
                + p["m"] * np.einsum("ica,icb->ab", Jc, Jc))

    def _non_inert(self, J, alpha, ad, u, w):
        p = self.params
        Jx, Jc, Hx, Hc = J
        S = p["m"] * np.einsum("ipca,icb->pab", Hc, Jc)
        if w[0] != 0.0:
            S = S + w[0] * np.einsum("pca,cb->pab", Hx[-1], Jx[-1])
        dD = S + S.transpose(0, 2, 1)                         # dD[p] = d Dq / d alpha_p
        dDa = dD @ ad                                         # [p, i] = sum_j dD[p,i,j] ad_j
        grav = np.array([-np.sin(w[1]), np.cos(w[1])])       # Arm.m:164-166
        dPE = p["k"] * alpha - w[0] * p["g"] * (grav @ Jx[-1]) - p["m"] * p["g"] * (grav @ Jc.sum(axis=0))
        damp = p["d"] * ad                                    # :204
        inp = -p["ku"] * (np.repeat(u, int(p["nlinks"])) - alpha)   # :209
        # Dq_dt*ad = sum_p ad_p dD[p] ad ;  dKE/dalpha_p = ad' dD[p] ad / 2
        return ad @ dDa - (0.5 * (dDa @ ad) - dPE) + damp + inp

    def get_massMatrix(self, alpha, w=(0.0, 0.0)):
        """Dq, Arm.m:146-151."""
        Jx, Jc, _, _ = self._jac(alpha)
        return self._mass(Jx, Jc, w)

    def get_nonInert(self, alpha, alphadot, u, w=(0.0, 0.0)):
        """Dq_dt*alphadot - dL/dalpha + damping + input, Arm.m:176-221."""
        alpha = np.asarray(alpha, dtype=np.float64)
        return self._non_inert(self._jac(alpha), alpha, np.asarray(alphadot, dtype=np.float64),
                               np.asarray(u, dtype=np.float64).ravel(), w)

    def vf(self, x, u, w=(0.0, 0.0)):
        """State derivative [alphadot; alphaddot] = mass matrix \\ vf_RHS (Arm.m:256-300)."""
        n = int(self.params["Nlinks"])
        a, ad = x[:n], x[n:]
        J = self._jac(a)
        return np.concatenate([ad, np.linalg.solve(self._mass(J[0], J[1], w), -self._non_inert(J, a, ad, u, w))])

    def simulate_Ts(self, x_k, u_k, w_k=None, tstep=None):
        """Arm.m:932-957: the state one sampling period later under a held input."""
        w = (0.0, 0.0) if w_k is None or len(np.ravel(w_k)) == 0 else tuple(np.ravel(w_k))
        T = self.params["Ts"] if tstep is None else tstep
        u = np.asarray(u_k, dtype=np.float64).ravel()
        return dopri45(lambda t, x: self.vf(x, u, w), 0.0, float(T), np.asarray(x_k, dtype=np.float64).ravel())

    def simulate(self, t_in, u_in, w_in=None):
        """Arm.m:960-1049 with input_type 'zoh': rest initial condition, piecewise constant input,
        one integration restart per sample (the reference integrates across samples with the same
        held input; the restart only changes where the integrator places its steps)."""
        t_in = np.asarray(t_in, dtype=np.float64).ravel()
        u_in = np.atleast_2d(np.asarray(u_in, dtype=np.float64))
        n = int(self.params["Nlinks"])
        if u_in.shape[0] != t_in.size:
            raise ValueError("t_in and u_in vectors need to be the same length")
        if u_in.shape[1] != int(self.params["Nmods"]):
            raise ValueError("u_in width must be the same as the number of modules")
        w_in = np.zeros((t_in.size, 2)) if w_in is None else np.broadcast_to(np.atleast_2d(w_in), (t_in.size, 2))
        X = np.zeros((t_in.size, 2 * n))
        for k in range(t_in.size - 1):
            X[k + 1] = self.simulate_Ts(X[k], u_in[k], w_in[k], t_in[k + 1] - t_in[k])
        return {"t": t_in, "x": X, "alpha": X[:, :n], "alphadot": X[:, n:], "y": self.get_y(X), "u": u_in,
                "w": np.array(w_in), "params": self.params}
