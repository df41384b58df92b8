"""Host-side mirror of the reference's `Ksysid` class (Ksysid.m) over libkoopman_hip.so.

Same method names, argument meaning and result fields as the MATLAB class so that the
scripts example_sysid.m / evaluate_rand_models.m translate line by line; data are numpy
arrays with MATLAB's shapes (rows = time steps / snapshots).  All heavy arithmetic
(lifting, Gram accumulation, solves, rollouts) runs in HIP kernels through the C ABI;
this file only does what the MATLAB host does around those calls: option parsing,
scaling bookkeeping, snapshot selection, dictionary description.

loaded=True (Ksysid.m:539-626: lifted state [1; w] (x) psi for a load vector w) is expressed with the same device
kernels: [1; u] (x) [1; w] (x) psi is the bilinear row of psi for a pseudo-input built from u and w, so any dictionary
(every obs_type, with or without dim_red) goes through the Kronecker Gram kernels unchanged (`_def_observables_loaded`).
Not supported (KP scope): time_type='continuous'.
"""
from __future__ import annotations

import math

import numpy as np

from . import _ffi as F
from .device import Basis, Context, Snapshots, fit, fit_gram, fit_refine

_default_ctx = None


def default_context() -> Context:
    global _default_ctx
    if _default_ctx is None:
        _default_ctx = Context(0)
    return _default_ctx


# ---- monomial ordering (partitions.m:206-219 as called at Ksysid.m:647) ---------------

def _compositions(total, n):
    """Exponent rows of n variables summing to `total`; last variable is the slowest,
    ascending (the order partitions(total, ones(1,n)) returns)."""
    if n == 1:
        return [[total]]
    rows = []
    for last in range(total + 1):
        for head in _compositions(total - last, n - 1):
            rows.append(head + [last])
    return rows


def poly_exponent_table(nvars, degree):
    """Rows of def_polyLift's `exponents` (Ksysid.m:645-648), degree blocks 1..degree."""
    rows = []
    for d in range(1, degree + 1):
        rows.extend(_compositions(d, nvars))
    return np.array(rows, dtype=np.uint8).reshape(-1, nvars)


class _Lift:
    """obj.lift.* of the reference: callables taking column vectors or row batches."""

    def __init__(self, owner):
        self._o = owner

    @staticmethod
    def _as_rows(v):
        v = np.asarray(v, dtype=np.float64)
        if v.ndim == 1:
            return v.reshape(1, -1), True
        if v.ndim == 2 and v.shape[1] == 1:      # MATLAB column vector
            return v.reshape(1, -1), True
        return v, False

    def _lift(self, what, v):
        b = self._o.basis_dev
        V, vec = self._as_rows(v)
        zeta = V[:, :b.nzeta]
        u = V[:, b.nzeta:b.nzeta + b.m] if b.model_type == "nonlinear" else None
        out = b.lift(what, zeta, u)
        return out[0] if vec else out

    def full(self, v):
        """lift.full (Ksysid.m:533): accepts zeta or [zeta;u]; entries past nvars are ignored
        for linear/bilinear (as the matlabFunction handle does)."""
        return self._lift(F.LIFT_FULL, v)

    def econ_full(self, v):
        """lift.econ_full (Ksysid.m:1615-1618 / 1443-1491)."""
        return self._lift(F.LIFT_ECON, v)

    def econ_full_loaded(self, v, w):
        """lift.econ_full_loaded (Ksysid.m:1606-1612): [psi, w_1 psi, ...] with psi = econ_full(v)."""
        return self._o._lift_loaded(F.LIFT_FULL, v, w, None)

    full_loaded = econ_full_loaded        # the handle the reference keeps after dim_red is the econ one (:1524)

    def econ_full_loaded_input(self, zeta, w, u):
        """lift.econ_full_loaded_input (Ksysid.m:1580-1591), bilinear only: kron(eye(m+1), full_loaded) * [1; u]."""
        return self._o._lift_loaded(F.LIFT_ROW, zeta, w, u)

    full_loaded_input = econ_full_loaded_input

    def econ_full_input(self, zeta, u):
        """lift.econ_full_input (Ksysid.m:1594-1604), bilinear only."""
        b = self._o.basis_dev
        z = np.asarray(zeta, dtype=np.float64); uu = np.asarray(u, dtype=np.float64)
        vec = z.ndim == 1
        out = b.lift(F.LIFT_ROW, z.reshape(1, -1) if vec else z, uu.reshape(1, -1) if vec else uu)
        return out[0] if vec else out


class KoopData(dict):
    """koopData of get_Koopman (Ksysid.m:1084-1091): a dict whose `Px` / `Py` entries are materialised (device lift + the
    transfer of Ns x N doubles each) the first time they are read."""

    def lazy(self, key, make):
        self.__dict__.setdefault("_lazy", {})[key] = make

    def __missing__(self, key):
        make = self.__dict__.get("_lazy", {}).pop(key, None)
        if make is None:
            raise KeyError(key)
        self[key] = val = make()
        return val

    def __contains__(self, key):
        return dict.__contains__(self, key) or key in self.__dict__.get("_lazy", {})

    def get(self, key, default=None):
        return self[key] if key in self else default

    def keys(self):
        return list(dict.keys(self)) + list(self.__dict__.get("_lazy", {}))

    def materialise(self):
        """Evaluate the pending entries now (the lift closures hold the device dictionary: do this before it is closed)."""
        for key in list(self.__dict__.get("_lazy", {})):
            self[key]
        return self

    # every view that a caller who STORES koopData uses (dict(kd), iteration, copy, pickle, savemat) sees Px / Py:
    def __iter__(self):
        self.materialise()
        return dict.__iter__(self)

    def __len__(self):
        return dict.__len__(self) + len(self.__dict__.get("_lazy", {}))

    def items(self):
        self.materialise()
        return dict.items(self)

    def values(self):
        self.materialise()
        return dict.values(self)

    def copy(self):
        return dict(self.items())

    def __reduce__(self):
        return (dict, (dict(self.items()),))


class Ksysid:
    """Koopman-based system identification (mirror of classdef Ksysid, Ksysid.m:1)."""

    def __init__(self, data4sysid, ctx: Context | None = None, snapshot_seed=0, gaussian_centres=None, **kwargs):
        if "train" not in data4sysid or "val" not in data4sysid:          # Ksysid.m:46-48
            raise ValueError("Input must have *train* and *val* fields")
        self.ctx = ctx or default_context()
        data = data4sysid["train"][0]
        y0 = np.asarray(data["y"], dtype=np.float64); u0 = np.asarray(data["u"], dtype=np.float64)
        t0 = np.asarray(data["t"], dtype=np.float64).ravel()
        self.params = {"n": y0.shape[1], "m": u0.shape[1], "Ts": float(np.mean(t0[1:] - t0[:-1]))}  # :57-59
        # defaults (:73-81)
        self.isupdate = False
        self.obs_type = ["poly"]; self.obs_degree = [1]
        self.snapshots = math.inf; self.lasso = 1e6; self.delays = 0
        self.model_type = "linear"; self.loaded = False; self.time_type = "discrete"; self.dim_red = False
        self.ls_refine = 1     # (not a reference property) refinement steps after the normal-equations solve: `\` is a QR solve
        self._host_only = bool(kwargs.pop("_host_only", False))           # sweeps: scaling / pairs only, no device dictionary
        self._pca_host = bool(kwargs.pop("_pca_host", False))             # cross-check: pca by host SVD of the lifted matrix
        for k, v in kwargs.items():                                        # parse_args :147-158
            if not hasattr(self, k):
                raise AttributeError(f"unknown Ksysid property {k}")
            setattr(self, k, v)
        las = np.atleast_1d(np.asarray(self.lasso, dtype=np.float64))
        las = np.where(np.isinf(las), 1e6, las)                            # :155-157
        self.lasso = las if las.size > 1 else float(las[0])
        if self.loaded and "w" not in data:                                # :106-109
            raise ValueError("You have specified a loaded system, but your training data does not have the required load field (w)")
        if self.time_type != "discrete":
            raise NotImplementedError("continuous-time models are out of scope (SURVEY section 8)")
        if self.model_type not in ("linear", "bilinear", "nonlinear"):    # :96-104
            raise ValueError("Invalid model_type chosen. Must be linear, bilinear, or nonlinear.")
        self.liftinput = {"linear": 0, "nonlinear": 1, "bilinear": 2}[self.model_type]
        if isinstance(self.obs_type, str):
            self.obs_type = [self.obs_type]
        self.obs_degree = list(np.atleast_1d(self.obs_degree).astype(int))
        if len(self.obs_type) != len(self.obs_degree):                     # :465-467
            raise ValueError("inputs must be of the same size")
        p = self.params
        p["nd"] = int(self.delays)
        p["nzeta"] = p["n"] * (p["nd"] + 1) + p["m"] * p["nd"]             # :86
        p["nw"] = np.atleast_2d(np.asarray(data["w"], dtype=np.float64)).reshape(len(t0), -1).shape[1] if self.loaded else 0   # :89-93
        self.basis_loaded_dev = None
        self._gauss_centres = gaussian_centres
        self._rng = np.random.default_rng(snapshot_seed)
        self.basis = {}
        self.model = None; self.candidates = None; self.koopData = None
        self._def_observables()                                            # :115
        merged = self.merge_trials(data4sysid["train"])                    # :119
        self.traindata = self.get_scale(merged)                            # :122
        self.valdata = [self.scale_data(v) for v in data4sysid["val"]]    # :123-126
        self.snapshotPairs = self.get_snapshotPairs(self.traindata, self.snapshots)   # :134
        # pca (Ksysid.m:1498): covariance of the lifted snapshots from the device Grams at any width (round 5: dictionaries beyond
        # the LDS-staged kernels - poly-3 on a delayed arm state has 816 functions, fourier on six states 728 - go through lifted
        # panels + TN products, csrc/kp_wide.hip), eigenvectors by kp_sym_eig up to its 1024 columns; beyond that the lift on the
        # device and the SVD of the lifted matrix on the host, as the reference's `pca` does
        wide = self.basis_dev.nfull + self.params["m"] > 1024
        Px = self.lift_snapshots(self.snapshotPairs) if (self.dim_red and (self._pca_host or wide)) else None   # :137-141
        self.get_econ_observables(Px)                                      # :142
        if self.loaded and not self._host_only:
            self._def_observables_loaded()                                 # :112-113 (after dim_red: the econ_* loaded lifts, :1521-1565)
        self.lift = _Lift(self)

    # ---- dictionary ----------------------------------------------------------------
    def _def_observables(self):
        """def_observables (Ksysid.m:455-536): the dictionary as data (exponent tables,
        fourier degree, gaussian centres) instead of symbolic expressions."""
        p = self.params
        nv = p["nzeta"] + (p["m"] if self.model_type == "nonlinear" else 0)   # :475-477
        blocks = []
        gi = 0
        for kind, deg in zip(self.obs_type, self.obs_degree):
            if kind == "poly":
                ex = poly_exponent_table(nv, deg)
                blocks.append(("poly", ex[nv:]))                          # :488
            elif kind == "fourier":
                blocks.append(("fourier", deg))
            elif kind == "gaussian":
                if self._gauss_centres is None:                           # :803 (global RNG in the reference)
                    c = 2.0 * self._rng.random((nv, deg)) - 1.0
                else:
                    c = np.asarray(self._gauss_centres[gi], dtype=np.float64); gi += 1
                blocks.append(("gaussian", c))
            elif kind == "hermite":
                blocks.append(("hermite", poly_exponent_table(nv, deg)))   # :836-844, every row of degree 1..deg
            elif kind == "fourier_sparser":
                blocks.append(("fourier_sparser", poly_exponent_table(2 * nv, deg)))   # :746-750
            else:
                raise ValueError(f"unknown obs_type {kind!r}")
        self._blocks = blocks
        self._nvars = nv
        if self._host_only:
            self.basis_dev = None
            p["N"] = None
            return
        self.basis_dev = Basis(self.ctx, self.model_type, p["nzeta"], p["m"], blocks, None)
        p["N"] = self.basis_dev.nfull                                      # :534
        self.basis["blocks"] = blocks

    def _def_observables_loaded(self):
        """def_observables_loaded (Ksysid.m:539-626) and the econ_* loaded lifts (:1580-1612).  The loaded lifted state
        is [1; w] (x) psi with psi = econ_full (any dictionary, with or without dim_red), the bilinear row
        [1; u] (x) [1; w] (x) psi (:1587-1590) - by associativity the ordinary bilinear row of psi for the pseudo-input
        ([1; u] (x) [1; w]) minus its leading 1, in the reference's column order.  So the device dictionary of a loaded
        system is the unloaded one declared 'bilinear' with a pseudo-input:
          bilinear   zeta,        pseudo-input kron([1;u],[1;w])(2:end)           width N (nw+1)(m+1)
          nonlinear  [zeta; u],   pseudo-input w                                   width N (nw+1)        (:1036)
          linear     zeta,        pseudo-input [w; u]; the reference's row [ [1;w] (x) psi , u ] (:1056-1063) is the
                     column subset `_loaded_sub` of it (u_j = u_j * the constant observable that ends psi)."""
        p = self.params
        nz, nw, m, N = p["nzeta"], p["nw"], p["m"], p["N"]
        pcs = self.basis.get("pcs")
        NL = N * (nw + 1)
        if self.model_type == "bilinear":
            nvar, mp, sub = nz, (m + 1) * (nw + 1) - 1, None
        elif self.model_type == "nonlinear":
            nvar, mp, sub = nz + m, nw, None
        else:
            nvar, mp = nz, nw + m
            sub = np.concatenate([np.arange(NL), NL + N * np.arange(m) + N - 1])
        self.basis_loaded_dev = Basis(self.ctx, "bilinear", nvar, mp, self._blocks, pcs)
        assert self.basis_loaded_dev.N == N
        self._loaded_sub = sub

    def _loaded_args(self, v, w, u):
        """(state rows, pseudo-input rows) of the device dictionary of a loaded system."""
        p = self.params; nw, m = p["nw"], p["m"]
        rows = v.shape[0]
        Wl = np.asarray(w, dtype=np.float64).reshape(rows, -1)
        if self.model_type == "nonlinear":
            return v, Wl
        uu = np.zeros((rows, m)) if u is None else np.asarray(u, dtype=np.float64).reshape(rows, -1)
        if self.model_type == "linear":
            return v, np.hstack([Wl, uu])
        one = np.ones((rows, 1))
        wu = (np.hstack([one, uu])[:, :, None] * np.hstack([one, Wl])[:, None, :]).reshape(rows, -1)
        return v, wu[:, 1:]

    def _lift_loaded(self, what, v, w, u):
        """Rows of the loaded lift in the reference's column order; v = zeta (or [zeta, u] for 'nonlinear').
        LIFT_FULL: [1; w] (x) psi (econ_full_loaded, :1606-1612); LIFT_ROW: the row of Px (:1034-1064)."""
        V = np.asarray(v, dtype=np.float64); vec = V.ndim == 1 or (V.ndim == 2 and V.shape[1] == 1)
        V = V.reshape(1, -1) if vec else V
        x, pu = self._loaded_args(V, w, u)
        out = self.basis_loaded_dev.lift(F.LIFT_ROW, x, pu)
        NL = self.params["N"] * (self.params["nw"] + 1)
        if what != F.LIFT_ROW:
            out = out[:, :NL]
        elif self._loaded_sub is not None:
            out = out[:, self._loaded_sub]
        return out[0] if vec else out

    def _loaded_snapshots(self, sp):
        w = np.asarray(sp["w"], dtype=np.float64).reshape(len(sp["alpha"]), -1)
        if self.model_type == "nonlinear":
            a, b = np.hstack([sp["alpha"], sp["u"]]), np.hstack([sp["beta"], sp["u"]])       # :1036-1037
        else:
            a, b = sp["alpha"], sp["beta"]
        _, pu = self._loaded_args(a, w, sp["u"])
        return Snapshots(self.ctx, a, b, pu), w

    def _loaded_grams(self, snaps):
        """Px'Px, Px'Py of the loaded rows (for 'linear': the sub-matrices of the pseudo-input Grams)."""
        G, Cm = fit_gram(self.ctx, self.basis_loaded_dev, snaps)
        if self._loaded_sub is not None:
            ix = np.ix_(self._loaded_sub, self._loaded_sub)
            G, Cm = np.asfortranarray(G[ix]), np.asfortranarray(Cm[ix])
        return G, Cm

    # ---- data handling ---------------------------------------------------------------
    @staticmethod
    def merge_trials(data):
        """Ksysid.m:380-401."""
        if isinstance(data, (list, tuple)):
            keys = ("t", "y", "u") + (("w",) if all("w" in d for d in data) else ())
            return {k: np.vstack([np.asarray(d[k], dtype=np.float64).reshape(len(np.ravel(d["t"])), -1) for d in data])
                    for k in keys}
        return data

    def get_scale(self, data):
        """Ksysid.m:180-229 (scale factors kept in params.scale)."""
        sc = {}
        out = {"t": np.asarray(data["t"], dtype=np.float64)}
        for k in ("y", "u") + (("w",) if "w" in data else ()):              # w: :246-264 (same affine map)
            v = np.asarray(data[k], dtype=np.float64)
            mn, mx = v.min(axis=0), v.max(axis=0)
            off = (mx + mn) / 2.0
            fac = (mx - mn) / 2.0
            fac = np.where(fac == 0, 1.0, fac)
            sc[k + "_offset"], sc[k + "_factor"] = off, fac
            out[k] = (v - off) / fac
        self.params["scale"] = sc
        return out

    def scaledown_y(self, y):
        s = self.params["scale"]; return (np.asarray(y, dtype=np.float64) - s["y_offset"]) / s["y_factor"]

    def scaledown_u(self, u):
        s = self.params["scale"]; return (np.asarray(u, dtype=np.float64) - s["u_offset"]) / s["u_factor"]

    def scaledown_w(self, w):
        s = self.params["scale"]; return (np.asarray(w, dtype=np.float64) - s["w_offset"]) / s["w_factor"]

    def scaleup_w(self, w):
        s = self.params["scale"]; return np.asarray(w, dtype=np.float64) * s["w_factor"] + s["w_offset"]

    def scaleup_y(self, y):
        s = self.params["scale"]; return np.asarray(y, dtype=np.float64) * s["y_factor"] + s["y_offset"]

    def scaleup_u(self, u):
        s = self.params["scale"]; return np.asarray(u, dtype=np.float64) * s["u_factor"] + s["u_offset"]

    def scale_data(self, data, down=True):
        """Ksysid.m:308-343."""
        fy, fu = (self.scaledown_y, self.scaledown_u) if down else (self.scaleup_y, self.scaleup_u)
        out = {"t": np.asarray(data["t"], dtype=np.float64), "y": fy(data["y"]), "u": fu(data["u"])}
        if "w" in data and "w_factor" in self.params.get("scale", {}):          # :327-330
            out["w"] = (self.scaledown_w if down else self.scaleup_w)(np.asarray(data["w"], dtype=np.float64).reshape(len(out["y"]), -1))
        return out

    def get_zeta(self, data_in):
        """Ksysid.m:868-907.  Returns (data_out, zeta)."""
        nd, n, m = self.params["nd"], self.params["n"], self.params["m"]
        y = np.atleast_2d(np.asarray(data_in["y"], dtype=np.float64))
        u = np.atleast_2d(np.asarray(data_in["u"], dtype=np.float64))
        out = dict(data_in)
        if nd == 0:
            out["zeta"], out["uzeta"] = y, u
        else:
            T = y.shape[0]
            cols = [y[nd:]] + [y[nd - j:T - j] for j in range(1, nd + 1)] + [u[nd - j:T - j] for j in range(1, nd + 1)]
            out["zeta"] = np.hstack(cols)
            out["uzeta"] = u[nd:]
        if "w" in data_in:                                                 # :895-897, :902-904
            out["wzeta"] = np.asarray(data_in["w"], dtype=np.float64).reshape(y.shape[0], -1)[nd:]
        return out, out["zeta"]

    def get_snapshotPairs(self, data, num=math.inf):
        """Ksysid.m:910-984.  The reference draws `num` of the num_max pairs without
        replacement from RandStream('mlfg6331_64'); here a seeded numpy Generator draws them
        (MATLAB's stream is not reproducible outside MATLAB; with snapshots=Inf the draw is a
        permutation, to which the least-squares fit is invariant up to rounding)."""
        if isinstance(data, (list, tuple)):
            data = self.merge_trials(data)
        if "zeta" not in data:
            data, _ = self.get_zeta(data)
        if "snapshots" in data:                                            # :932-938
            s = data["snapshots"]
            # column-major, MATLAB's own layout: what the library uploads without a host-side copy on every get_Koopman
            sp = {"alpha": F.fcol(s["alpha"]), "beta": F.fcol(s["beta"]), "u": F.fcol(s["u"])}
            if "w" in s:
                sp["w"] = F.fcol(s["w"])
            return sp
        nd = self.params["nd"]
        t = np.asarray(data["t"], dtype=np.float64).ravel()
        good = np.nonzero(t[nd:-1] < t[nd + 1:])[0]                        # :941-948
        before, after, u = data["zeta"][:-1][good], data["zeta"][1:][good], data["uzeta"][:-1][good]
        num_max = before.shape[0] - 1                                      # :960
        if num > num_max - 1:                                              # :963-967
            num = num_max
        index = self._rng.permutation(num_max)[:int(num)]                  # :974-975
        sp = {"alpha": F.fcol(before[index]), "beta": F.fcol(after[index]), "u": F.fcol(u[index])}     # column-major (MATLAB's layout)
        if "wzeta" in data:                                                # :953-957, :980-982
            sp["w"] = F.fcol(data["wzeta"][:-1][good][index])
        return sp

    # ---- dimension reduction -----------------------------------------------------------
    def lift_snapshots(self, snapshotPairs):
        """Ksysid.m:1394-1432: lift.full on alpha (with u for 'nonlinear') — on the device."""
        u = snapshotPairs["u"] if self.model_type == "nonlinear" else None
        return self.basis_dev.lift(F.LIFT_FULL, snapshotPairs["alpha"], u)

    def get_econ_observables(self, Px=None):
        """Ksysid.m:1435-1577.  `pca` (Statistics toolbox, :1498) = principal axes of the centred lifted snapshots:
        eigenvectors of their covariance.  The covariance comes from the fused Gram kernel on the full dictionary (its
        constant column carries the column sums, so the Ns x Nfull lifted matrix is not needed) and is diagonalised on
        the device (kp_sym_eig, parallel Jacobi); sign convention of MATLAB's pca (largest-magnitude entry of each
        column positive), `explained` and the 99 % cut (:1501-1504) as in the reference.
        (`Px` given: the host LAPACK SVD of the lifted matrix instead - kept for cross-checks.)"""
        p = self.params
        if not self.dim_red:
            self.basis["pcs"] = None
            return
        if Px is not None:
            Xc = Px - Px.mean(axis=0)
            _, sv, vt = np.linalg.svd(Xc, full_matrices=False)
            coeff, latent = vt.T, sv ** 2
        else:
            sp = self.snapshotPairs
            snaps = Snapshots(self.ctx, sp["alpha"], sp["beta"], sp["u"])
            # full dictionary: Psi'Psi is the leading block of the LINEAR row [psi, u] (the bilinear row psi (x) [1; u] would
            # compute (m + 1)^2 times as much, and be too wide for the Gram kernels from nfull = 140 on)
            pb = Basis(self.ctx, "linear", p["nzeta"], p["m"], self._blocks, None) if self.model_type == "bilinear" else self.basis_dev
            try:
                G, _ = fit_gram(self.ctx, pb, snaps)
            finally:
                snaps.close()
                if pb is not self.basis_dev:
                    pb.close()
            Nf = self.basis_dev.nfull
            Gf = G[:Nf, :Nf]
            cnt = Gf[Nf - 1, Nf - 1]                                     # constant observable: number of snapshots
            sums = Gf[:, Nf - 1]
            cov = (Gf - np.outer(sums, sums) / cnt) / (cnt - 1.0)
            latent, coeff, _ = self.ctx.sym_eig(cov)
            latent = np.maximum(latent, 0.0)
        sign = np.sign(coeff[np.argmax(np.abs(coeff), axis=0), np.arange(coeff.shape[1])])
        sign[sign == 0] = 1.0
        coeff = coeff * sign
        explained = 100.0 * latent / latent.sum()
        num_pcs = 1
        while explained[:num_pcs].sum() < 99:                              # :1501-1504
            num_pcs += 1
        pcs = np.ascontiguousarray(coeff[:, :num_pcs])
        self.basis["pcs"] = pcs
        self.basis_dev.close()
        self.basis_dev = Basis(self.ctx, self.model_type, p["nzeta"], p["m"], self._blocks, pcs)
        p["N"] = self.basis_dev.N                                          # :1512-1516

    # ---- fitting ---------------------------------------------------------------------------
    def _resident_snapshots(self, alpha, beta, u):
        """One device snapshot object per Ksysid, refilled in place by every get_Koopman call (kp_snapshots_update: no device
        allocation per call, staged chunked transfer) - what matlab/KsysidHip.m does with kp_mex('snapshots_resident')."""
        a = np.asarray(alpha); uu = np.asarray(u)
        cur = getattr(self, "_snaps_res", None)
        if cur is not None and cur.handle and cur.nzeta == a.shape[1] and cur.m == uu.shape[1]:
            return cur.update(alpha, beta, u)
        if cur is not None:
            cur.close()
        self._snaps_res = Snapshots(self.ctx, alpha, beta, u)
        return self._snaps_res

    def get_Koopman(self, snapshotPairs, lasso=None, want_PxPy=True):
        """Ksysid.m:987-1092.  The per-row lift loop, Px'Px / Px'Py and the solve run on the
        GPU; Px/Py are only materialised (kp_lift) for the koopData fields the reference
        returns (:1085-1086)."""
        N = self.params["N"]
        if self.loaded:
            return self._get_Koopman_loaded(snapshotPairs, lasso, want_PxPy)
        snaps = self._resident_snapshots(snapshotPairs["alpha"], snapshotPairs["beta"], snapshotPairs["u"])
        try:
            obj_lasso = np.atleast_1d(self.lasso)
            if np.all(obj_lasso >= 1e6):                                   # :1068 tests the PROPERTY
                K = fit(self.ctx, self.basis_dev, snaps, [np.inf])[0]
                rank = self.ctx.last_rank()
                if rank < self.basis_dev.W:                                # MATLAB's `\`: "Warning: Rank deficient, rank = ..."
                    import warnings
                    warnings.warn(f"Rank deficient, rank = {rank} of {self.basis_dev.W}: basic solution returned "
                                  "(use dim_red=True as example_sysid.m does)", RuntimeWarning)
                elif self.ls_refine and self.ctx.last_pivot_ratio() < 1e-5:
                    # K = Px \ Py (:1069) to QR accuracy: the normal equations lose cond(G) eps ~ eps / pivot ratio; above
                    # 1e-5 they are already at 1e-11 and the pass over the lifted data buys nothing
                    K = fit_refine(self.ctx, self.basis_dev, snaps, K, int(self.ls_refine))
            else:                                                          # :994-999: t = lasso * N
                lval = 1e4 if lasso is None else float(lasso)
                if self.model_type == "linear" and self.params["nd"] >= 1:
                    K = self._lasso_with_delay_rows(snaps, lval * N)           # :1139-1164
                else:
                    K = fit(self.ctx, self.basis_dev, snaps, [lval])[0]
        finally:
            pass                                                           # the object stays resident for the next call
        koop = KoopData({"K": K, "u": snapshotPairs["u"], "alpha": snapshotPairs["alpha"]})
        if want_PxPy:
            # koopData.Px / .Py = Px(:, 1:N), Py(:, 1:N) (:1085-1086).  For linear and bilinear rows the first N columns ARE
            # the econ lift psi(x) (rows [psi(x), u] / psi(x) (x) [1; u], :1049-1063), so the device lifts N columns instead
            # of the W-wide row block (4x less over PCIe at the bilinear config); nonlinear rows are psi([x; u]) (W = N).
            # Nothing of this package reads them (get_model takes the Grams, kp_model_project): they are materialised at
            # first access - `koopData["Px"]` - not on every fit.
            sp = snapshotPairs
            if self.model_type == "nonlinear":
                koop.lazy("Px", lambda: self.basis_dev.lift(F.LIFT_ROW, sp["alpha"], sp["u"])[:, :N])
                koop.lazy("Py", lambda: self.basis_dev.lift(F.LIFT_ROW, sp["beta"], sp["u"])[:, :N])
            else:
                koop.lazy("Px", lambda: self.basis_dev.lift(F.LIFT_ECON, sp["alpha"]))
                koop.lazy("Py", lambda: self.basis_dev.lift(F.LIFT_ECON, sp["beta"]))
        return koop

    def _get_Koopman_loaded(self, sp, lasso, want_PxPy):
        """get_Koopman with loads (Ksysid.m:1005-1092): lift, Grams and solve on the device through the pseudo-input
        dictionary (`_def_observables_loaded`); K comes back in the reference's column order."""
        N = self.params["N"]; NL = N * (self.params["nw"] + 1)
        snaps, w = self._loaded_snapshots(sp)
        ls = bool(np.all(np.atleast_1d(self.lasso) >= 1e6))
        lval = 1e4 if lasso is None else float(lasso)
        try:
            if self._loaded_sub is None:
                K = fit(self.ctx, self.basis_loaded_dev, snaps, [np.inf] if ls else [lval])[0]     # t = lasso * N (:996)
            else:
                G, Cm = self._loaded_grams(snaps)
                K = self.ctx.fit_solve(G, Cm) if ls else self.ctx.fit_lasso(G, Cm, lval * N)[0]
            if ls and self.ctx.last_rank() < K.shape[0]:
                import warnings
                warnings.warn(f"Rank deficient, rank = {self.ctx.last_rank()} of {K.shape[0]}: basic solution returned", RuntimeWarning)
        finally:
            snaps.close()
        koop = {"K": K, "u": sp["u"], "alpha": sp["alpha"], "w": w}
        if want_PxPy:                                                      # :1085-1086
            koop["Px"] = self._lift_loaded(F.LIFT_ROW, sp["alpha"], w, sp["u"])[:, :NL] if self.model_type != "nonlinear" else \
                self._lift_loaded(F.LIFT_FULL, np.hstack([sp["alpha"], sp["u"]]), w, None)
            koop["Py"] = self._lift_loaded(F.LIFT_ROW, sp["beta"], w, sp["u"])[:, :NL] if self.model_type != "nonlinear" else \
                self._lift_loaded(F.LIFT_FULL, np.hstack([sp["beta"], sp["u"]]), w, None)
        return koop

    def _lasso_with_delay_rows(self, snaps, t):
        """solve_KoopmanQP of a LINEAR model with delays (Ksysid.m:1139-1164): the columns of K that produce the
        delayed part of zeta are pinned to 0/1 by equality rows.  The objective separates by column, so the pinned
        columns are constants and each pinned 1 uses one unit of the L1 budget: the free columns are the lasso
        with t - #ones, solved on the device from the same Grams."""
        p = self.params; n, m, nd, N = p["n"], p["m"], p["nd"], p["N"]
        Nm, nnd, mnd = N + m, n * nd, m * nd
        ones = []
        for i in range(nnd):                               # :1146-1149 (index - 1 = Nm * column offset + row)
            idx = (Nm + 1) * i; ones.append((idx % Nm, n + idx // Nm))
        for i in range(m):                                 # :1150-1153
            idx = Nm * nnd + N + (Nm + 1) * i; ones.append((idx % Nm, n + idx // Nm))
        for i in range(m * (nd - 1)):                      # :1154-1157
            idx = Nm * (nnd + m) + nnd + (Nm + 1) * i; ones.append((idx % Nm, n + idx // Nm))
        c0, c1 = n, n * (nd + 1) + mnd
        G, C = fit_gram(self.ctx, self.basis_dev, snaps)
        free = [j for j in range(Nm) if not (c0 <= j < c1)]
        K = np.zeros((Nm, Nm), order="F")
        for r, c in ones:
            K[r, c] = 1.0
        K[:, free] = self.ctx.fit_lasso(G, np.asfortranarray(C[:, free]), t - len(ones))[0]
        return K

    def get_model(self, koopData):
        """Ksysid.m:1179-1235 (discrete): A, B, C and the projection M."""
        p = self.params; N, n, m = p["N"], p["n"], p["m"]
        K = koopData["K"]
        if self.loaded:                                                    # :1192-1200: every size is N (nw + 1)
            N = N * (p["nw"] + 1)
            snaps, _ = self._loaded_snapshots(koopData)
            try:
                G, Cm = self._loaded_grams(snaps)
            finally:
                snaps.close()
        else:
            snaps = Snapshots(self.ctx, koopData["alpha"], koopData["beta"], koopData["u"])
            try:
                G, Cm = fit_gram(self.ctx, self.basis_dev, snaps)
            finally:
                snaps.close()
        A, B, M = self.ctx.model_project(K, G, Cm, N, m)
        out = {"A": A, "B": B, "C": np.hstack([np.eye(n), np.zeros((n, N - n))]), "M": M, "params": dict(p), "K": K}
        self.model = out
        return out

    def get_BLmodel(self, koopData):
        """Ksysid.m:1238-1282 (with loads every size is N (nw + 1), :1251-1259)."""
        p = self.params; N, n, m = p["N"] * (p["nw"] + 1), p["n"], p["m"]
        UT = koopData["K"].T
        A = np.asfortranarray(UT[:N, :N]); B = np.asfortranarray(UT[:N, N:])
        out = {"A": A, "B": B, "C": np.hstack([np.eye(n), np.zeros((n, N - n))]), "params": dict(p), "K": koopData["K"],
               "Beta": lambda z: np.stack([B[:, i * N:(i + 1) * N] @ np.ravel(z) for i in range(m)], axis=1)}
        self.model = out
        return out

    def get_NLmodel(self, koopData):
        """Ksysid.m:1298-1341: F(zeta,u) = K(:,1:nzeta)' * basis([zeta;u])."""
        p = self.params
        Kf = np.ascontiguousarray(koopData["K"][:, :p["nzeta"]].T)
        if self.loaded:                                                    # :1320-1327
            ff = lambda zeta, u, w: Kf @ self.lift.econ_full_loaded(np.concatenate([np.ravel(zeta), np.ravel(u)]), np.ravel(w))
        else:
            ff = lambda zeta, u: Kf @ self.lift.econ_full(np.concatenate([np.ravel(zeta), np.ravel(u)]))
        out = {"Kf": Kf, "C": np.eye(p["n"]), "params": dict(p), "K": koopData["K"], "F_func": ff}
        self.model = out
        return out

    def train_models(self, lasso=None):
        """Ksysid.m:1344-1389."""
        if lasso is None:
            lasso = self.lasso
        las = np.atleast_1d(np.asarray(lasso, dtype=np.float64))
        extract = {"nonlinear": self.get_NLmodel, "bilinear": self.get_BLmodel, "linear": self.get_model}[self.model_type]
        if las.size < 2:
            self.koopData = self.get_Koopman(self.snapshotPairs, float(las[0]))
            self.koopData["beta"] = self.snapshotPairs["beta"]
            self.candidates = extract(self.koopData)
            self.candidates["lasso"] = float(las[0])
            self.model = self.candidates
        else:
            self.koopData, self.candidates = [], []
            for lv in las:
                kd = self.get_Koopman(self.snapshotPairs, float(lv))
                kd["beta"] = self.snapshotPairs["beta"]
                self.koopData.append(kd)
                c = extract(kd); c["lasso"] = float(lv)
                self.candidates.append(c)
            self.model = self.candidates[0]
        return self

    # ---- validation -------------------------------------------------------------------------
    def _val_common(self, valdata):
        nd = self.params["nd"]
        y = np.asarray(valdata["y"], dtype=np.float64); u = np.asarray(valdata["u"], dtype=np.float64)
        _, zetareal = self.get_zeta(valdata)
        return np.ravel(valdata["t"])[nd:], y[nd:], u[nd:], zetareal

    def _results(self, t, usim, ysim, yreal):
        res = {"t": t, "sim": {"t": t, "u": usim, "y": ysim}, "real": {"t": t, "u": usim, "y": yreal}}
        res["error"] = self.get_error(res["sim"], res["real"])
        return res

    def _val_loaded(self, model, valdata, kind):
        """Loaded rollouts (Ksysid.m:1657-1671, :1751-1765, :1857-1858).  Every step re-lifts the first N entries
        of the state with the load of that step, znow = kron(eye(nw+1), z(1:N)) [1; w]; only those N entries (and
        y = C z, the first n of them) feed forward, so for a constant load the recursion is an ordinary N-dimensional
        linear / bilinear model with A_eff = sum_i wt_i A(1:N, block i) - rolled out on the device, one launch per
        run of constant load (loads are constant per trial in practice)."""
        p = self.params; N, n, m, nw, nz = p["N"], p["n"], p["m"], p["nw"], p["nzeta"]
        t, yreal, ureal, zetareal = self._val_common(valdata)
        wreal = np.asarray(valdata["w"], dtype=np.float64).reshape(len(np.ravel(valdata["t"])), -1)[p["nd"]:]
        T = yreal.shape[0]
        NL = N * (nw + 1)
        cuts = [0] + [j for j in range(1, T - 1) if np.any(wreal[j] != wreal[j - 1])] + [T - 1]
        if kind == "nonlinear":
            Kf = model["Kf"]
            zs = np.zeros((T, nz)); zs[0] = zetareal[0]
            for a, b in zip(cuts[:-1], cuts[1:]):
                if b > a:                                                     # F = (sum_i wt_i Kf(:, block i)) psi([zeta; u])
                    wt = np.concatenate([[1.0], wreal[a]])
                    Keff = np.ascontiguousarray(sum(wt[i] * Kf[:, i * N:(i + 1) * N] for i in range(nw + 1)))
                    zs[a:b + 1] = self.ctx.rollout_nl(self.basis_dev, Keff, zs[a], ureal[a:b + 1])[:, :nz]
            return self._results(t, ureal, zs[:, :n], yreal)
        A, B = model["A"], model["B"]
        Y = np.zeros((T, N)); Y[0] = self.lift.econ_full(zetareal[0])
        for a, b in zip(cuts[:-1], cuts[1:]):
            if b <= a:
                continue
            wt = np.concatenate([[1.0], wreal[a]])
            Aeff = sum(wt[i] * A[:N, i * N:(i + 1) * N] for i in range(nw + 1))
            if kind == "bilinear":
                Beff = np.hstack([sum(wt[i] * B[:N, j * NL + i * N:j * NL + (i + 1) * N] for i in range(nw + 1)) for j in range(m)])
            else:
                Beff = B[:N, :]
            Y[a:b + 1] = self.ctx.rollout(kind, Aeff, Beff, Y[a], ureal[a:b + 1], N)
        ys = Y[:, :n].copy()
        ys[0] = yreal[0]
        res = self._results(t, ureal, ys, yreal)
        res["sim"]["w"] = res["real"]["w"] = wreal                             # :1707-1710
        return res

    def val_model(self, model, valdata):
        """Ksysid.m:1623-1714: z+ = A z + B u rolled out on the device."""
        if self.loaded:
            return self._val_loaded(model, valdata, "linear")
        t, yreal, ureal, zetareal = self._val_common(valdata)
        z0 = self.lift.econ_full(zetareal[0])
        Y = self.ctx.rollout("linear", model["A"], model["B"], z0, ureal, self.params["n"])
        Y[0] = yreal[0]                                                    # :1654
        return self._results(t, ureal, Y, yreal)

    def val_BLmodel(self, model, valdata):
        """Ksysid.m:1717-1812: z+ = A z + B kron(I,z) u."""
        if self.loaded:
            return self._val_loaded(model, valdata, "bilinear")
        t, yreal, ureal, zetareal = self._val_common(valdata)
        z0 = self.lift.econ_full(zetareal[0])
        Y = self.ctx.rollout("bilinear", model["A"], model["B"], z0, ureal, self.params["n"])
        Y[0] = yreal[0]
        return self._results(t, ureal, Y, yreal)

    def val_NLmodel(self, model, valdata):
        """Ksysid.m:1815-1879: zeta+ = F(zeta,u) (serial; lift per step on the device)."""
        if self.loaded:
            return self._val_loaded(model, valdata, "nonlinear")
        t, yreal, ureal, zetareal = self._val_common(valdata)
        zs = self.ctx.rollout_nl(self.basis_dev, model["Kf"], zetareal[0], ureal)   # one launch for the whole trial
        return self._results(t, ureal, zs[:, :self.params["n"]], yreal)

    def get_error(self, simdata, realdata):
        """Ksysid.m:1882-1898.  Diverged rollouts give Inf / NaN errors silently, as MATLAB does."""
        ys, yr = simdata["y"], realdata["y"]
        T = len(realdata["t"])
        with np.errstate(over="ignore", divide="ignore", invalid="ignore"):
            d = ys - yr
            err = {"abs": np.abs(d)}
            err["mean"] = err["abs"].mean(axis=0)
            err["rmse"] = np.sqrt((d ** 2).sum(axis=0) / T)
            err["nrmse"] = err["rmse"] / np.abs(yr.max(axis=0) - yr.min(axis=0))
            err["euclid"] = np.sqrt((d ** 2).sum(axis=1))
            err["euclid_mean"] = err["euclid"].sum() / T
            du = self.scaleup_y(ys) - self.scaleup_y(yr)
            err["unscaled"] = {"euclid": np.sqrt((du ** 2).sum(axis=1))}
            err["unscaled"]["euclid_mean"] = err["unscaled"]["euclid"].sum() / T
        return err
