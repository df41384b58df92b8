"""CPU restatement (numpy, f64) of the Koopman fit + MPC hot path of
roahmlab/koopman-realizations (Ksysid.m / Kmpc.m / partitions.m / Ksim.m).

TEST INFRASTRUCTURE ONLY.  Nothing in the product package may import this module:
only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg use it, and
only as the checker / reported CPU baseline, never as the thing shipped.

Parity pinning (see DESIGN.md "Oracle"):
  * scale -> snapshot pairs -> monomial order -> pca -> econ lift are PINNED by the
    reference's stored closed-loop results (res_bilin.Z / res_lin.Z 300x34,
    res_nonlin.Z 300x88): tests/test_oracle_golden.py reproduces them to ~1e-14.
  * `\\` (mldivide), `quadprog` and `pca` are MathWorks built-ins absent from
    /root/reference (MATLAB R2019a, README.txt:10).  For them the published
    definition is restated: unique least-squares solution of a full-rank system,
    unique optimum of a strictly convex QP, centred economy SVD with MATLAB's sign
    convention.  The fitted-model files are missing from the checkout
    (.MISSING_LARGE_BLOBS:14-16), but the stored closed-loop INPUT sequences
    (res_bilin.U, res_lin.U, 301x3) PIN the chain fit -> get_model/get_BLmodel ->
    cost/constraint assembly -> quadprog end to end: replayed teacher-forced (state
    Y(k), previous input U(k), reference rows k..k+Np; example_control.m settings with
    input_bounds = [] - the stored inputs leave the +-7pi/8 box) this oracle returns
    the stored U(k+1) with median deviation 3.1e-7 (bilinear) / 3.2e-8 (linear) over
    all 299 steps, and on every step MATLAB's input is an optimum of the restated QP
    to 3e-8 relative cost (tests/test_oracle_golden.py::test_stored_matlab_*).
    The lasso `quadprog` (Ksysid.m:1169) has no stored artefact: "parity unpinned by
    artefacts" for that row only (pinned by KKT residuals of the convex problem).

Every function cites the reference file:line it follows.  Layout conventions are
MATLAB's: rows = snapshots / time steps, matrices are what the .m file builds.
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field

import numpy as np

# ----------------------------------------------------------------------------------
# partitions.m:206-219 (recursion) restricted to candidate_set = ones(1,n), as called
# from Ksysid.m:647.
# ----------------------------------------------------------------------------------

def partitions_ones(total: int, n: int) -> np.ndarray:
    """partitions(total, ones(1,n)): all exponent rows of n variables summing to
    `total`, in the reference's order (partitions.m:213-219: the LAST variable's
    count is the outer loop, ascending, recursion on the first n-1)."""
    if total == 0:                       # partitions.m:168-170
        return np.zeros((1, n), dtype=np.int64)
    if n == 0:                           # partitions.m:171-173
        return np.zeros((0, 0), dtype=np.int64)
    if n == 1:                           # partitions.m:174-183
        return np.array([[total]], dtype=np.int64)
    rows = []
    for i in range(0, total + 1):        # partitions.m:214
        sub = partitions_ones(total - i, n - 1)
        rows.append(np.hstack([sub, np.full((sub.shape[0], 1), i, dtype=np.int64)]))
    return np.vstack(rows)


def poly_exponents(nvars: int, degree: int) -> np.ndarray:
    """Ksysid.m:645-648: exponents for total degree 1..degree stacked; N-1 rows
    (constant excluded, Ksysid.m:641,652)."""
    blocks = [partitions_ones(d, nvars) for d in range(1, degree + 1)]
    return np.vstack(blocks) if blocks else np.zeros((0, nvars), dtype=np.int64)


# ----------------------------------------------------------------------------------
# Dictionary (Ksysid.m:455-536) carried as data
# ----------------------------------------------------------------------------------

@dataclass
class Basis:
    """fullBasis of Ksysid.m:484-505 over `nvars` variables (zeta, or [zeta;u] for
    the 'nonlinear' model type, Ksysid.m:475-477)."""
    nvars: int
    blocks: list = field(default_factory=list)   # ('poly', exps) | ('fourier', deg) | ('gaussian', centres) |
                                                 # ('hermite', orders) | ('fourier_sparser', multipliers)

    @property
    def nfull(self) -> int:
        n = self.nvars
        for kind, arg in self.blocks:
            if kind == 'poly':
                n += arg.shape[0] - self.nvars      # Ksysid.m:488 (skip first nvars repeats)
            elif kind == 'fourier':
                n += (1 + 2 * arg) ** self.nvars - 1  # Ksysid.m:718-724
            elif kind == 'gaussian':
                n += arg.shape[1]
            elif kind in ('hermite', 'fourier_sparser'):
                n += arg.shape[0]                   # Ksysid.m:497,503: whole block appended
        return n + 1                                 # Ksysid.m:505 constant at the end


def make_basis(nvars, obs_type, obs_degree, gaussian_centres=None) -> Basis:
    """def_observables, Ksysid.m:484-505.  Gaussian centres are an INPUT (the
    reference draws them from the unseeded global RNG, Ksysid.m:803)."""
    b = Basis(nvars)
    gi = 0
    for kind, deg in zip(obs_type, obs_degree):
        if kind == 'poly':
            b.blocks.append(('poly', poly_exponents(nvars, int(deg))))
        elif kind == 'fourier':
            b.blocks.append(('fourier', int(deg)))
        elif kind == 'gaussian':
            c = np.asarray(gaussian_centres[gi], dtype=np.float64)
            gi += 1
            assert c.shape == (nvars, int(deg))      # Ksysid.m:803 columns are centres
            b.blocks.append(('gaussian', c))
        elif kind == 'hermite':                       # Ksysid.m:836-844: ALL rows of degree 1..deg
            b.blocks.append(('hermite', poly_exponents(nvars, int(deg))))
        elif kind == 'fourier_sparser':               # Ksysid.m:746-750 over 2*nvars multipliers
            b.blocks.append(('fourier_sparser', poly_exponents(2 * nvars, int(deg))))
        else:
            raise ValueError(kind)
    return b


def hermite_h(n, x):
    """MATLAB hermiteH(n, x): physicists' Hermite polynomial, H0 = 1, H1 = 2x,
    H_{k+1} = 2x H_k - 2k H_{k-1}."""
    h0, h1 = np.ones_like(x), 2.0 * x
    if n == 0:
        return h0
    for k in range(1, n):
        h0, h1 = h1, 2.0 * x * h1 - 2.0 * k * h0
    return h1


def lift_full(basis: Basis, V: np.ndarray) -> np.ndarray:
    """lift.full (Ksysid.m:533) evaluated on rows of V (rows = points, cols = the
    basis variables).  Returns rows x Nfull."""
    V = np.atleast_2d(np.asarray(V, dtype=np.float64))
    nv = basis.nvars
    assert V.shape[1] == nv
    cols = [V]                                        # Ksysid.m:484
    for kind, arg in basis.blocks:
        if kind == 'poly':                            # Ksysid.m:652-654,680-691
            ex = arg[nv:]                             # Ksysid.m:488
            out = np.ones((V.shape[0], ex.shape[0]))
            for j in range(nv):
                out *= V[:, [j]] ** ex[:, j][None, :]
            cols.append(out)
        elif kind == 'fourier':                       # Ksysid.m:708-724
            d = arg
            fb = None
            for i in range(nv):
                col = [np.ones(V.shape[0])]
                for j in range(1, d + 1):
                    col.append(np.cos(2 * np.pi * j * V[:, i]))
                    col.append(np.sin(2 * np.pi * j * V[:, i]))
                col = np.stack(col, axis=1)           # rows x (1+2d)
                if fb is None:
                    fb = col
                else:                                 # kron(fourierBasis, poop(:,i)) Ksysid.m:720
                    fb = (fb[:, :, None] * col[:, None, :]).reshape(V.shape[0], -1)
            cols.append(fb[:, 1:])                    # Ksysid.m:724
        elif kind == 'gaussian':                      # Ksysid.m:804-806
            c = arg
            r2 = ((V[:, :, None] - c[None, :, :]) ** 2).sum(axis=1)
            cols.append(np.exp(-r2))
        elif kind == 'hermite':                       # get_hermite, Ksysid.m:806-817
            out = np.ones((V.shape[0], arg.shape[0]))
            for r, orders in enumerate(arg):
                for j in range(nv):
                    out[:, r] *= hermite_h(int(orders[j]), V[:, j])
            cols.append(out)
        elif kind == 'fourier_sparser':               # get_sinusoid, Ksysid.m:768-786
            out = np.ones((V.shape[0], arg.shape[0]))
            for r, mult in enumerate(arg):
                for j in range(nv):
                    if mult[j]:
                        out[:, r] *= np.sin(2 * np.pi * mult[j] * V[:, j])
                    if mult[nv + j]:
                        out[:, r] *= np.cos(2 * np.pi * mult[nv + j] * V[:, j])
            cols.append(out)
    cols.append(np.ones((V.shape[0], 1)))             # Ksysid.m:505
    return np.hstack(cols)


# ----------------------------------------------------------------------------------
# Data handling: scaling, zeta, snapshot pairs
# ----------------------------------------------------------------------------------

def merge_trials(trials):
    """Ksysid.m:380-401 (numeric fields t,y,u only)."""
    return {k: np.vstack([np.asarray(tr[k], dtype=np.float64).reshape(len(tr['t']), -1) for tr in trials])
            for k in ('t', 'y', 'u')}


def get_scale(data):
    """Ksysid.m:187-229. Returns (scaled data, scale dict)."""
    sc = {}
    out = {'t': data['t']}
    for k in ('y', 'u'):
        mn, mx = data[k].min(axis=0), data[k].max(axis=0)
        dc = (mx + mn) / 2.0                          # :193-194
        fac = (mx - mn) / 2.0                         # :197,201
        fac = np.where(fac == 0, 1.0, fac)            # :198-204
        sc[k + '_offset'], sc[k + '_factor'] = dc, fac
        out[k] = (data[k] - dc) / fac                 # :209-210
    return out, sc


def scaledown(sc, k, v):
    return (np.asarray(v, dtype=np.float64) - sc[k + '_offset']) / sc[k + '_factor']


def scaleup(sc, k, v):
    return np.asarray(v, dtype=np.float64) * sc[k + '_factor'] + sc[k + '_offset']


def get_zeta(y, u, nd):
    """Ksysid.m:868-907: zeta_k = [y_k, y_{k-1}..y_{k-nd}, u_{k-1}..u_{k-nd}], uzeta = u_k."""
    y = np.atleast_2d(y); u = np.atleast_2d(u)
    if nd == 0:
        return y.copy(), u.copy()
    rows = []
    for i in range(nd, y.shape[0]):                   # :877
        ydel = [y[i - j] for j in range(1, nd + 1)]
        udel = [u[i - j] for j in range(1, nd + 1)]
        rows.append(np.concatenate([y[i]] + ydel + udel))
    return np.array(rows), u[nd:].copy()


def snapshot_pairs(data, nd, index=None):
    """Ksysid.m:941-978 with the random draw (RandStream('mlfg6331_64') + datasample,
    :974-975, not reproducible) replaced by an explicit 0-based `index` into the
    1..num_max good pairs (default: all of them in order)."""
    zeta, uz = get_zeta(data['y'], data['u'], nd)
    t = np.asarray(data['t']).ravel()
    bt, at = t[nd:-1], t[nd + 1:]                     # :941,943
    good = np.nonzero(bt < at)[0]                     # :948
    before, after, u = zeta[:-1][good], zeta[1:][good], uz[:-1][good]
    num_max = before.shape[0] - 1                     # :960
    if index is None:
        index = np.arange(num_max)
    index = np.asarray(index)
    assert index.max() < num_max
    return {'alpha': before[index], 'beta': after[index], 'u': u[index]}


# ----------------------------------------------------------------------------------
# PCA (Statistics Toolbox `pca`, Ksysid.m:1498) and the econ lift
# ----------------------------------------------------------------------------------

def pca(X):
    """MATLAB pca defaults: centre columns, economy SVD, coeff column sign such that
    the largest-magnitude entry is positive; explained = 100*latent/sum(latent)."""
    Xc = X - X.mean(axis=0)
    _, s, vt = np.linalg.svd(Xc, full_matrices=False)
    coeff = vt.T
    idx = np.argmax(np.abs(coeff), axis=0)
    sign = np.sign(coeff[idx, np.arange(coeff.shape[1])])
    sign[sign == 0] = 1.0
    coeff = coeff * sign
    latent = s ** 2 / (X.shape[0] - 1)
    return coeff, 100.0 * latent / latent.sum()


def econ_pcs(Pfull):
    """Ksysid.m:1498-1507: smallest k with sum(explained(1:k)) >= 99."""
    coeff, explained = pca(Pfull)
    k = 1
    while explained[:k].sum() < 99:                   # :1502
        k += 1
    return coeff[:, :k]


@dataclass
class Dictionary:
    """Everything needed to evaluate lift.econ_full / econ_full_input
    (Ksysid.m:1594-1618 with dim_red, :1443-1491 without)."""
    model_type: str          # 'linear' | 'bilinear' | 'nonlinear'
    nzeta: int
    m: int
    basis: Basis             # over nzeta vars (linear/bilinear) or nzeta+m (nonlinear)
    pcs: np.ndarray | None = None

    @property
    def N(self):
        if self.pcs is None:
            return self.basis.nfull                   # Ksysid.m:534
        k = self.pcs.shape[1]
        if self.model_type == 'nonlinear':
            return k + self.nzeta + self.m + 1        # :1512
        return k + self.nzeta + 1                     # :1514-1516

    @property
    def W(self):
        """Width of Px (Ksysid.m:1019-1028)."""
        if self.model_type == 'bilinear':
            return self.N * (self.m + 1)
        if self.model_type == 'linear':
            return self.N + self.m
        return self.N


def econ_full(dic: Dictionary, V):
    """lift.econ_full: rows of V are zeta (linear/bilinear) or [zeta,u] (nonlinear).
    dim_red: [V ; pcs'*lift.full(V) ; 1] (Ksysid.m:1615-1618); else lift.full."""
    V = np.atleast_2d(V)
    full = lift_full(dic.basis, V)
    if dic.pcs is None:
        return full
    return np.hstack([V, full @ dic.pcs, np.ones((V.shape[0], 1))])


def lift_rows(dic: Dictionary, zeta, u):
    """One row block of Px/Py, Ksysid.m:1034-1064."""
    zeta = np.atleast_2d(zeta); u = np.atleast_2d(u)
    if dic.model_type == 'nonlinear':                 # :1039-1040
        return econ_full(dic, np.hstack([zeta, u]))
    psi = econ_full(dic, zeta)
    if dic.model_type == 'bilinear':                  # :1049, 1601-1602: kron(eye(m+1),psi)*[1;u]
        return np.hstack([psi] + [psi * u[:, [i]] for i in range(dic.m)])
    return np.hstack([psi, u])                        # :1062


def build_dictionary(model_type, nzeta, m, obs_type, obs_degree, pairs=None, dim_red=False,
                     gaussian_centres=None):
    """Constructor path Ksysid.m:115,137-142."""
    nv = nzeta + m if model_type == 'nonlinear' else nzeta
    basis = make_basis(nv, obs_type, obs_degree, gaussian_centres)
    dic = Dictionary(model_type, nzeta, m, basis)
    if dim_red:                                       # lift_snapshots Ksysid.m:1417-1431
        V = np.hstack([pairs['alpha'], pairs['u']]) if model_type == 'nonlinear' else pairs['alpha']
        dic.pcs = econ_pcs(lift_full(basis, V))
    return dic


# ----------------------------------------------------------------------------------
# get_Koopman (Ksysid.m:987-1092) and solve_KoopmanQP (:1095-1176)
# ----------------------------------------------------------------------------------

def px_py(dic, pairs):
    return lift_rows(dic, pairs['alpha'], pairs['u']), lift_rows(dic, pairs['beta'], pairs['u'])


def gram(Px, Py):
    """PxTPx, PxTPy of Ksysid.m:1114,1125."""
    return Px.T @ Px, Px.T @ Py


def koopman_ls(Px, Py):
    """K = Px \\ Py (Ksysid.m:1069) for full-column-rank Px: the unique least-squares
    solution (LAPACK QR in MATLAB; SVD-based lstsq here gives the same matrix)."""
    return np.linalg.lstsq(Px, Py, rcond=None)[0]


def project_l1_ball(v, t):
    """Euclidean projection of v onto {x : ||x||_1 <= t} (sort-based, exact)."""
    a = np.abs(v)
    if a.sum() <= t:
        return v.copy()
    s = np.sort(a)[::-1]
    css = np.cumsum(s)
    k = np.nonzero(s * np.arange(1, len(s) + 1) > (css - t))[0][-1]
    theta = (css[k] - t) / (k + 1.0)
    return np.sign(v) * np.maximum(a - theta, 0.0)


def koopman_lasso(G, C, t, iters=200000, tol=1e-13):
    """solve_KoopmanQP (Ksysid.m:1126-1137, delays=0): the QP in [K+;K-] >= 0,
    1'x <= t with H = M'(I (x) G)M is  min 1/2||Px K - Py||_F^2  s.t. ||vec K||_1 <= t.
    Solved by accelerated projected gradient with exact projection; returns K and a
    KKT residual.  The PSD guard of :1117-1120 is applied first."""
    if np.linalg.eigvalsh((G + G.T) / 2).min() < 0:   # :1117-1120
        G = G + 1e-6 * np.eye(G.shape[0])
    W = G.shape[0]
    L = np.linalg.eigvalsh((G + G.T) / 2).max()
    K = np.zeros((W, C.shape[1])); Y = K.copy(); tk = 1.0
    for _ in range(iters):
        grad = G @ Y - C
        Kn = project_l1_ball((Y - grad / L).ravel(), t).reshape(K.shape)
        tn = (1 + math.sqrt(1 + 4 * tk * tk)) / 2
        if np.vdot(Y - Kn, Kn - K) > 0:               # gradient restart
            tn = 1.0; Y = Kn.copy()
        else:
            Y = Kn + ((tk - 1) / tn) * (Kn - K)
        done = np.abs(Kn - K).max() <= tol * max(1.0, np.abs(Kn).max())
        K, tk = Kn, tn
        if done:
            break
    return K


def _lasso_column_path(G, c, theta_stop, max_steps):
    """Regularisation path of ONE column of the QP of Ksysid.m:1126-1137 in its multiplier form
    k(theta) = argmin 1/2 k'Gk - c'k + theta |k|_1, from theta = max|c| (k = 0) down to theta_stop, by the homotopy (LARS with
    drops): between breakpoints the support S and signs s are fixed and dk_S/d(-theta) = G_SS^-1 s_S (a dense solve per step here -
    the device keeps the inverse by rank-1 updates instead).  Returns breakpoints [(theta, |k|_1)] and k(theta_stop)."""
    W = G.shape[0]
    k = np.zeros(W); sgn = np.zeros(W); r = c.astype(float).copy()
    j0 = int(np.argmax(np.abs(r))); theta = abs(r[j0])
    bps = [(theta, 0.0)]
    if theta <= theta_stop or theta == 0.0:
        return bps, k
    S = [j0]; sgn[j0] = np.sign(r[j0])
    last_add, last_del, last_del_sgn = j0, -1, 0.0
    for _ in range(max_steps):
        if theta <= theta_stop:
            break
        Sa = np.array(S)
        d = np.linalg.solve(G[np.ix_(Sa, Sa)], sgn[Sa])
        a = G[:, Sa] @ d
        best, ev = theta - theta_stop, None
        off = np.ones(W, bool); off[Sa] = False
        io = np.flatnonzero(off)
        for s in (1.0, -1.0):                          # entry i enters with sign s when s (r_i - delta a_i) = theta - delta
            den = s * a[io] - 1.0; num = s * r[io] - theta
            with np.errstate(divide="ignore", invalid="ignore"):
                dl = num / den
            allowed = (den != 0) & ~((io == last_del) & (s == last_del_sgn))
            ok = allowed & (dl > 1e-14 * theta)
            now = allowed & ~ok & (num >= 0) & (den < 0)       # on the boundary and moving out (a tie with the event just taken): enters now
            dl = np.where(now, 0.0, dl); ok = ok | now
            if ok.any():
                j = int(np.argmin(np.where(ok, dl, np.inf)))
                if dl[j] < best:
                    best, ev = dl[j], ("add", int(io[j]), s)
        mov = (d * sgn[Sa] < 0) & (Sa != last_add)     # entry on the support reaches zero
        if mov.any():
            dl = np.where(mov, -k[Sa] / np.where(mov, d, 1.0), np.inf)
            j = int(np.argmin(dl))
            if max(dl[j], 0.0) < best:
                best, ev = max(dl[j], 0.0), ("del", int(Sa[j]), 0.0)
        k[Sa] += best * d; r = r - best * a; theta = theta - best if ev is not None else theta_stop
        last_add, last_del, last_del_sgn = -1, -1, 0.0
        if ev is not None:
            if ev[0] == "add":
                S.append(ev[1]); sgn[ev[1]] = ev[2]; last_add = ev[1]
            else:
                last_del, last_del_sgn = ev[1], sgn[ev[1]]
                S.remove(ev[1]); sgn[ev[1]] = 0.0; k[ev[1]] = 0.0
        bps.append((theta, np.abs(k).sum()))
    return bps, k


def koopman_lasso_path(G, C, t, max_steps_per_w=64):
    """solve_KoopmanQP (Ksysid.m:1126-1137, delays = 0) by the regularisation path: with the multiplier theta of the L1 row the
    columns of K separate, |K(theta)|_1 is piecewise linear and decreasing, and theta* with |K(theta*)|_1 = t is found on the
    breakpoints of the column paths; then every column is walked to theta*.  Exact (active-set) optimum whatever cond(G) is -
    the checker of the device's homotopy on the ill-conditioned arm Grams, where koopman_lasso's first-order iteration stalls.
    Returns K and theta*.  The PSD guard of :1117-1120 is applied first."""
    G = (G + G.T) / 2
    if np.linalg.eigvalsh(G).min() < 0:               # :1117-1120
        G = G + 1e-6 * np.eye(G.shape[0])
    W, nc = C.shape
    cap = max_steps_per_w * W + 512
    paths = [_lasso_column_path(G, C[:, j], 0.0, cap)[0] for j in range(nc)]

    def total(theta):
        s = 0.0
        for bp in paths:
            th = np.array([b[0] for b in bp]); l1 = np.array([b[1] for b in bp])
            s += 0.0 if theta >= th[0] else np.interp(-theta, -th, l1)
        return s
    if total(0.0) <= t:                               # inactive L1 row: the least-squares solution
        return np.column_stack([_lasso_column_path(G, C[:, j], 0.0, cap)[1] for j in range(nc)]), 0.0
    lo, hi = 0.0, max(bp[0][0] for bp in paths)
    for _ in range(300):
        mid = math.sqrt(lo * hi) if lo > 0 else hi * 2.0 ** -24
        if not (lo < mid < hi):
            break
        if total(mid) >= t:
            lo = mid
        else:
            hi = mid
    K = np.column_stack([_lasso_column_path(G, C[:, j], lo, cap)[1] for j in range(nc)])
    return K, lo


def lasso_kkt(G, C, K, t):
    """Optimality conditions of min 1/2<K,GK> - <C,K> s.t. |K|_1 <= t at K (PSD guard of Ksysid.m:1117-1120 applied): returns
    (theta, on, off, feas): the multiplier estimated on the support (median of -g_i sign k_i), max |g_i + theta sign k_i| on the
    support, max |g_i| / theta off it, and |K|_1 / t."""
    G = (G + G.T) / 2
    if np.linalg.eigvalsh(G).min() < 0:
        G = G + 1e-6 * np.eye(G.shape[0])
    g = G @ K - C
    on = K != 0
    theta = float(np.median(-(g * np.sign(K))[on])) if on.any() else float(np.abs(g).max())
    res_on = float(np.abs(g + theta * np.sign(K))[on].max()) if on.any() else 0.0
    off = float(np.abs(g[~on]).max() / theta) if (~on).any() and theta > 0 else 0.0
    return theta, res_on, off, float(np.abs(K).sum() / t)


def delay_pins(n, m, nd, N):
    """Equality rows of solve_KoopmanQP for LINEAR models with delays (Ksysid.m:1139-1164): the columns of K that
    produce the delayed part of zeta (columns n .. n(nd+1)+m nd - 1, i.e. vec rows n*Nm+1 : Nm*(n(nd+1)+mnd))
    are pinned to 0/1.  Returns (first fixed column, one past the last, list of (row, col) entries equal to 1),
    0-based, decoded from the literal index formulas of :1146-1157 (index - 1 = Nm * col_offset + row)."""
    Nm, nnd, mnd = N + m, n * nd, m * nd
    ones = []
    for i in range(nnd):                               # :1146-1149  state delays
        idx = (Nm + 1) * i
        ones.append((idx % Nm, n + idx // Nm))
    for i in range(m):                                 # :1150-1153  first input delay
        idx = Nm * nnd + N + (Nm + 1) * i
        ones.append((idx % Nm, n + idx // Nm))
    for i in range(m * (nd - 1)):                      # :1154-1157  subsequent input delays
        idx = Nm * (nnd + m) + nnd + (Nm + 1) * i
        ones.append((idx % Nm, n + idx // Nm))
    return n, n * (nd + 1) + mnd, ones


def koopman_lasso_delays(G, C, t, n, m, nd, N):
    """solve_KoopmanQP for a linear model with nd >= 1: the pinned columns are constants of the (column-separable)
    objective and use |1| of the L1 budget each, so the free columns solve the same problem with t - #ones."""
    c0, c1, ones = delay_pins(n, m, nd, N)
    W = G.shape[0]
    free = [j for j in range(C.shape[1]) if not (c0 <= j < c1)]
    K = np.zeros((W, C.shape[1]))
    for r, c in ones:
        K[r, c] = 1.0
    K[:, free] = koopman_lasso(G, C[:, free], t - len(ones))
    return K


def lasso_kkt_residual(G, C, K, t):
    """Optimality measure for the L1-ball problem: || K - P_ball(K - grad) ||_inf."""
    g = G @ K - C
    return np.abs(K - project_l1_ball((K - g).ravel(), t).reshape(K.shape)).max()


def get_koopman(dic, pairs, lasso=None, obj_lasso=1e6, n=None, nd=0):
    """Ksysid.m:987-1092.  `lasso` is the per-call argument (t = lasso*N, :996;
    default 1e4*N, :994,999); `obj_lasso` the class property tested at :1068."""
    Px, Py = px_py(dic, pairs)
    if obj_lasso >= 1e6:                              # :1068
        K = koopman_ls(Px, Py)
    else:
        t = (1e4 if lasso is None else lasso) * dic.N
        G, C = gram(Px, Py)
        if dic.model_type == 'linear' and nd >= 1:    # :1139 liftinput == 0 && nd >= 1
            K = koopman_lasso_delays(G, C, t, n, dic.m, nd, dic.N)
        else:
            K = koopman_lasso(G, C, t)
    N = dic.N
    return {'K': K, 'Px': Px[:, :N], 'Py': Py[:, :N], 'u': pairs['u'], 'alpha': pairs['alpha']}


# ----------------------------------------------------------------------------------
# Model extraction (Ksysid.m:1179-1341)
# ----------------------------------------------------------------------------------

def get_model(dic, koop, n):
    """Linear model with the M-projection, Ksysid.m:1189-1231 (discrete time)."""
    N = dic.N
    UT = koop['K'].T                                  # :1189
    A, B = UT[:N, :N], UT[:N, N:]                     # :1199-1200
    L = koop['Px'] @ A.T + koop['u'] @ B.T            # :1209-1211
    Mt = np.linalg.lstsq(L, koop['Py'], rcond=None)[0]  # :1216
    M = Mt.T
    C = np.hstack([np.eye(n), np.zeros((n, N - n))])  # :1203
    return {'A': M @ A, 'B': M @ B, 'C': C, 'M': M, 'K': koop['K'], 'A_raw': A, 'B_raw': B}


def get_blmodel(dic, koop, n):
    """Ksysid.m:1248-1278."""
    N = dic.N
    UT = koop['K'].T
    C = np.hstack([np.eye(n), np.zeros((n, N - n))])
    return {'A': UT[:N, :N], 'B': UT[:N, N:], 'C': C, 'K': koop['K']}


def beta_bilinear(B, z, m):
    """get_Beta_bilinear Ksysid.m:1285-1295: B*kron(eye(m), z)."""
    N = z.shape[0]
    return np.stack([B[:, i * N:(i + 1) * N] @ z for i in range(m)], axis=1)


def get_nlmodel(dic, koop, n):
    """Ksysid.m:1329: F(zeta,u) = K(:,1:nzeta)' * basis([zeta;u]); C = I_n (:1337)."""
    return {'Kf': koop['K'][:, :dic.nzeta].T.copy(), 'C': np.eye(n), 'K': koop['K']}


def nl_step(dic, model, zeta, u):
    return (model['Kf'] @ econ_full(dic, np.concatenate([zeta, u])[None, :])[0])


# ----------------------------------------------------------------------------------
# Validation rollouts and errors (Ksysid.m:1623-1898), delays = 0 or more
# ----------------------------------------------------------------------------------

def val_model(dic, model, val, nd, model_type=None):
    """val_model / val_BLmodel / val_NLmodel.  Returns dict(sim_y, real_y)."""
    mt = model_type or dic.model_type
    yreal = val['y'][nd:]; ureal = val['u'][nd:]       # :1630-1633
    zetareal, _ = get_zeta(val['y'], val['u'], nd)
    T = yreal.shape[0]
    n = yreal.shape[1]
    if mt == 'nonlinear':                              # :1841-1864
        zs = np.zeros_like(zetareal); zs[0] = zetareal[0]
        for j in range(T - 1):
            zs[j + 1] = nl_step(dic, model, zs[j], ureal[j])
        return {'sim_y': zs[:, :n], 'real_y': yreal, 't': val['t'][nd:]}
    z = econ_full(dic, zetareal[0][None, :])[0]        # :1674
    ysim = np.zeros_like(yreal); ysim[0] = yreal[0]    # :1654
    for j in range(T - 1):
        if mt == 'bilinear':                           # :1783
            z = model['A'] @ z + beta_bilinear(model['B'], z, dic.m) @ ureal[j]
        else:                                          # :1685
            z = model['A'] @ z + model['B'] @ ureal[j]
        ysim[j + 1] = model['C'] @ z
    return {'sim_y': ysim, 'real_y': yreal, 't': val['t'][nd:]}


def get_error(sim_y, real_y, sc=None):
    """Ksysid.m:1886-1897."""
    T = real_y.shape[0]
    d = sim_y - real_y
    err = {'abs': np.abs(d)}
    err['mean'] = err['abs'].mean(axis=0)
    err['rmse'] = np.sqrt((d ** 2).sum(axis=0) / T)
    err['nrmse'] = err['rmse'] / np.abs(real_y.max(axis=0) - real_y.min(axis=0))
    err['euclid'] = np.sqrt((d ** 2).sum(axis=1))
    err['euclid_mean'] = err['euclid'].sum() / T
    if sc is not None:
        du = scaleup(sc, 'y', sim_y) - scaleup(sc, 'y', real_y)
        err['unscaled_euclid_mean'] = np.sqrt((du ** 2).sum(axis=1)).sum() / T
    return err


# ----------------------------------------------------------------------------------
# Loaded systems (Ksysid.m `loaded` = true): the lifted state is psi (x) [1; w] for a load
# vector w (scaled like y and u, Ksysid.m:246-264)
# ----------------------------------------------------------------------------------

def loaded_lift(psi, w):
    """lift.econ_full_loaded (Ksysid.m:1606-1612, :596-599): kron(eye(nw+1), psi) * [1; w], i.e. the blocks
    [psi, w_1 psi, ..., w_nw psi].  psi: T x N, w: T x nw (or one row each)."""
    psi = np.atleast_2d(psi); w = np.atleast_2d(w)
    return np.hstack([psi] + [psi * w[:, [i]] for i in range(w.shape[1])])


def lift_rows_loaded(dic: Dictionary, zeta, u, w):
    """One row block of Px / Py for a loaded system, Ksysid.m:1034-1064 (loaded branches)."""
    zeta = np.atleast_2d(zeta); u = np.atleast_2d(u); w = np.atleast_2d(w)
    if dic.model_type == 'nonlinear':                 # :1036
        return loaded_lift(econ_full(dic, np.hstack([zeta, u])), w)
    psi = loaded_lift(econ_full(dic, zeta), w)        # :1056
    if dic.model_type == 'bilinear':                  # :1046, 1587-1590: kron(eye(m+1), full_loaded) * [1; u]
        return np.hstack([psi] + [psi * u[:, [i]] for i in range(dic.m)])
    return np.hstack([psi, u])                        # :1060-1061


def snapshot_pairs_loaded(data, nd, index=None):
    """snapshot_pairs plus the load that acts between the two states (Ksysid.m:953-957, :980-982);
    wzeta is w at the current time step (:896, :903)."""
    pairs = snapshot_pairs(data, nd, index)
    t = np.asarray(data['t']).ravel()
    good = np.nonzero(t[nd:-1] < t[nd + 1:])[0]
    w = np.atleast_2d(data['w'])[nd:][:-1][good]
    num_max = w.shape[0] - 1
    idx = np.arange(num_max) if index is None else np.asarray(index)
    pairs['w'] = w[idx]
    return pairs


def get_koopman_loaded(dic, pairs, lasso=None, obj_lasso=1e6):
    """get_Koopman for a loaded system (Ksysid.m:1005-1092): K is square of width N(nw+1) [+ m | x (m+1)];
    koopData.Px / Py keep the first N(nw+1) columns (:1085-1086)."""
    Px = lift_rows_loaded(dic, pairs['alpha'], pairs['u'], pairs['w'])
    Py = lift_rows_loaded(dic, pairs['beta'], pairs['u'], pairs['w'])
    nw = pairs['w'].shape[1]
    NL = dic.N * (nw + 1)
    if obj_lasso >= 1e6:
        K = koopman_ls(Px, Py)
    else:
        G, C = gram(Px, Py)
        K = koopman_lasso(G, C, (1e4 if lasso is None else lasso) * dic.N)      # t = lasso * N (:996)
    return {'K': K, 'Px': Px[:, :NL], 'Py': Py[:, :NL], 'u': pairs['u'], 'alpha': pairs['alpha'], 'w': pairs['w']}


def get_model_loaded(dic, koop, n):
    """get_model with loads (Ksysid.m:1192-1231): A is N(nw+1) square, B is N(nw+1) x m, M-projection as unloaded."""
    NL = koop['Px'].shape[1]
    UT = koop['K'].T
    A, B = UT[:NL, :NL], UT[:NL, NL:]                 # :1199-1200
    L = koop['Px'] @ A.T + koop['u'] @ B.T            # :1208-1210
    M = np.linalg.lstsq(L, koop['Py'], rcond=None)[0].T
    C = np.hstack([np.eye(n), np.zeros((n, NL - n))]) # :1203
    return {'A': M @ A, 'B': M @ B, 'C': C, 'M': M, 'K': koop['K']}


def get_blmodel_loaded(dic, koop, n):
    """get_BLmodel with loads (Ksysid.m:1251-1278)."""
    NL = koop['Px'].shape[1]
    UT = koop['K'].T
    C = np.hstack([np.eye(n), np.zeros((n, NL - n))])
    return {'A': UT[:NL, :NL], 'B': UT[:NL, NL:], 'C': C, 'K': koop['K']}


def get_nlmodel_loaded(dic, koop, n):
    """get_NLmodel with loads (Ksysid.m:1320-1327): F(zeta,u,w) = K(:,1:nzeta)' * basis_loaded."""
    return {'Kf': koop['K'][:, :dic.nzeta].T.copy(), 'C': np.eye(n), 'K': koop['K']}


def val_model_loaded(dic, model, val, nd, model_type=None):
    """val_model / val_BLmodel / val_NLmodel for loaded systems (Ksysid.m:1657-1671, :1751-1765, :1857-1858):
    every step re-lifts the first N entries of the simulated state with the ACTUAL load of that step,
    znow = kron(eye(nw+1), zsim(j,1:N)') * [1; w_j]."""
    mt = model_type or dic.model_type
    yreal = val['y'][nd:]; ureal = val['u'][nd:]; wreal = np.atleast_2d(val['w'])[nd:]
    zetareal, _ = get_zeta(val['y'], val['u'], nd)
    T, n, N = yreal.shape[0], yreal.shape[1], dic.N
    if mt == 'nonlinear':
        zs = np.zeros_like(zetareal); zs[0] = zetareal[0]
        for j in range(T - 1):
            psi = loaded_lift(econ_full(dic, np.concatenate([zs[j], ureal[j]])[None, :]), wreal[j][None, :])[0]
            zs[j + 1] = model['Kf'] @ psi
        return {'sim_y': zs[:, :n], 'real_y': yreal, 't': val['t'][nd:]}
    z = loaded_lift(econ_full(dic, zetareal[0][None, :]), wreal[0][None, :])[0]
    ysim = np.zeros_like(yreal); ysim[0] = yreal[0]
    for j in range(T - 1):
        znow = loaded_lift(z[None, :N], wreal[j][None, :])[0]
        if mt == 'bilinear':
            z = model['A'] @ znow + beta_bilinear(model['B'], znow, dic.m) @ ureal[j]
        else:
            z = model['A'] @ znow + model['B'] @ ureal[j]
        ysim[j + 1] = model['C'] @ z
    return {'sim_y': ysim, 'real_y': yreal, 't': val['t'][nd:]}


def _lsqlin_load(Cl, dl, nw, whatpast, pin_last_zero):
    """The lsqlin call of the load estimators (Kmpc.m:1354, :1442): min ||C x - d||^2 over x = [1; w] with
    x_1 = 1 (Aeq/beq), -1 <= x <= 1, and, when the previous estimate is given, |w_i - whatpast_i x_1| <= 0.01
    (the inequality block A, b at :1344-1347).  Solved as a QP in w after substituting x_1 = 1 (the toolbox solver
    is third-party; the problem is strictly convex whenever C(:,2:end) has full column rank, so its optimum is
    unique)."""
    C1, Cw = Cl[:, 0], Cl[:, 1:]
    free = list(range(nw - 1)) if (pin_last_zero and nw >= 1) else list(range(nw))
    what = np.zeros(nw)
    if free:
        Cf = Cw[:, free]
        r = dl - C1
        H = 2.0 * Cf.T @ Cf
        f = -2.0 * Cf.T @ r
        rows, rhs = [], []
        for k, i in enumerate(free):
            lo, hi = -1.0, 1.0
            if whatpast is not None:
                lo, hi = max(lo, whatpast[i] - 0.01), min(hi, whatpast[i] + 0.01)
            e = np.zeros(len(free)); e[k] = 1.0
            rows += [e, -e]; rhs += [hi, -lo]
        x, _, ok = qp_solve(H, f, np.array(rows), np.array(rhs))
        assert ok
        what[free] = x
    res = Cl @ np.concatenate([[1.0], what]) - dl
    return what, float(res @ res)


def estimate_load_linear(dic, model, ypast, upast, nw, nd=0, whatpast=None):
    """Kmpc.estimate_load_linear (Kmpc.m:1298-1356): over the past horizon, zeta_{k+1} = CA kron(I, psi(zeta_k)) [1; w]
    + CB u_k is linear in [1; w].  The shipped code pins the LAST load to zero through the debugging equality
    Aeq = blkdiag(1, 0, 1) (:1350), which only has the right size for nw = 2; that is reproduced for nw = 2."""
    zetapast, _ = get_zeta(ypast, upast, nd)
    hor, N, nz = zetapast.shape[0], dic.N, dic.nzeta
    CA, CB = model['A'][:nz, :], model['B'][:nz, :]
    rows, rhs = [], []
    for i in range(hor - 1):
        g = econ_full(dic, zetapast[i][None, :])[0]
        Om = np.kron(np.eye(nw + 1), g[:, None])        # :1323
        rows.append(CA @ Om)
        rhs.append(zetapast[i + 1, :nz] - CB @ upast[nd + i])      # :1328-1335
    return _lsqlin_load(np.vstack(rows), np.concatenate(rhs), nw, whatpast, pin_last_zero=(nw == 2))


def estimate_load_bilinear(dic, model, ypast, upast, nw, nd=0, whatpast=None):
    """Kmpc.estimate_load_bilinear (Kmpc.m:1360-1444): rows (CA + sum_j u_j CB_j) kron(I, psi(zeta_k)); Aeq pins
    only the leading 1 (:1436)."""
    zetapast, _ = get_zeta(ypast, upast, nd)
    hor, N, nz, m = zetapast.shape[0], dic.N, dic.nzeta, dic.m
    NL = N * (nw + 1)
    CA = model['A'][:nz, :]
    rows, rhs = [], []
    for i in range(hor - 1):
        g = econ_full(dic, zetapast[i][None, :])[0]
        Om = np.kron(np.eye(nw + 1), g[:, None])
        R = CA @ Om
        for j in range(m):                               # :1388-1393
            R = R + model['B'][:nz, j * NL:(j + 1) * NL] @ Om * upast[i, j]
        rows.append(R)
        rhs.append(zetapast[i + 1, :nz])
    return _lsqlin_load(np.vstack(rows), np.concatenate(rhs), nw, whatpast, pin_last_zero=False)


# ----------------------------------------------------------------------------------
# Kmpc: cost / constraint matrices and the per-step QP (Kmpc.m)
# ----------------------------------------------------------------------------------

@dataclass
class MpcSetup:
    """Kmpc constructor state (Kmpc.m:44-103) for the linear-MPC types."""
    model_type: str
    A: np.ndarray
    B: np.ndarray
    m: int
    Np: int
    projmtx: np.ndarray
    cost_running: float = 0.1
    cost_terminal: float = 100.0
    cost_input: object = 0.0          # scalar or length-m vector (column vector in MATLAB)
    input_bounds: np.ndarray | None = None   # m x 2, ALREADY scaled down (Kmpc.m:659)
    slope_lim: float | None = None           # input_slopeConst*mean(u_factor) (Kmpc.m:684)
    smooth_lim: float | None = None          # Ts^2*input_smoothConst*mean(u_factor) (:706)
    state_bounds: np.ndarray | None = None   # n x 2 scaled down (:726)
    n: int = 0


def _cost_common(s: MpcSetup):
    N = s.A.shape[0]; Np = s.Np
    Ahat = np.vstack([np.linalg.matrix_power(s.A, i) for i in range(Np + 1)])  # Kmpc.m:168-172,528-532
    nproj = s.projmtx.shape[0]
    Chat = np.kron(np.eye(Np + 1), s.projmtx)          # :193,540
    Q = np.kron(np.eye(Np + 1), np.eye(nproj) * s.cost_running)  # :197,544
    Q[-nproj:, -nproj:] = np.eye(nproj) * s.cost_terminal        # :198,545
    ci = np.asarray(s.cost_input, dtype=np.float64)
    Ri = np.eye(s.m) * (ci.reshape(-1, 1) if ci.ndim else ci)    # eye(m).*cost_input  :201,548
    R = np.kron(np.eye(Np), Ri)
    return N, Ahat, Chat, Q, R


def cost_B(s: MpcSetup, Z):
    """get_costB_bilinear (Kmpc.m:569-596) / the Bhat of get_costMatrices (:175-190).
    Z: (rows x N) lifted state row(s); ignored for the linear model."""
    N = s.A.shape[0]; Np = s.Np; m = s.m
    Z = np.atleast_2d(Z) if Z is not None else None
    Bcol = np.zeros((N * (Np + 1), m))
    for i in range(1, Np + 1):                         # :578-585 / :178-180
        if s.model_type == 'bilinear':
            z = Z[i - 1] if Z.shape[0] > 1 else Z[0]
            Bm = beta_bilinear(s.B, z, m)
        else:
            Bm = s.B
        Bcol[N * i:N * (i + 1)] = np.linalg.matrix_power(s.A, i - 1) @ Bm
    Bh = np.zeros((N * (Np + 1), m * Np))
    Bh[:, :m] = Bcol
    for i in range(1, Np):                             # Lshift :587-595
        Bh[N:, i * m:(i + 1) * m] = Bh[:-N, (i - 1) * m:i * m]
    return Bh


def constraint_FEc(s: MpcSetup):
    """get_constraintMatrices(_bilinear) Kmpc.m:226-318 / :638-730."""
    N = s.A.shape[0]; Np = s.Np; m = s.m
    ncolB, nrowB = m * Np, N * (Np + 1)
    F, E, c = [], [], []
    if s.input_bounds is not None:                     # :230-253
        num = 2 * m
        Fb = np.zeros((num * (Np + 1), ncolB))
        Fb[:num * Np, :] = np.kron(np.eye(Np), np.vstack([-np.eye(m), np.eye(m)]))
        cb = np.zeros(num * (Np + 1))
        cb[:num * Np] = np.tile(np.concatenate([-s.input_bounds[:, 0], s.input_bounds[:, 1]]), Np)
        F.append(Fb); E.append(np.zeros((Fb.shape[0], nrowB))); c.append(cb)
    if s.slope_lim is not None:                        # :256-277
        neg = np.hstack([np.kron(np.eye(Np - 1), -np.eye(m)), np.zeros((m * (Np - 1), m))])
        pos = np.hstack([np.zeros((m * (Np - 1), m)), np.kron(np.eye(Np - 1), np.eye(m))])
        top = neg + pos
        Fs = np.vstack([top, -top])
        F.append(Fs); E.append(np.zeros((Fs.shape[0], nrowB)))
        c.append(np.full(Fs.shape[0], s.slope_lim))
    if s.smooth_lim is not None:                       # :280-297
        I = np.eye(m); K2 = np.kron(np.eye(Np - 2), I); Zc = np.zeros((m * (Np - 2), m))
        top = (np.hstack([K2, Zc, Zc]) + np.hstack([Zc, -2 * K2, Zc]) + np.hstack([Zc, Zc, K2]))
        Fm = np.vstack([top, -top])
        F.append(Fm); E.append(np.zeros((Fm.shape[0], nrowB)))
        c.append(np.full(Fm.shape[0], s.smooth_lim))
    if s.state_bounds is not None:                     # :300-318
        n = s.n; num = 2 * n
        Es = np.zeros((num * (Np + 1), nrowB))
        # NOTE literal restatement of :306: the kron block is written into the first
        # (Np+1)*n COLUMNS of E (it is not strided by N).
        Es[:, :(Np + 1) * n] = np.kron(np.eye(Np + 1), np.vstack([-np.eye(n), np.eye(n)]))
        E.append(Es); F.append(np.zeros((Es.shape[0], ncolB)))
        c.append(np.tile(np.concatenate([-s.state_bounds[:, 0], s.state_bounds[:, 1]]), Np + 1))
    if not F:
        return np.zeros((0, ncolB)), np.zeros((0, nrowB)), np.zeros(0)
    return np.vstack(F), np.vstack(E), np.concatenate(c)


def pad_ref(ref, Np):
    """Kmpc.m:354-365: truncate / repeat last row to Np+1 rows, Yr = vec(ref')."""
    ref = np.atleast_2d(ref)
    if ref.shape[0] > Np + 1:
        ref = ref[:Np + 1]
    elif ref.shape[0] < Np + 1:
        ref = np.vstack([ref, np.tile(ref[-1], (Np + 1 - ref.shape[0], 1))])
    return ref.reshape(-1)


def mpc_qp(s: MpcSetup, z, u_prev, ref, zhor=None):
    """The QP of one get_mpcInput / get_mpcInput_bilinear(_iter) call, literally:
    returns (Hq, f, Aineq, b) for quadprog(Hq=2H, f, Aineq, b)  (Kmpc.m:368-383,
    861-883).  zhor: rows used by get_costB_bilinear (default z)."""
    N, Ahat, Chat, Q, R = _cost_common(s)
    Zb = np.atleast_2d(z if zhor is None else zhor)
    Bh = cost_B(s, Zb)
    CB = Chat @ Bh
    H = CB.T @ Q @ CB + R                              # :204,604
    G = 2 * Ahat.T @ Chat.T @ Q @ CB                   # :205,613
    D = -2 * Q @ CB                                    # :206,621
    Yr = pad_ref(ref, s.Np)
    f = (z @ G + Yr @ D)                               # :369,879
    F, E, c = constraint_FEc(s)
    # the constraint matrix is formed ONCE from the current lifted state, before the linearisation passes
    # (A = get_constraintL_bilinear(zrow), Kmpc.m:861, outside the loop of :874-899): it does not follow zhorizon
    L = F + E @ (Bh if zhor is None else cost_B(s, np.atleast_2d(z)))     # :324,745
    M = E @ Ahat                                       # :325,737
    b = -M @ z + c                                     # :371,862
    m = s.m
    Atack = np.hstack([np.vstack([np.eye(m), -np.eye(m)]), np.zeros((2 * m, L.shape[1] - m))])  # :374,865
    Aq = np.vstack([L, Atack])
    bq = np.concatenate([b, u_prev, -u_prev])          # :377-379
    return 2 * H, f, Aq, bq


def qp_solve(Hq, f, A, b, tol=1e-10, maxit=5000):
    """quadprog(Hq,f,A,b): min 1/2 x'Hq x + f'x s.t. A x <= b for SPD Hq.
    Goldfarb-Idnani dual active set (exact up to rounding).  Returns (x, lam, ok).
    Stands in for the Optimization Toolbox call at Kmpc.m:383,810,883 (third-party,
    not in the repo); on failure the reference seam returns NaN
    (quadprog_gurobi.m:18-23).

    Invariants: H x + f + A_act' lam = 0, a_i'x = b_i on the active set, lam >= 0.
    Adding violated row p: x(t) = x - t z, lam(t) = lam - t r, lam_p(t) = lam_p + t with
    r = (N'H^-1 N)^-1 N'H^-1 a_p,  z = H^-1 (a_p - N r),  N = A_act'."""
    n = Hq.shape[0]
    Hinv = np.linalg.inv(Hq)
    x = -Hinv @ f
    nrm = np.linalg.norm(A, axis=1)
    safe = np.where(nrm > 0, nrm, 1.0)
    act, lam = [], np.zeros(0)
    it = 0
    while it < maxit:
        it += 1
        viol = (A @ x - b) / safe
        cand = viol.copy()
        cand[act] = -np.inf
        p = int(np.argmax(cand))
        if cand[p] <= tol:
            lam_full = np.zeros(A.shape[0]); lam_full[act] = lam
            return x, lam_full, True
        if nrm[p] == 0:                   # 0'x <= b_p with b_p < 0
            return np.full(n, np.nan), None, False
        ap = A[p]
        lam_p = 0.0
        while it < maxit:
            it += 1
            if act:
                Na = A[act].T
                HN = Hinv @ Na
                r = np.linalg.solve(Na.T @ HN, HN.T @ ap)
                z = Hinv @ ap - HN @ r
            else:
                r = np.zeros(0)
                z = Hinv @ ap
            t1, l = np.inf, -1
            pos = np.nonzero(r > 1e-13)[0]
            if pos.size:
                ratios = lam[pos] / r[pos]
                j = int(np.argmin(ratios)); t1 = ratios[j]; l = int(pos[j])
            apz = ap @ z
            t2 = (ap @ x - b[p]) / apz if apz > 1e-13 * (ap @ Hinv @ ap) else np.inf
            t = min(t1, t2)
            if not np.isfinite(t):
                return np.full(n, np.nan), None, False      # infeasible
            lam = lam - t * r
            lam_p += t
            if np.isfinite(t2):
                x = x - t * z
            if t2 <= t1:                  # full step: p becomes active
                act.append(p); lam = np.append(lam, lam_p)
                break
            act.pop(l); lam = np.delete(lam, l)             # partial step: drop l
    return np.full(n, np.nan), None, False


def qp_kkt_residual(Hq, f, A, b, x, lam):
    """max of stationarity, primal infeasibility, dual infeasibility, complementarity."""
    stat = np.abs(Hq @ x + f + A.T @ lam).max()
    prim = max(0.0, (A @ x - b).max())
    dual = max(0.0, (-lam).max())
    comp = np.abs(lam * (A @ x - b)).max()
    return max(stat, prim, dual, comp)


def mpc_step(s: MpcSetup, z, u_prev, ref, iters=1):
    """get_mpcInput (Kmpc.m:329-387) / get_mpcInput_bilinear_iter (:817-904).
    Returns U (Np x m; NaN on failure) and the KKT residual of the last QP."""
    zhor = np.atleast_2d(z)
    for it in range(iters):
        Hq, f, Aq, bq = mpc_qp(s, z, u_prev, ref, zhor)
        x, lam, ok = qp_solve(Hq, f, Aq, bq)
        if not ok:
            return np.full((s.Np, s.m), np.nan), np.inf
        U = x.reshape(s.Np, s.m)                       # :884
        if it == iters - 1:
            break
        zh = np.zeros((s.Np + 1, z.shape[0])); zh[0] = z   # :891-895
        for j in range(s.Np):
            zh[j + 1] = s.A @ zh[j] + beta_bilinear(s.B, zh[j], s.m) @ U[j]
        zhor = zh
    return U, qp_kkt_residual(Hq, f, Aq, bq, x, lam)
