"""kp_sym_eig (the `pca` eigensolver) on covariance-like matrices: time, residual, orthogonality."""
import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, koopman_realizations_amd as kra
ctx = kra.Context(0)
for n in (34, 41, 63, 84, 85, 136, 220, 256, 511, 816):
    rng = np.random.default_rng(n)
    X = rng.standard_normal((4000, n)) * np.logspace(0, -6, n)      # decaying spectrum like a lifted-snapshot covariance
    S = np.cov(X, rowvar=False)
    ctx.sym_eig(S)
    t0 = time.perf_counter(); w, V, sw = ctx.sym_eig(S); dt = time.perf_counter() - t0
    t1 = time.perf_counter(); wl, Vl = np.linalg.eigh(S); dl = time.perf_counter() - t1
    print(f"n {n:3d}: device {dt * 1e3:7.2f} ms ({sw} sweeps)  host LAPACK eigh {dl * 1e3:6.2f} ms  |S V - V w| {np.abs(S @ V - V * w).max():.1e}"
          f"  |V'V - I| {np.abs(V.T @ V - np.eye(n)).max():.1e}  max|w - w_lapack| {np.abs(np.sort(w) - wl).max():.1e}")
