"""Randomised parity sweep of the fused lift+Gram kernels against the numpy oracle: bilinear monomial dictionaries over many
(nzeta, m, degree, dim_red, Ns) shapes - ragged tails, one to three tiles, splits with a single tile, every Kronecker-kernel variant
(m = 1, 2, 3; in-kernel projection and prelifted rows; in-loop power table at its dense and its padded row stride).
usage: python tools/gram_fuzz.py [cases] [seed]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import koopman_realizations_amd as kra
from oracle import koopman_oracle as ko
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from conftest import synth_pairs
from test_gpu_fit import make_basis

ncases = int(sys.argv[1]) if len(sys.argv) > 1 else 200
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
ctx = kra.Context(0)
worst = 0.0
t0 = time.time()
done = 0
refused = 0
for case in range(ncases):
    mt = rng.choice(["bilinear", "bilinear", "bilinear", "linear", "nonlinear"])
    nz = int(rng.integers(1, 13)); m = int(rng.integers(1, 4))
    deg = int(rng.integers(1, 5))
    nv = nz + (m if mt == "nonlinear" else 0)
    from math import comb
    nfull = comb(nv + deg, deg)
    if nfull > 230 or nfull < 3:
        continue
    dim_red = bool(rng.integers(0, 4) == 0) and nfull > nv + 3
    Ns = int(rng.choice([1, 5, 7, 8, 9, 15, 16, 17, 23, 24, 25, 63, 100, 511, 512, 513, 1003, 4099, 5999, 6001, 8191, 20011, 100003]))
    pairs = synth_pairs(Ns, nz, m, seed=int(rng.integers(1 << 30)))
    types, degs, centres = ["poly"], [deg], []
    kind = int(rng.integers(0, 6))
    if kind == 1 and nz <= 3:
        types, degs = ["poly", "fourier"], [min(deg, 2), 1]
    elif kind == 2 and nz <= 8:
        ng = int(rng.integers(2, 21))
        types, degs, centres = ["gaussian", "poly"], [ng, min(deg, 2)], [rng.uniform(-1, 1, (nv, ng))]
    elif kind == 3 and nz <= 3:
        types, degs = ["fourier"], [int(rng.integers(1, 3))]
    try:
        dic = ko.build_dictionary(mt, nz, m, types, degs, pairs if dim_red else None, dim_red, centres)
    except Exception as e:
        continue
    if dic.W > 512:
        continue
    b = make_basis(ctx, dic)
    Px, Py = ko.px_py(dic, pairs)
    Gr, Cr = ko.gram(Px, Py)
    snaps = kra.Snapshots(ctx, pairs["alpha"], pairs["beta"], pairs["u"])
    try:
        G, C = kra.fit_gram(ctx, b, snaps)
    except kra.KoopmanHipError as e:                 # a documented refusal (KP_ERR_ARG with a message), never a wrong answer
        refused += 1
        snaps.close(); b.close()
        continue
    G2, C2 = kra.fit_gram(ctx, b, snaps)
    scale = max(np.abs(Gr).max(), 1e-300)
    err = max(np.abs(G - Gr).max(), np.abs(C - Cr).max()) / scale
    ok = err <= 2e-12 and (G == G.T).all() and (G2 == G).all() and (C2 == C).all()
    worst = max(worst, err)
    done += 1
    if not ok:
        print("FAIL", mt, nz, m, types, degs, dim_red, Ns, "W", dic.W, "err", err, "sym", (G == G.T).all(), "repeat", (G2 == G).all() and (C2 == C).all())
        sys.exit(1)
    snaps.close(); b.close()
print(f"{done} cases OK ({refused} refused with an error message) in {time.time() - t0:.1f} s, worst relative error {worst:.2e}")
