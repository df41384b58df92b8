import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import koopman_realizations_amd as kra
from koopman_realizations_amd import _ffi as F
from bench import synth_pairs
ctx = kra.Context(0)
a, b, u = synth_pairs(100000, seed=5)
tab = kra.poly_exponent_table(6, 3)
pcs = np.linalg.qr(np.random.default_rng(3).standard_normal((84, 27)))[0]
bs = kra.Basis(ctx, "bilinear", 6, 3, [("poly", tab[6:])], pcs)
for what in (F.LIFT_ECON, F.LIFT_FULL):
    for _ in range(3):
        bs.lift(what, a, u)
        print("what", what, "lift kernel ms", ctx.timer(4))
