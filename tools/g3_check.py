"""Gram parity of the Kronecker kernel on small bilinear dictionaries for several snapshot counts."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import koopman_realizations_amd as kra
from oracle import koopman_oracle as ko
from conftest import synth_pairs
ctx = kra.Context(0)
for nz, m in ((1, 1), (2, 1), (6, 3)):
    for deg in (1, 2, 3, 6) if nz < 6 else (2, 3):
        for Ns in (1003, 9000, 9008, 20000):
            pairs = synth_pairs(Ns, nz, m, seed=7)
            dic = ko.build_dictionary("bilinear", nz, m, ["poly"], [deg])
            blocks = [("poly", dic.basis.blocks[0][1][dic.basis.nvars:].astype(np.uint8))]
            b = kra.Basis(ctx, "bilinear", nz, m, blocks, None)
            snaps = kra.Snapshots(ctx, pairs["alpha"], pairs["beta"], pairs["u"])
            G, C = kra.fit_gram(ctx, b, snaps)
            Px, Py = ko.px_py(dic, pairs)
            Gr, Cr = ko.gram(Px, Py)
            print(nz, m, deg, Ns, "G err %.2e C err %.2e" % (np.abs(G - Gr).max() / np.abs(Gr).max(), np.abs(C - Cr).max() / np.abs(Gr).max()))
