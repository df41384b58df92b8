import sys, os, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
import koopman_realizations_amd as kra
from oracle import koopman_oracle as ko
from test_gpu_fit import make_basis
g = np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests', 'golden', 'arm_data.npz'))
import conftest
ctx = kra.Context(0)
# arm fixture equivalent
lens = g["train_len"]; off = np.concatenate([[0], np.cumsum(lens)])
train = [{"t": g["train_t"][a:b], "y": g["train_y"][a:b], "u": g["train_u"][a:b]} for a, b in zip(off[:-1], off[1:])]
sd, so = ko.get_scale(ko.merge_trials(train)); pairs = ko.snapshot_pairs(sd, 0)
for env in ("", "1"):
    if env: os.environ["KP_NO_GRAM_CONGRUENCE"] = "1"; os.environ["KP_NO_GRAM3_LINEAR"] = "1"
    for mt in ("linear", "nonlinear"):
        dic = ko.build_dictionary(mt, 6, 3, ["poly"], [3], pairs, True)
        b = make_basis(ctx, dic)
        snaps = kra.Snapshots(ctx, pairs["alpha"], pairs["beta"], pairs["u"])
        K = kra.fit(ctx, b, snaps)[0]
        Px, Py = ko.px_py(dic, pairs)
        Kref = ko.koopman_ls(Px, Py)
        G, C = kra.fit_gram(ctx, b, snaps)
        print("general kernel" if env else "congruence   ", mt, "N", dic.N, "cond(Px) %.2e" % np.linalg.cond(Px), "K err %.2e" % (np.abs(K - Kref).max() / np.abs(Kref).max()),
              "G err %.2e" % (np.abs(G - Px.T @ Px).max() / np.abs(G).max()), "C err %.2e" % (np.abs(C - Px.T @ Py).max() / np.abs(G).max()))
