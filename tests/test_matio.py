"""MAT-file reader on files written with the reference's nesting (cell arrays of 1x1 structs)."""
import os
import subprocess
import sys

import numpy as np
import pytest

from tests._loaded_system import make_trials

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def write_mat(path, trials, nval=2, wrap=False):
    import scipy.io as sio
    def cell(ts):
        c = np.empty((1, len(ts)), dtype=object)
        for i, t in enumerate(ts):
            c[0, i] = {"t": t["t"].reshape(-1, 1), "y": t["y"], "u": t["u"], "w": t["w"]}
        return c
    d = {"train": cell(trials[:-nval]), "val": cell(trials[-nval:])}
    sio.savemat(path, {"data4sysid": d} if wrap else d)


@pytest.mark.parametrize("wrap", [False, True])
def test_reader_maps_cells_of_structs(tmp_path, wrap):
    sys.path.insert(0, ROOT)
    import importlib.util
    spec = importlib.util.spec_from_file_location("kp_matio", os.path.join(ROOT, "koopman-realizations_amd", "matio.py"))
    matio = importlib.util.module_from_spec(spec); spec.loader.exec_module(matio)
    trials = make_trials(5, 40, nw=2)
    f = str(tmp_path / "d.mat")
    write_mat(f, trials, wrap=wrap)
    d = matio.load_data4sysid(f)
    assert len(d["train"]) == 3 and len(d["val"]) == 2
    for got, exp in zip(d["train"] + d["val"], trials):
        assert got["t"].shape == (40,) and np.array_equal(got["y"], exp["y"]) and np.array_equal(got["u"], exp["u"]) and np.array_equal(got["w"], exp["w"])


@pytest.mark.gpu
def test_cli_fits_a_mat_file(tmp_path):
    trials = make_trials(8, 120, nw=1)
    f = str(tmp_path / "d.mat"); out = str(tmp_path / "model.npz")
    write_mat(f, trials)
    r = subprocess.run([sys.executable, "-m", "koopman_realizations_amd", "sysid", f, "--model_type", "bilinear", "--obs_degree", "2", "--loaded",
                        "--out", out], cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    assert "val trial 1" in r.stdout
    m = np.load(out)
    assert m["A"].shape == (12, 12) and m["B"].shape == (12, 12)           # N = 6, nw = 1: N (nw + 1) = 12; m = 1
