import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden():
    import numpy as np
    d = os.path.join(ROOT, "tests", "golden")
    return {k: np.load(os.path.join(d, k + ".npz")) for k in ("arm_data", "arm_blockM", "blockM_ref", "rand_systems", "arm_plant", "arm_circle")}


@pytest.fixture(scope="session")
def ctx():
    import koopman_realizations_amd as kra
    c = kra.Context(0)
    yield c
    c.close()


@pytest.fixture(scope="session")
def arm(golden):
    """Scaled arm training data, snapshot pairs (all, in order) and scale factors via the oracle."""
    from oracle import koopman_oracle as ko
    g = golden["arm_data"]
    data = {"t": g["train_t"], "y": g["train_y"], "u": g["train_u"]}
    sd, sc = ko.get_scale(data)
    pairs = ko.snapshot_pairs(sd, 0)
    return {"scaled": sd, "scale": sc, "pairs": pairs, "raw": data,
            "val": {"t": g["val_t"], "y": g["val_y"], "u": g["val_u"]}}


def synth_pairs(Ns, nz=6, m=3, seed=0):
    """Synthetic snapshot pairs of SURVEY 8(d): uniform alpha,u; beta a smooth map of them."""
    import numpy as np
    rng = np.random.default_rng(seed)
    alpha = rng.uniform(-1, 1, (Ns, nz)); u = rng.uniform(-1, 1, (Ns, m))
    Mx = rng.standard_normal((nz + m, nz)) * 0.3
    beta = np.clip(alpha + 0.05 * np.tanh(np.hstack([alpha, u]) @ Mx), -1, 1)
    return {"alpha": alpha, "beta": beta, "u": u}
