"""Oracle error tables of the 64 generated random systems of tests/test_gpu_sweep.py (Rsys restatement, seed 21):
evaluate_rand_models.m:47-143 per system by the numpy oracle (scale -> pairs -> SVD least squares -> model -> rollout ->
normalised mean error), 23 values each.  1.6 s per system on one core, so the table is computed here (8 processes) and
committed; the GPU test recomputes three systems live and checks them against the stored rows before it trusts the rest.

    python tests/golden/make_sweep_oracle.py        ->  tests/golden/sweep_oracle_seed21.npz
"""
import multiprocessing as mp
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(HERE))

DEGREES = {"linear": 13, "bilinear": 6, "nonlinear": 4}


def _one(d):
    os.environ.setdefault("OMP_NUM_THREADS", "1")
    src = open(os.path.join(os.path.dirname(HERE), "test_gpu_sweep.py")).read().split("def test_eval_system_matches_oracle")[0]
    ns = {}
    exec(src.replace("pytestmark = pytest.mark.gpu", ""), ns)
    return ns["_oracle_system"](d, DEGREES)


def main():
    from koopman_realizations_amd.rsys import Rsys
    r = Rsys(64, 3, 3, 2, seed=21)
    systems = Rsys.save_data(r.simulate_systems_fast(10.0, 0.01, 11, np.zeros((1, 1))))
    with mp.get_context("fork").Pool(8) as pool:
        res = pool.map(_one, systems)
    out = {mt: np.stack([x[mt] for x in res], axis=1) for mt in DEGREES}        # (degrees, 64), the sweep's own layout
    np.savez_compressed(os.path.join(HERE, "sweep_oracle_seed21.npz"), **out)
    print({k: v.shape for k, v in out.items()})


if __name__ == "__main__":
    main()
