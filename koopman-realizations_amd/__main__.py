"""Command line: identify a Koopman model from a reference data file, without MATLAB.

    python -m koopman_realizations_amd sysid DATA.mat --model_type bilinear --obs_degree 3 [--dim_red] [--out model.npz]

is example_sysid.m:22-65 (Ksysid constructor, train_models, validation of every `val` trial) with the options of
Ksysid_setup.m; prints the validation errors and optionally stores the model matrices."""
from __future__ import annotations

import argparse
import sys

import numpy as np


def main(argv=None):
    ap = argparse.ArgumentParser(prog="koopman_realizations_amd")
    sub = ap.add_subparsers(dest="cmd", required=True)
    s = sub.add_parser("sysid", help="fit a model to a data4sysid MAT-file and validate it")
    s.add_argument("data")
    s.add_argument("--model_type", default="linear", choices=["linear", "bilinear", "nonlinear"])
    s.add_argument("--obs_type", nargs="+", default=["poly"])
    s.add_argument("--obs_degree", nargs="+", type=int, default=[3])
    s.add_argument("--snapshots", type=float, default=float("inf"))
    s.add_argument("--lasso", nargs="+", type=float, default=[float("inf")])
    s.add_argument("--delays", type=int, default=0)
    s.add_argument("--dim_red", action="store_true")
    s.add_argument("--loaded", action="store_true")
    s.add_argument("--device", type=int, default=0)
    s.add_argument("--out", default=None, help="write the model (A, B, C, K, scale) to this .npz")
    a = ap.parse_args(argv)

    from . import Context, Ksysid          # needs libkoopman_hip.so and a GPU: fails loudly otherwise
    from .matio import load_data4sysid
    data = load_data4sysid(a.data)
    ctx = Context(a.device)
    ks = Ksysid(data, ctx=ctx, model_type=a.model_type, obs_type=a.obs_type, obs_degree=a.obs_degree, snapshots=a.snapshots,
                lasso=a.lasso if len(a.lasso) > 1 else a.lasso[0], delays=a.delays, dim_red=a.dim_red, loaded=a.loaded)
    ks.train_models()
    val = {"linear": ks.val_model, "bilinear": ks.val_BLmodel, "nonlinear": ks.val_NLmodel}[a.model_type]
    p = ks.params
    print(f"{a.model_type} model: n={p['n']} m={p['m']} nzeta={p['nzeta']} N={p['N']}  pairs={len(ks.snapshotPairs['alpha'])}")
    for i, v in enumerate(ks.valdata):
        e = val(ks.model, v)["error"]
        print(f"val trial {i}: rmse {np.array2string(e['rmse'], precision=4)}  nrmse {np.array2string(e['nrmse'], precision=4)}  "
              f"mean euclid {e['euclid_mean']:.5f}")
    if a.out:
        m = ks.model
        np.savez(a.out, **{k: np.asarray(m[k]) for k in ("A", "B", "C", "K", "Kf", "M") if k in m},
                 **{"scale_" + k: v for k, v in p["scale"].items()})
        print("wrote", a.out)
    return 0


if __name__ == "__main__":
    sys.exit(main())
