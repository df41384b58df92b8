"""Reader for the reference's MAT-file (v5) data sets, so that a fit can be run without MATLAB.

Layout of a `data4sysid` file as the reference's Data class writes it and Ksysid reads it (Ksysid.m:46-59, 119-126;
e.g. datafiles/arm-3link-markers-noload-50trials_train-10_val-5.mat): top-level `train` and `val`, each a 1 x k cell
array of 1 x 1 structs with fields t (T x 1), y (T x n), u (T x m) and optionally w (T x nw, loads), x, params.
Also accepts a file holding one struct `data4sysid` with those two fields.  (scipy.io.loadmat does the v5 container
parsing; this module only maps the cell / struct nesting onto the dicts the Ksysid mirror takes.)"""
from __future__ import annotations

import numpy as np


def _trial(s):
    names = getattr(s, "_fieldnames", None)
    if names is None:
        raise ValueError("expected a struct with fields t, y, u")
    out = {}
    for k in ("t", "y", "u", "w", "x"):
        if k in names:
            v = np.asarray(getattr(s, k), dtype=np.float64)
            out[k] = v.ravel() if k == "t" else np.atleast_2d(v)
    for k in ("t", "y", "u"):
        if k not in out:
            raise ValueError(f"trial without field {k!r}")
    return out


def _trials(cell):
    a = np.asarray(cell, dtype=object)
    items = []
    for e in a.ravel():
        while isinstance(e, np.ndarray) and e.dtype == object and e.size == 1:      # 1 x 1 struct array inside a cell
            e = e.ravel()[0]
        if isinstance(e, np.ndarray) and e.dtype == object:
            items.extend(_trials(e))
        else:
            items.append(_trial(e))
    return items


def load_data4sysid(path):
    """-> {'train': [trial, ...], 'val': [trial, ...]}, trial = {'t','y','u'[, 'w', 'x']} as float64 arrays."""
    import scipy.io as sio
    d = sio.loadmat(path, squeeze_me=False, struct_as_record=False)
    if "train" not in d and "data4sysid" in d:
        s = d["data4sysid"].ravel()[0]
        d = {"train": s.train, "val": s.val}
    if "train" not in d or "val" not in d:
        raise ValueError("MAT-file has neither train/val nor data4sysid")
    return {"train": _trials(d["train"]), "val": _trials(d["val"])}
