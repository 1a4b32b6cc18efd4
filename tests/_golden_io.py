"""Loading the committed fixtures of tests/golden/*.npz (made by tests/golden/make_golden.py)."""
import json
from pathlib import Path

import numpy as np

GOLDEN = Path(__file__).parent / "golden"


def load(name):
    z = np.load(GOLDEN / name)
    meta = json.loads(str(z["meta"]))
    inputs = {k[3:]: np.asfortranarray(z[k]) for k in z.files if k.startswith("in_")}
    outputs = {k[4:]: np.asfortranarray(z[k]) for k in z.files if k.startswith("out_")}
    return inputs, outputs, meta


def setup_from(jr, inputs, meta):
    """the miniapp builder that made the inputs gives the grid / coefficients / BC objects; the arrays are the stored ones"""
    kw = dict(meta["builder_kwargs"])
    ni = tuple(kw.pop("ni"))
    s = getattr(jr.miniapps, meta["builder"])(ni, **kw)
    drift = [k for k in inputs if not np.array_equal(s.arrays[k], inputs[k], equal_nan=True)]
    s.arrays = {k: v.copy(order="F") for k, v in inputs.items()}
    return s, drift


def rel_err(got, ref):
    scale = np.abs(ref[np.isfinite(ref)]).max() if np.isfinite(ref).any() else 1.0
    m = np.isfinite(ref)
    if not np.array_equal(np.isfinite(got), m):
        return float("inf")
    return float(np.abs(got[m] - ref[m]).max() / max(scale, 1e-300)) if m.any() else 0.0
