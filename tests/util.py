"""Helpers shared by the tests: golden loading and ulp-aware comparisons."""
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load(name):
    return np.load(os.path.join(GOLDEN, name + ".npz"))


def steps(g):
    out = []
    for t in range(int(g["n_steps"])):
        pre = "s%d_" % t
        out.append({k[len(pre):]: g[k] for k in g.files if k.startswith(pre)})
    return out


def rel_err(a, b):
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    den = np.maximum(np.abs(b), 1e-300)
    with np.errstate(invalid="ignore"):
        e = np.abs(a - b) / den
    e = np.where((a == b) | (np.isnan(a) & np.isnan(b)), 0.0, e)
    return float(np.max(e)) if e.size else 0.0
