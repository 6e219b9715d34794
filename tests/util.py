"""Shared helpers for the parity tests."""
import json
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")


def load_golden(name):
    return np.load(os.path.join(GOLDEN, name + ".npz"))


def spec_R4():
    with open(os.path.join(GOLDEN, "state_spec_R4.json")) as f:
        return json.load(f)


def rand(shape, seed, scale=1.0):
    """Must match oracle/make_golden.py:rand."""
    return (np.random.RandomState(seed).randn(*shape) * scale).astype(np.float32)


def rel_err(a, b):
    """max |a-b| / max|b|  (the north_star's '1e-4 rel' is read as relative to the tensor's scale)."""
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-30))


def l2_rel(a, b):
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    return float(np.sqrt(((a - b) ** 2).sum()) / max(np.sqrt((b ** 2).sum()), 1e-30))


def check_probe(g, tag, x, tol, what=""):
    """Compare a full tensor ``x`` with the probe ``tag`` stored in golden file ``g``."""
    x = np.asarray(x, np.float32)
    shape = tuple(int(v) for v in g[tag + ".shape"])
    assert x.shape == shape, f"{what}{tag}: shape {x.shape} != golden {shape}"
    val = g[tag + ".val"]
    got = x.reshape(-1)[g[tag + ".idx"]]
    scale = max(float(np.abs(val).max()), 1e-30)
    err = float(np.abs(got.astype(np.float64) - val).max()) / scale
    assert err <= tol, f"{what}{tag}: probe rel err {err:.3e} > {tol:.1e}"
    st = g[tag + ".stats"]
    flat = x.reshape(-1).astype(np.float64)
    mine = np.array([flat.mean(), np.abs(flat).mean(), np.sqrt((flat ** 2).sum())])
    # abs-mean and L2 are scale-like; the mean can cancel, so compare it against abs-mean
    assert abs(mine[1] - st[1]) <= 10 * tol * st[1] + 1e-12, f"{what}{tag}: abs-mean {mine[1]} vs {st[1]}"
    assert abs(mine[2] - st[2]) <= 10 * tol * st[2] + 1e-12, f"{what}{tag}: L2 {mine[2]} vs {st[2]}"
    assert abs(mine[0] - st[0]) <= 10 * tol * st[1] + 1e-12, f"{what}{tag}: mean {mine[0]} vs {st[0]}"
    return err
