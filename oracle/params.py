"""Deterministic synthetic parameters for parity tests (TEST INFRASTRUCTURE ONLY).

``make_state_dict(spec, seed)`` fills every entry of a reference ``state_dict`` layout
(``spec`` = list of ``[name, shape, dtype]`` as captured in
``tests/golden/state_spec_*.json``) from a per-name ``numpy.random.RandomState`` so
that the golden-vector generator (which loads them into the imported reference),
the oracle and the HIP path all see bit-identical weights without the weights
themselves being committed.  Norm affines, PReLU slopes and BatchNorm running
statistics get non-default values so no term is silently an identity.
"""
from __future__ import annotations

import json
import math
import os
import zlib

import numpy as np


def load_spec(path_or_name: str):
    if not os.path.exists(path_or_name):
        path_or_name = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden", path_or_name)
    with open(path_or_name) as f:
        return json.load(f)


def _pe(shape):
    _, n, c = shape
    pos = np.arange(n, dtype=np.float32)[:, None]
    div = np.exp(np.arange(0, c, 2, dtype=np.float32) * np.float32(-(math.log(10000.0) / c))).astype(np.float32)
    pe = np.zeros((n, c), np.float32)
    pe[:, 0::2] = np.sin((pos * div).astype(np.float32))
    pe[:, 1::2] = np.cos((pos * div).astype(np.float32))
    return pe[None]


def _is_norm_param(name: str) -> bool:
    toks = name.split(".")
    if "norm" in toks or "norm1" in toks or "norm2" in toks:
        return True
    # BatchNorm sits at Sequential index 3 of a ConvNormAct: ...full_layer.3.weight
    return len(toks) >= 3 and toks[-3] == "full_layer" and toks[-2] == "3"


def make_param(name: str, shape, dtype: str, seed: int = 0) -> np.ndarray:
    rs = np.random.RandomState((zlib.crc32(name.encode()) ^ (seed * 0x9E3779B1)) & 0x7FFFFFFF)
    shape = tuple(shape)
    leaf = name.split(".")[-1]
    if leaf == "num_batches_tracked":
        return np.zeros(shape, np.int64)
    if leaf == "pe":
        return _pe(shape)
    if leaf == "running_mean":
        return (rs.randn(*shape) * 0.1).astype(np.float32)
    if leaf == "running_var":
        return rs.uniform(0.5, 1.5, shape).astype(np.float32)
    if _is_norm_param(name):
        if leaf in ("weight", "gamma"):
            return rs.uniform(0.5, 1.5, shape).astype(np.float32)
        return rs.uniform(-0.2, 0.2, shape).astype(np.float32)
    if shape == (1,) and leaf == "weight":  # PReLU slope
        return rs.uniform(0.1, 0.4, shape).astype(np.float32)
    if leaf == "weight_c":
        return rs.uniform(-1.0, 1.0, shape).astype(np.float32)
    if "weight_hh_l" in leaf or "weight_ih_l" in leaf:  # nn.LSTM: (4*hidden, in)
        a = math.sqrt(3.0 / shape[1])
        return rs.uniform(-a, a, shape).astype(np.float32)
    if "bias_ih_l" in leaf or "bias_hh_l" in leaf:
        return rs.uniform(-0.3, 0.3, shape).astype(np.float32)
    if leaf in ("bias", "in_proj_bias"):
        lim = 0.5 if "rnn_lst" in name else 0.1
        return rs.uniform(-lim, lim, shape).astype(np.float32)
    # dense / conv / conv-transpose / SRU projection weights
    if "rnn_lst" in name:
        fan_in = shape[0]
    elif name.endswith("linear.weight") or name.endswith("decoder.decoder.weight"):
        fan_in = shape[0] * int(np.prod(shape[2:]))
    else:
        fan_in = int(np.prod(shape[1:]))
    a = math.sqrt(3.0 / max(fan_in, 1))
    return rs.uniform(-a, a, shape).astype(np.float32)


def make_state_dict(spec, seed: int = 0) -> dict:
    return {name: make_param(name, shape, dtype, seed) for name, shape, dtype in spec}


def make_inputs(B: int, L: int, Tv: int, seed: int = 0):
    """Synthetic 2-speaker mixture + lip embedding (SURVEY 8d distributions)."""
    rs = np.random.RandomState(1234 + seed)
    s1 = (rs.randn(B, L) * 0.05).astype(np.float32)
    s2 = (rs.randn(B, L) * 0.05).astype(np.float32)
    emb = rs.randn(B, 512, Tv).astype(np.float32)
    return (s1 + s2).astype(np.float32), emb
