"""Stock-torch-ops CPU composition of the same forward (TEST / MEASUREMENT INFRASTRUCTURE ONLY -- never imported by the
product package).  SURVEY 8d asks for two CPU baselines on the GPU box's host cores: the numpy restatement
(``rtfs_oracle``) and "the stock-torch-ops composition of the same modules".  This module provides the second one
without a second copy of the composition logic: ``torch_ops()`` temporarily rebinds the oracle's compute primitives
(1x1 / depthwise / dense convolutions, gLN, eval BatchNorm, LayerNorm4D, PReLU / ReLU / sigmoid) to
``torch.nn.functional`` calls on CPU tensors (multi-threaded MKL / oneDNN kernels, fp32), while the call graph, shapes
and parameter handling stay the oracle's (each primitive cites the reference in rtfs_oracle.py).  The SRU recurrence,
attention softmax and index glue remain numpy.  ``tests/test_oracle_golden.py`` holds this path to the numpy oracle.
"""
from __future__ import annotations

import contextlib

import numpy as np
import torch
import torch.nn.functional as F

from . import rtfs_oracle as O

f32 = np.float32


def _t(a):
    return torch.from_numpy(np.ascontiguousarray(a, dtype=f32))


def _n(t):
    return t.detach().numpy().astype(f32, copy=False)


def sigmoid(x):
    return _n(torch.sigmoid(_t(x)))


def prelu(x, a):
    return _n(F.prelu(_t(x), _t(np.asarray(a).reshape(-1)[:1])))


def relu(x):
    return _n(F.relu(_t(x)))


def gln(x, w, b, eps=O.EPS):
    return _n(F.group_norm(_t(x), 1, _t(w), _t(b), eps))


def batchnorm_eval(x, w, b, rm, rv, eps=O.EPS):
    return _n(F.batch_norm(_t(x), _t(rm), _t(rv), _t(w), _t(b), False, 0.0, eps))


def ln4d(x, gamma, beta, eps=O.EPS):
    xt = _t(x)
    axes = (1, 3) if gamma.shape[-1] > 1 else (1,)
    mu = xt.mean(dim=axes, keepdim=True)
    var = xt.var(dim=axes, keepdim=True, unbiased=False)
    return _n((xt - mu) / torch.sqrt(var + eps) * _t(gamma) + _t(beta))


def pointwise(x, w, b=None):
    xt = _t(x)
    sp = xt.shape[2:]
    y = F.conv1d(xt.reshape(xt.shape[0], xt.shape[1], -1), _t(w).reshape(w.shape[0], w.shape[1], 1), None if b is None else _t(b))
    return _n(y.reshape((xt.shape[0], w.shape[0]) + tuple(sp)))


def grouped_pointwise_1d(x, w, b, groups):
    return _n(F.conv1d(_t(x), _t(w), None if b is None else _t(b), groups=groups))


def dwconv2d(x, w, b=None, stride=1, pad=None):
    xt = _t(x)
    kh, kw = w.shape[-2:]
    if pad is None:
        (pt, pb), (pl, pr) = O._same_pads(kh), O._same_pads(kw)
    else:
        pt = pb = pl = pr = pad
    xt = F.pad(xt, (pl, pr, pt, pb))
    return _n(F.conv2d(xt, _t(w), None if b is None else _t(b), stride=stride, groups=xt.shape[1]))


def dwconv1d(x, w, b=None, stride=1, pad=None):
    xt = _t(x)
    k = w.shape[-1]
    pl, pr = O._same_pads(k) if pad is None else (pad, pad)
    return _n(F.conv1d(F.pad(xt, (pl, pr)), _t(w), None if b is None else _t(b), stride=stride, groups=xt.shape[1]))


def conv2d_dense_same(x, w):
    kh, kw = w.shape[-2:]
    (pt, pb), (pl, pr) = O._same_pads(kh), O._same_pads(kw)
    return _n(F.conv2d(F.pad(_t(x), (pl, pr, pt, pb)), _t(w)))


def conv_transpose2d_s1(x, w, pad):
    return _n(F.conv_transpose2d(_t(x), _t(w), None, stride=1, padding=pad))


_PRIMS = ("sigmoid", "prelu", "relu", "gln", "batchnorm_eval", "ln4d", "pointwise", "grouped_pointwise_1d", "dwconv2d",
          "dwconv1d", "conv2d_dense_same", "conv_transpose2d_s1")


@contextlib.contextmanager
def torch_ops():
    """Within the block the oracle's primitives run on stock torch CPU ops."""
    saved = {k: getattr(O, k) for k in _PRIMS}
    try:
        for k in _PRIMS:
            setattr(O, k, globals()[k])
        yield
    finally:
        for k, v in saved.items():
            setattr(O, k, v)


def avnet_forward(wav, mouth_emb, sd, repeats=4):
    with torch_ops(), torch.no_grad():
        return O.avnet_forward(wav, mouth_emb, sd, repeats=repeats)
