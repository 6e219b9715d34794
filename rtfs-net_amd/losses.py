"""Losses of the reference on the device: ``src/losses/matrix.py`` ``PairwiseNegSDR`` and
``src/losses/pit_wrapper.py`` ``PITLossWrapper`` (``pit_from="pw_mtx"``, factorial search), same class names, constructor
keywords and return values.  The arithmetic runs in ``librtfs_amd.so`` (``rtfs_pit_pairwise_sdr_f32``: one pass over the
signals, float64 moments, permutation search in the same kernel; ``rtfs_pit_sdr_backward_f32`` for the gradient w.r.t. the
estimates when autograd is recording); there is no CPU fallback.
"""
from __future__ import annotations

import torch
import torch.nn as nn

from . import _lib

_KIND = {"snr": 0, "sisdr": 1, "sdsdr": 2}


def _pairwise(ests, targets, kind, zero_mean, take_log):
    if targets.size() != ests.size() or targets.ndim != 3:  # matrix.py:23-24
        raise TypeError(f"Inputs must be of shape [batch, n_src, time], got {ests.size()} and {targets.size()} instead")
    _lib.need_gpu(ests, targets)
    lib = _lib.load()
    ests, targets = ests.contiguous().float(), targets.contiguous().float()
    B, n, L = ests.shape
    if n > 4:
        raise ValueError("MI355X PairwiseNegSDR supports n_src <= 4")
    pw = torch.empty(B, n, n, device=ests.device, dtype=torch.float32)
    min_loss = torch.empty(B, device=ests.device, dtype=torch.float32)
    perm = torch.empty(B, n, device=ests.device, dtype=torch.int32)
    _lib.check(lib.rtfs_pit_pairwise_sdr_f32(_lib.ptr(ests), _lib.ptr(targets), B, n, L, _KIND[kind], int(zero_mean), int(take_log),
                                             _lib.ptr(pw), _lib.ptr(min_loss), _lib.ptr(perm), _lib.stream_of(ests)),
               "rtfs_pit_pairwise_sdr_f32")
    return pw, min_loss, perm


class _PitLossFn(torch.autograd.Function):
    """min over permutations of the mean pairwise loss, per batch element, with its gradient w.r.t. the estimates."""

    @staticmethod
    def forward(ctx, ests, targets, kind, zero_mean, take_log):
        pw, min_loss, perm = _pairwise(ests, targets, kind, zero_mean, take_log)
        ctx.save_for_backward(ests.contiguous().float(), targets.contiguous().float(), perm)
        ctx.cfg = (kind, zero_mean, take_log)
        ctx.mark_non_differentiable(perm)
        return min_loss, perm

    @staticmethod
    def backward(ctx, dmin, _dperm):
        lib = _lib.load()
        ests, targets, perm = ctx.saved_tensors
        kind, zero_mean, take_log = ctx.cfg
        B, n, L = ests.shape
        dmin = dmin.contiguous().float()
        dests = torch.empty_like(ests)
        _lib.check(lib.rtfs_pit_sdr_backward_f32(_lib.ptr(ests), _lib.ptr(targets), _lib.ptr(perm), _lib.ptr(dmin), _lib.ptr(dests), B, n, L,
                                                 _KIND[kind], int(zero_mean), int(take_log), _lib.stream_of(ests)), "rtfs_pit_sdr_backward_f32")
        return dests, None, None, None, None


class PairwiseNegSDR(nn.Module):
    """reference matrix.py:13-53.  forward(ests, targets) -> (B, n_src, n_src) negative SDR, [b, est, target]."""

    def __init__(self, sdr_type, zero_mean=True, take_log=True, EPS=1e-8):
        super().__init__()
        assert sdr_type in ["snr", "sisdr", "sdsdr"]
        if EPS != 1e-8:
            raise ValueError("MI355X PairwiseNegSDR uses the reference's EPS = 1e-8")
        self.sdr_type, self.zero_mean, self.take_log, self.EPS = sdr_type, zero_mean, take_log, EPS

    def forward(self, ests, targets):
        return _pairwise(ests, targets, self.sdr_type, self.zero_mean, self.take_log)[0]


class PITLossWrapper(nn.Module):
    """reference pit_wrapper.py:15-116 for ``pit_from="pw_mtx"`` with a ``PairwiseNegSDR`` loss and ``perm_reduce=None``
    (what the RTFS-Net configs use).  forward(ests, targets, return_ests=False) -> mean loss [, reordered estimates]."""

    def __init__(self, loss_func, pit_from="pw_mtx", perm_reduce=None):
        super().__init__()
        if pit_from not in ["pw_mtx", "pw_pt", "perm_avg"]:
            raise ValueError("Unsupported loss function type {} for now. Expectedone of [`pw_mtx`, `pw_pt`, `perm_avg`]".format(pit_from))
        if pit_from != "pw_mtx" or perm_reduce is not None or not isinstance(loss_func, PairwiseNegSDR):
            raise ValueError("MI355X PITLossWrapper supports pit_from='pw_mtx' around PairwiseNegSDR with perm_reduce=None")
        self.loss_func, self.pit_from, self.perm_reduce = loss_func, pit_from, perm_reduce

    def forward(self, ests, targets, return_ests=False, reduce_kwargs=None, **kwargs):
        f = self.loss_func
        if torch.is_grad_enabled() and ests.requires_grad:  # training: the HIP backward kernel hangs off min_loss
            min_loss, perm = _PitLossFn.apply(ests, targets, f.sdr_type, f.zero_mean, f.take_log)
        else:
            _, min_loss, perm = _pairwise(ests, targets, f.sdr_type, f.zero_mean, f.take_log)
        mean_loss = torch.mean(min_loss)
        if not return_ests:
            return mean_loss
        return mean_loss, self.reordered_sources(ests, perm.long())

    @staticmethod
    def reordered_sources(source, batch_indices):
        return torch.gather(source, 1, batch_indices[:, :, None].expand(-1, -1, source.shape[2]))


pairwise_neg_sisdr = PairwiseNegSDR("sisdr")
pairwise_neg_sdsdr = PairwiseNegSDR("sdsdr")
pairwise_neg_snr = PairwiseNegSDR("snr")
