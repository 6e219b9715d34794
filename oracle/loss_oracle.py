"""CPU restatement of the evaluation-side loss path (TEST INFRASTRUCTURE ONLY): pairwise negative SNR / SI-SDR / SD-SDR
and the permutation-invariant wrapper around it ("the step after the path").

  pairwise_neg_sdr   reference src/losses/matrix.py:13-53   (PairwiseNegSDR.forward)
  pit_from_pw_mtx    reference src/losses/pit_wrapper.py:28-52,84-110 (PITLossWrapper.forward with pit_from="pw_mtx",
                     find_best_perm_factorial with perm_reduce=None, reordered_sources)

Pinned by tests/golden/loss_*.npz, generated from the reference's own classes (oracle/make_golden_loss.py).
float64 arithmetic on float32 inputs; the reference runs the same formulas in float32.
"""
from __future__ import annotations

from itertools import permutations

import numpy as np

EPS = 1e-8


def pairwise_neg_sdr(ests, targets, sdr_type="sisdr", zero_mean=True, take_log=True):
    """ests, targets (B, n_src, L) -> (B, n_src[est], n_src[target]) negative SDR (matrix.py:22-53)."""
    if ests.shape != targets.shape or targets.ndim != 3:
        raise TypeError(f"Inputs must be of shape [batch, n_src, time], got {ests.shape} and {targets.shape} instead")
    e = ests.astype(np.float64)
    t = targets.astype(np.float64)
    if zero_mean:  # matrix.py:27-31
        t = t - t.mean(2, keepdims=True)
        e = e - e.mean(2, keepdims=True)
    s_t = t[:, None, :, :]  # (B, 1, n, L)
    s_e = e[:, :, None, :]  # (B, n, 1, L)
    if sdr_type in ("sisdr", "sdsdr"):  # matrix.py:35-41
        dot = (s_e * s_t).sum(3, keepdims=True)
        energy = (s_t ** 2).sum(3, keepdims=True) + EPS
        proj = dot * s_t / energy
    else:
        proj = np.broadcast_to(s_t, (e.shape[0], e.shape[1], e.shape[1], e.shape[2]))
    noise = s_e - s_t if sdr_type in ("sdsdr", "snr") else s_e - proj  # matrix.py:45-48
    sdr = (proj ** 2).sum(3) / ((noise ** 2).sum(3) + EPS)
    if take_log:
        sdr = 10.0 * np.log10(sdr + EPS)
    return (-sdr).astype(np.float32)


def pit_from_pw_mtx(pw_loss, ests=None):
    """pw_loss (B, n_est, n_tgt) -> (mean over batch of the best permutation's loss, min_loss (B), perms (B, n) with
    perms[b][i] = estimate assigned to target i, reordered estimates or None) (pit_wrapper.py:40-52,84-110)."""
    B, n, _ = pw_loss.shape
    perms = np.array(list(permutations(range(n))), dtype=np.int64)  # lexicographic, as itertools in the reference
    pwl = pw_loss.astype(np.float64).transpose(0, 2, 1)  # dim 1 targets, dim 2 estimates
    loss_set = np.stack([pwl[:, np.arange(n), p].sum(1) / n for p in perms], 1)  # (B, n!)
    idx = loss_set.argmin(1)  # first minimum, like torch.min on CPU
    min_loss = loss_set[np.arange(B), idx]
    batch_idx = perms[idx]
    reordered = None if ests is None else np.stack([ests[b][batch_idx[b]] for b in range(B)])
    return np.float32(min_loss.mean()), min_loss.astype(np.float32), batch_idx, reordered


LOSS_CASES = [(3, 2, 4000, "random"), (4, 2, 32000, "close"), (2, 3, 5000, "perm"), (2, 2, 257, "offset")]


def make_loss_case(k: int):
    """Seeded (estimates, targets) of golden case k (shared by oracle/make_golden_loss.py and the tests, so the fixture
    file only holds the reference's outputs)."""
    B, n, T, mode = LOSS_CASES[k]
    rs = np.random.RandomState(7700 + k)
    tgt = (rs.randn(B, n, T) * 0.05).astype(np.float32)
    if mode == "random":
        est = (rs.randn(B, n, T) * 0.05).astype(np.float32)
    elif mode == "close":  # good separation, sources swapped for odd b: exercises the high-SDR cancellation
        est = tgt + (rs.randn(B, n, T) * 1e-3).astype(np.float32)
        est[1::2] = est[1::2, ::-1].copy()
    elif mode == "perm":
        est = (tgt[:, [2, 0, 1]] * 0.7 + (rs.randn(B, n, T) * 0.01)).astype(np.float32)
    else:  # non-zero means
        est = (tgt * 1.3 + 0.02 + rs.randn(B, n, T) * 0.02).astype(np.float32)
        tgt = (tgt - 0.01).astype(np.float32)
    return np.ascontiguousarray(est), np.ascontiguousarray(tgt)
