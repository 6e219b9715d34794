#!/usr/bin/env python3
"""Golden GRADIENT vectors (CONTAINER ONLY: needs /root/reference; never runs on the GPU box).

Runs the reference's own ``AVNet`` (stubs as in make_golden.py; the absent third-party ``sru.SRU`` is a differentiable torch
restatement, oracle/grad_oracle.py:sru_forward_torch) in float64 through the reference's own ``PITLossWrapper(PairwiseNegSDR("snr"))``,
calls ``loss.backward()`` and stores, for every parameter, the gradient (tensors up to 4096 elements in full, larger ones as 4096 seeded
samples + their L2 norm).  Two cases: ``eval`` (BatchNorm running statistics, dropout off - torch autograd still records) and
``train`` (BatchNorm batch statistics; the dropout probabilities are set to 0 so the case is deterministic).
tests/test_oracle_golden.py checks oracle/grad_oracle.py's whole-model gradients against these, which pins the gradient oracle the HIP
backward is tested against to the reference's autograd (except inside the SRU cell, which is third-party: see DESIGN.md (c)).
Writes tests/golden/grad_R2_L4096_B2.npz.
"""
import importlib
import os
import sys
import zlib

import numpy as np
import torch
import torch.nn as nn

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
from oracle import grad_oracle, make_golden as MG  # noqa: E402
from oracle.params import make_inputs  # noqa: E402

N_SAMPLE = 4096


class _SRU(nn.Module):
    """Differentiable stand-in for sru.SRU (same parameter names as upstream)."""

    def __init__(self, input_size, hidden_size, num_layers=2, bidirectional=False, **kw):
        super().__init__()
        assert bidirectional
        self.rnn_lst = nn.ModuleList([MG._SRUCell(input_size if i == 0 else 2 * hidden_size, hidden_size, True) for i in range(num_layers)])

    def forward(self, x):
        return grad_oracle.sru_forward_torch(x, [(c.weight, c.weight_c, c.bias) for c in self.rnn_lst]), None


def sample_idx(name, n):
    rs = np.random.RandomState(zlib.crc32(name.encode()) & 0x7FFFFFFF)
    return rs.choice(n, N_SAMPLE, replace=False).astype(np.int64)


def main():
    MG._SRU = _SRU
    AVNet = MG.import_reference()
    L = importlib.import_module("src.losses")
    B, Ls, Tv = 2, 4096, 7
    wav, emb = make_inputs(B, Ls, Tv, seed=5)
    tgt = (0.05 * np.random.default_rng(6).standard_normal((B, 1, Ls))).astype(np.float32)
    out = {}
    for case in ("eval", "train"):
        m, spec, sd = MG.build(AVNet, 2, seed=0)
        m = m.double()
        m.train(case == "train")
        for mod in m.modules():  # deterministic: no dropout (DropPath is the identity stub)
            if isinstance(mod, nn.Dropout):
                mod.p = 0.0
            if isinstance(mod, nn.MultiheadAttention):
                mod.dropout = 0.0
        with torch.enable_grad():
            est = m(torch.from_numpy(wav).double(), torch.from_numpy(emb).double())
            loss = L.PITLossWrapper(L.PairwiseNegSDR("snr"), pit_from="pw_mtx")(est, torch.from_numpy(tgt).double())
            loss.backward()
        out[f"{case}/loss"] = np.float64(loss.item())
        out[f"{case}/est"] = est.detach().numpy().astype(np.float32)
        for k, p in m.named_parameters():
            g = p.grad.detach().numpy().reshape(-1)
            key = f"{case}/{k}"
            if g.size <= N_SAMPLE:
                out[key] = g.astype(np.float64)
            else:
                out[key] = g[sample_idx(k, g.size)].astype(np.float64)
                out[key + "#l2"] = np.float64(np.sqrt((g ** 2).sum()))
        print(case, "loss", loss.item(), "parameters", sum(1 for _ in m.named_parameters()))
    path = os.path.join(ROOT, "tests", "golden", "grad_R2_L4096_B2.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path) // 1024, "KiB")


if __name__ == "__main__":
    main()
