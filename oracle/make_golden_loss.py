#!/usr/bin/env python3
"""Golden vectors of the loss path (CONTAINER ONLY: needs /root/reference).  Runs the reference's own
``PairwiseNegSDR`` / ``PITLossWrapper`` (src/losses/matrix.py, pit_wrapper.py; pure torch) on seeded inputs and stores
the outputs under tests/golden/loss_cases.npz (inputs are regenerated from seeds by oracle.loss_oracle.make_loss_case)."""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
from oracle import make_golden as G  # noqa: E402


def main():
    G.import_reference()
    import importlib
    L = importlib.import_module("src.losses")
    from oracle.loss_oracle import LOSS_CASES, make_loss_case
    out = {}
    cases = []
    for k in range(len(LOSS_CASES)):
        est, tgt = make_loss_case(k)
        for kind in ("snr", "sisdr", "sdsdr"):
            pw = L.PairwiseNegSDR(kind)(torch.from_numpy(est), torch.from_numpy(tgt))
            mean_loss, reordered = L.PITLossWrapper(L.PairwiseNegSDR(kind), pit_from="pw_mtx")(
                torch.from_numpy(est), torch.from_numpy(tgt), return_ests=True)
            out[f"c{k}_{kind}_pw"] = pw.numpy()
            out[f"c{k}_{kind}_mean"] = np.float32(mean_loss.item())
            # the permutation, recovered from the reordered estimates: reordered[b][i] == est[b][perm[b][i]]
            ro = reordered.numpy()
            perm = np.array([[int(np.argmax([np.array_equal(ro[b, i], est[b, j]) for j in range(est.shape[1])]))
                              for i in range(est.shape[1])] for b in range(est.shape[0])], np.int32)
            assert all(np.array_equal(ro[b, i], est[b, perm[b, i]]) for b in range(est.shape[0]) for i in range(est.shape[1]))
            out[f"c{k}_{kind}_perm"] = perm
        cases.append(k)
    out["cases"] = np.array(cases)
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "loss_cases.npz"), **out)
    print("wrote loss_cases.npz", {k: v.shape for k, v in out.items() if k.endswith("_pw")})


if __name__ == "__main__":
    main()
