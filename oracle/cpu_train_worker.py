"""CPU baseline worker of `bench.py --train` (test infrastructure, never the product path): one training step - forward, PIT neg-SNR loss,
backward - of the autograd restatement (oracle/grad_oracle.py) on this box's host cores, torch intra-op threads = argv[3].
Prints one JSON line {"n": mixtures, "seconds": wall time of the timed steps, "threads": T}."""
import json
import sys
import time

import numpy as np
import torch


def main():
    repeats, batch, threads = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
    steps = int(sys.argv[4]) if len(sys.argv) > 4 else 1
    dt = torch.float32 if (len(sys.argv) > 5 and sys.argv[5] == "f32") else torch.float64  # the reference trains in f32; the checker is f64
    torch.set_num_threads(threads)
    from oracle import grad_oracle as G
    from oracle.params import load_spec, make_inputs, make_state_dict
    sd = make_state_dict(load_spec("state_spec_R4.json"), 0)
    wav, emb = make_inputs(batch, 32000, 50, 0)
    tgt = (0.05 * np.random.default_rng(1).standard_normal((batch, 1, 32000)))
    done, t0 = 0, time.perf_counter()
    for _ in range(steps):
        p = {k: torch.tensor(v, dtype=dt, requires_grad=True) for k, v in sd.items() if "num_batches" not in k}
        est = G.avnet_torch(torch.tensor(wav, dtype=dt), torch.tensor(emb, dtype=dt), p, repeats, vp_trainable=True,
                            bn_train=True)
        loss = G.pit_loss_torch(est, torch.tensor(tgt, dtype=dt), "snr")
        loss.backward()
        done += batch
    print(json.dumps({"n": done, "seconds": time.perf_counter() - t0, "steps": steps, "threads": threads, "loss": float(loss.detach()), "dtype": str(dt)}), flush=True)


if __name__ == "__main__":
    main()
