"""CPU baseline per SURVEY 8(d) (TEST / MEASUREMENT INFRASTRUCTURE ONLY): times the CPU restatement of the forward
(`oracle.rtfs_oracle`, numpy) and the stock-torch-ops composition of the same call graph (`oracle.torch_cpu`) on this box's host
cores at batch 1 and batch 4: `torch.set_num_threads(threads)` (and the BLAS pools), 3 warm-up + 5 timed forwards per leg.
The torch-ops legs - the faster implementation, the one `cpu_baseline.value` comes from - always run the full 3 + 5 (about 50 s in all on
the GPU box); the numpy legs (12 s per batch-4 forward) are bounded in wall time and report the forwards they finished (at least 1 warm-up + 2 timed).
bench.py starts this as a child process BEFORE it touches the GPU and folds the JSON line into `cpu_baseline`.

    python -m oracle.cpu_bench <repeats> <threads> <seconds_per_leg> [<L> <Tv>]
"""
import json
import os
import sys
import time


def main():
    repeats, threads, budget = int(sys.argv[1]), int(sys.argv[2]), float(sys.argv[3])
    L, Tv = (int(sys.argv[4]), int(sys.argv[5])) if len(sys.argv) > 5 else (32000, 50)
    for v in ("OMP_NUM_THREADS", "OPENBLAS_NUM_THREADS", "MKL_NUM_THREADS"):
        os.environ[v] = str(threads)
    import torch
    torch.set_num_threads(threads)
    from oracle import rtfs_oracle as O
    from oracle import torch_cpu as TC
    from oracle.params import load_spec, make_inputs, make_state_dict
    sd = make_state_dict(load_spec("state_spec_R4.json"), 0)
    legs = []
    for name, fwd in (("torch_cpu", TC.avnet_forward), ("numpy", O.avnet_forward)):
        for B in (1, 4):
            wav, emb = make_inputs(B, L, Tv, 7 + B)
            t_leg, warm, times = time.perf_counter(), 0, []
            while len(times) < 5:
                spent = time.perf_counter() - t_leg
                if name != "torch_cpu" and spent > budget and warm >= 1 and len(times) >= 2:
                    break
                t0 = time.perf_counter()
                out = fwd(wav, emb, sd, repeats=repeats)
                dt = time.perf_counter() - t0
                assert out.shape == (B, 1, L)
                # warm-ups: 3 when the budget allows it, fewer when one forward already eats a third of the leg
                if warm < 3 and (name == "torch_cpu" or warm == 0 or (time.perf_counter() - t_leg) < budget / 3):
                    warm += 1
                else:
                    times.append(dt)
            legs.append({"impl": name, "batch": B, "warmup": warm, "timed": len(times), "ms_per_forward": round(1e3 * sum(times) / len(times), 1),
                         "mixtures_per_s": round(B * len(times) / sum(times), 4)})
    print(json.dumps({"threads": threads, "host_cpus": os.cpu_count(), "legs": legs}), flush=True)


if __name__ == "__main__":
    main()
