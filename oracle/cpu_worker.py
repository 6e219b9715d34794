"""One CPU-baseline worker (TEST / MEASUREMENT INFRASTRUCTURE ONLY): runs the numpy oracle forward on `n` synthetic
2 s mixtures with a single BLAS thread and prints one JSON line.  bench.py starts one worker per host core of the GPU box's
CPU share, before it touches the GPU, and reports the aggregate rate as `cpu_baseline`.

    python -m oracle.cpu_worker <repeats> <n_mixtures> <seed>
"""
import json
import os
import sys
import time

for _v in ("OMP_NUM_THREADS", "OPENBLAS_NUM_THREADS", "MKL_NUM_THREADS"):
    os.environ[_v] = "1"


def main():
    repeats, n, seed = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
    from oracle import rtfs_oracle as O
    from oracle.params import load_spec, make_inputs, make_state_dict
    sd = make_state_dict(load_spec("state_spec_R4.json"), 0)
    t0 = time.perf_counter()
    for i in range(n):
        wav, emb = make_inputs(1, 32000, 50, seed * 1000 + i)
        out = O.avnet_forward(wav, emb, sd, repeats=repeats)
        assert out.shape[0] == 1 and out.shape[-1] == 32000
    print(json.dumps({"n": n, "seconds": time.perf_counter() - t0}), flush=True)


if __name__ == "__main__":
    main()
