#!/usr/bin/env python3
"""Probe (GPU box): does the forward gain from two half batches on two streams (HBM-bound kernels of one half under the latency-bound
sweeps of the other) compared with one batch of 32 on one stream?  Prints ms per 32 mixtures for both arrangements."""
import sys, os, time, copy
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import rtfs_net_amd as R
from rtfs_net_amd.configs import RTFS4_AUDIONET
torch.manual_seed(0)
m = R.AVNet(print_macs=False, **copy.deepcopy(RTFS4_AUDIONET)).cuda().eval()
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
nsplit = int(sys.argv[2]) if len(sys.argv) > 2 else 2
wav = torch.randn(B, 32000, device="cuda")
emb = torch.randn(B, 512, 50, device="cuda")
parts = [(wav[i::nsplit].contiguous(), emb[i::nsplit].contiguous()) for i in range(nsplit)]
streams = [torch.cuda.Stream() for _ in range(nsplit)]
N = 15
with torch.no_grad():
    for _ in range(3):
        m(wav, emb)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(N):
        m(wav, emb)
    torch.cuda.synchronize()
    one = (time.perf_counter() - t0) / N * 1e3
    for _ in range(3):
        for s, (w, e) in zip(streams, parts):
            with torch.cuda.stream(s):
                m(w, e)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(N):
        for s, (w, e) in zip(streams, parts):
            with torch.cuda.stream(s):
                m(w, e)
    torch.cuda.synchronize()
    two = (time.perf_counter() - t0) / N * 1e3
    # the same sub-batches one after the other on ONE stream
    t0 = time.perf_counter()
    for _ in range(N):
        for (w, e) in parts:
            m(w, e)
    torch.cuda.synchronize()
    seq = (time.perf_counter() - t0) / N * 1e3
print(f"batch {B}: one stream {one:.2f} ms | {nsplit} sub-batches on {nsplit} streams {two:.2f} ms | {nsplit} sub-batches in sequence {seq:.2f} ms")
