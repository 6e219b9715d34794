#!/usr/bin/env python3
"""Throughput of the video front-end alone (B=32 clips x 50 frames of 88x88, the lip stream that accompanies the bench's
2 s mixtures).  Not part of bench.py's contract; prints frames/s and clips/s."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import rtfs_net_amd as R

B, T = int(os.environ.get("B", 32)), int(os.environ.get("T", 50))
torch.manual_seed(0)
m = R.FRCNNVideoModel(print_macs=False).cuda().eval()
x = torch.rand(B, 1, T, 88, 88, device="cuda")
with torch.no_grad():
    for _ in range(3):
        y = m(x)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 10
    for _ in range(n):
        y = m(x)
    torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / n
print(f"video front-end: {dt * 1e3:.2f} ms per batch of {B} clips x {T} frames -> {B * T / dt:.0f} frames/s, {B / dt:.0f} clips/s; out {tuple(y.shape)}")
