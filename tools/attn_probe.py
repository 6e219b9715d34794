#!/usr/bin/env python3
"""Runs the TF-attention entry point alone at the bench shape (for rocprofv3 kernel timing of row_can / attn_core)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import rtfs_net_amd as R
from rtfs_net_amd import layers

B, T = int(os.environ.get("B", 32)), int(os.environ.get("T", 125))
torch.manual_seed(0)
m = layers.MultiHeadSelfAttention2D(64, 64).cuda().eval()
x = torch.randn(B, 64, T, 64, device="cuda")
with torch.no_grad():
    for _ in range(6):
        y = m(x)
torch.cuda.synchronize()
print("ok", float(y.abs().mean()))
