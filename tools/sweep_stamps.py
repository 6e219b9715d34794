#!/usr/bin/env python3
"""Diagnostic: where does a dual-path sweep workgroup spend its cycles?  (GPU box only)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import rtfs_net_amd as R
from rtfs_net_amd import _lib
from rtfs_net_amd.configs import RTFS4_AUDIONET
import copy
lib = _lib.load()
torch.manual_seed(0)
m = R.AVNet(print_macs=False, **copy.deepcopy(RTFS4_AUDIONET)).cuda().eval()
blk = m.refinement_module.audio_net.blocks
names = ["start", "load+LN", "L0 gemm", "L0 scan", "L1 gemm", "L1 scan", "L2 gemm", "L2 scan", "L3 gemm", "L3 scan", "convT gemm", "epilogue"]
if os.environ.get("RTFS_SWEEP_GEN4"):
    names = ["start", "load+LN"] + [f"L{l} {w}" for l in range(4) for w in ("gemm A", "chain", "gemm B + gates")] + ["convT gemm", "epilogue"]
GEN2 = bool(os.environ.get("RTFS_SWEEP_GEN2"))  # generation 2: 4 / 2 sequences per 512-thread workgroup; generation 3: 2 / 1 per 256-thread workgroup
for label, dp, Rr, Ls, per in [("F-path", blk.globalatt[0], 125, 64, 4 if GEN2 else 2), ("T-path", blk.globalatt[1], 64, 125, 2 if GEN2 else 1)]:
    B = int(os.environ.get('STAMP_B', '32'))
    x = torch.randn(B, 64, Rr, Ls, device="cuda")
    out = torch.empty_like(x)
    nwg = (B * Rr + per - 1) // per
    st = torch.zeros(nwg, 16, dtype=torch.int64, device="cuda")
    for _ in range(2):
        _lib.check(lib.rtfs_debug_sweep_stamps(_lib.ptr(x), _lib.ptr(dp.pack()), _lib.ptr(out), B, Rr, Ls, _lib.ptr(st), _lib.stream_of(x)), "stamps")
    torch.cuda.synchronize()
    s = st.cpu().numpy().astype(np.float64)
    NS = len(names)
    d = np.diff(s[:, :NS], axis=1)
    print(f"== {label}: {nwg} workgroups; cycles per phase (median / mean), total median {np.median(s[:,NS-1]-s[:,0]):.0f}")
    for i in range(NS - 1):
        print(f"   {names[i+1]:12s} {np.median(d[:, i]):9.0f} {d[:, i].mean():9.0f}")
    if not GEN2 and not os.environ.get("RTFS_SWEEP_GEN4"):
        sub = s[:, 12:16]
        base = s[:, 4]  # "L1 gemm" done
        print("   layer 1 scan, thread 0 (time part 0): prescale undone +%.0f | its chain +%.0f | barrier +%.0f | (loop: part 1's chain, own write-back, barrier) +%.0f | to scan done +%.0f"
              % (np.median(sub[:, 0] - base), np.median(sub[:, 1] - sub[:, 0]), np.median(sub[:, 2] - sub[:, 1]), np.median(sub[:, 3] - sub[:, 2]), np.median(s[:, 5] - sub[:, 3])))
        continue
    raw = st.cpu().numpy()[:, 12:15].astype(np.uint64)
    dl = np.stack([raw & np.uint64(0xFFFFFFFF), raw >> np.uint64(32)], -1).reshape(len(raw), 6).astype(np.float64)
    if dl.any():
        print("   inside K step 9 of layer 0, wave 0, issue-time deltas (median): B tiles 2-3 read issue %.0f | tiles 0-1 (12 MFMA) %.0f | staging %.0f | "
              "barrier %.0f | next-step reads issue %.0f | tiles 2-3 (12 MFMA) %.0f" % tuple(np.median(dl, axis=0)))
    t0 = s[:, 0].min(); t1 = s[:, NS - 1].max()
    print(f"   kernel span {t1 - t0:.0f} ticks (s_memtime 100MHz?)")
