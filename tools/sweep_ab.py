"""A/B timing of the dual-path sweep alone at the bench shape (B = 32, 2 s): median of 30 launches per sweep, HIP events.  (GPU box)
Environment knobs are read by the library at first use, so every configuration runs in its own process:  RTFS_SWEEP_STAGGER=n python tools/sweep_ab.py"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import rtfs_net_amd as R
from rtfs_net_amd.configs import audionet_config
torch.manual_seed(0)
m = R.AVNet(print_macs=False, **audionet_config(4)).cuda().eval()
blk = m.refinement_module.audio_net.blocks
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
x = torch.randn(B, 64, 125, 64, device="cuda")
res = []
with torch.no_grad():
    for name, mod, nbytes in (("F", blk.globalatt[0], 20.0 * 57 * B * 125 * 64 * 4 * 4), ("T", blk.globalatt[1], 20.0 * 118 * B * 64 * 64 * 4 * 4)):
        for _ in range(5):
            mod(x)
        ts = []
        for _ in range(30):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); mod(x); e1.record(); torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1))
        res.append((name, float(np.median(ts)), nbytes))
tot = sum(t for _, t, _ in res)
print(os.environ.get("RTFS_SWEEP_STAGGER", "default"), os.environ.get("RTFS_SWEEP_GEN2", ""), " ".join(f"{n} {t*1e3:.0f} us" for n, t, _ in res),
      f"| module-level (T incl. 2 transposes): roofline frac over both sweeps {sum(b for *_, b in res) / (tot * 1e-3) / 8e12:.3f}")
