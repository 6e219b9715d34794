"""Diagnostic 2: repeatability of one RTFS block at full size, and its TFAR sub-modules batch vs alone.  (GPU box)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from oracle.params import make_state_dict
from tests.util import spec_R4, rel_err
import rtfs_net_amd as R
from rtfs_net_amd.configs import audionet_config

sd = make_state_dict(spec_R4(), 0)
m = R.AVNet(print_macs=False, **audionet_config(4))
m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
m = m.cuda().eval()
blk = m.refinement_module.audio_net.get_block(0)
g = torch.Generator(device="cuda").manual_seed(3)
T, F = 251, 129


def cmp(a, b):
    a, b = a.float().cpu().numpy(), b.float().cpu().numpy()
    d = np.abs(a - b)
    return f"max-rel {d.max() / np.abs(b).max():.2e} nbad {(d > 1e-5 * np.abs(b).max()).sum()}"


with torch.no_grad():
    for B in (4, 8, 32):
        x = torch.randn(B, 256, T, F, device="cuda", generator=g)
        ones = [blk(x[i:i + 1].contiguous()) for i in range(min(B, 4))]
        for rep in range(4):
            y = blk(x)
            torch.cuda.synchronize()
            print(f"B={B} rep {rep}: " + " | ".join(cmp(y[i:i + 1], ones[i]) for i in range(min(B, 4))))
            bad = (y[0] - ones[0][0]).abs() > 1e-4 * ones[0].abs().max()
            if bad.any():
                idx = bad.nonzero()
                print("   bad elements of mixture 0: n", idx.shape[0], "c range", idx[:, 0].min().item(), idx[:, 0].max().item(),
                      "t range", idx[:, 1].min().item(), idx[:, 1].max().item(), "f range", idx[:, 2].min().item(), idx[:, 2].max().item())
        loc = torch.randn(B, 64, T, F, device="cuda", generator=g)
        glo = torch.randn(B, 64, T // 2, F // 2, device="cuda", generator=g)
        for nm, mod, l in (("fusion0 (up)", blk.fusion_layers[0], loc), ("fusion1 (same)", blk.fusion_layers[1], glo), ("concat0 (up)", blk.concat_layers[0], loc)):
            yb = mod(l, glo)
            y1 = mod(l[:1].contiguous(), glo[:1].contiguous())
            torch.cuda.synchronize()
            print(f"B={B} {nm}: {cmp(yb[:1], y1)}")
        for i, ds in enumerate(blk.downsample_layers):
            xin = loc
            yb = ds(xin); y1 = ds(xin[:1].contiguous())
            torch.cuda.synchronize()
            print(f"B={B} downsample[{i}]: {cmp(yb[:1], y1)}")
