"""Diagnostic 3: run rtfs_block_f32 on a caller-held workspace until a mixture comes out wrong, then compare every internal buffer of
that mixture (BlockWs layout, api.hip) with the same mixture's batch-1 run.  (GPU box)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from oracle.params import make_state_dict
from tests.util import spec_R4
import rtfs_net_amd as R
from rtfs_net_amd import _lib
from rtfs_net_amd.configs import audionet_config

sd = make_state_dict(spec_R4(), 0)
m = R.AVNet(print_macs=False, **audionet_config(4))
m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
m = m.cuda().eval()
blk = m.refinement_module.audio_net.get_block(0)
lib = _lib.load()
T, F = 251, 129
P, Pg = T * F, (T // 2) * (F // 2)
FULL = ["residual", "x_enc", "c0", "xf0", "expanded"]
GL = ["c1", "p0", "g", "gF", "tA", "tB", "gT", "gA", "v", "o", "E0", "G0", "E1", "G1", "L1", "xf1", "E2", "G2"]
STATS = ["S_C0", "S_C1", "S_E0", "S_G0", "S_E1", "S_G1", "S_L1", "S_E2", "S_G2", "S_L0", "S_L2"]


def layout(B):
    off, lay = 0, {}
    def take(name, nbytes):
        nonlocal off
        off = (off + 255) // 256 * 256
        lay[name] = (off, nbytes)
        off += nbytes
    take("residual", B * 256 * P * 4)
    for n in FULL[1:]:
        take(n, B * 64 * P * 4)
    for n in GL:
        take(n, B * 64 * Pg * 4)
    take("q", B * 64 * Pg)
    take("k", B * 64 * Pg)
    take("stats", 11 * B * 2 * 8)
    return lay


def run(x):
    B = x.shape[0]
    out = torch.empty_like(x)
    ws = torch.empty(lib.rtfs_block_workspace_bytes(B, T, F), dtype=torch.uint8, device="cuda")
    _lib.check(lib.rtfs_block_f32(_lib.ptr(x), None, _lib.ptr(blk.pack()), _lib.ptr(out), B, T, F, _lib.ptr(ws), ws.numel(), _lib.stream_of(x), 0), "blk")
    torch.cuda.synchronize()
    return out, ws


def view(ws, lay, name, B):
    off, nb = lay[name]
    raw = ws[off:off + nb]
    if name == "stats":
        return raw.view(torch.float64).view(11, B, 2)
    return raw.view(torch.float32).view(B, -1)


g = torch.Generator(device="cuda").manual_seed(3)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4
x = torch.randn(B, 256, T, F, device="cuda", generator=g)
refs = {}
for i in range(min(B, 2)):
    o1, w1 = run(x[i:i + 1].contiguous())
    l1 = layout(1)
    refs[i] = (o1.clone(), {n: view(w1, l1, n, 1).clone() for n in FULL + GL + ["stats"]})
lay = layout(B)
for rep in range(12):
    out, ws = run(x)
    for i in refs:
        err = float((out[i] - refs[i][0][0]).abs().max() / refs[i][0].abs().max())
        if err > 1e-5:
            print(f"rep {rep}: mixture {i} wrong by {err:.2e}; internals (max-rel vs batch-1 run):")
            for n in FULL + GL:
                a, b = view(ws, lay, n, B)[i], refs[i][1][n][0]
                print(f"   {n:9s} {float((a - b).abs().max() / b.abs().max().clamp_min(1e-30)):.2e}")
            for n, C in (("residual", 256), ("x_enc", 64)):
                a, b = view(ws, lay, n, B)[i].view(C, P), refs[i][1][n][0].view(C, P)
                bad = ((a - b).abs() > 1e-6 * b.abs().max()).nonzero()
                cs, ps = bad[:, 0], bad[:, 1]
                print(f"   {n}: {bad.shape[0]} bad elements; channels {cs.min().item()}..{cs.max().item()} ({cs.unique().numel()} distinct), pixels {ps.min().item()}..{ps.max().item()} ({ps.unique().numel()} distinct)")
                up = ps.unique()[:40].tolist()
                print("      first bad pixels:", up)
                c0 = cs[ps == ps.min()].unique()[:40].tolist()
                print("      bad channels at the first bad pixel:", c0)
                pp, cc = int(ps.min()), int(cs[ps == ps.min()].min())
                print("      got", a[cc, pp:pp + 4].tolist(), "want", b[cc, pp:pp + 4].tolist())
                # is the wrong value some other sample's / position's value?
                allb = view(ws, lay, n, B).view(B, C, P)
                print("      same (c,p) in other mixtures of the batch:", [float(allb[j, cc, pp]) for j in range(B)])
            sa, sb = view(ws, lay, "stats", B)[:, i], refs[i][1]["stats"][:, 0]
            for j, n in enumerate(STATS):
                print(f"   {n}: batch ({sa[j, 0].item():.6e}, {sa[j, 1].item():.6e})  alone ({sb[j, 0].item():.6e}, {sb[j, 1].item():.6e})")
            sys.exit(0)
    print(f"rep {rep}: clean")
