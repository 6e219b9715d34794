"""Diagnostic: where does mixture 0 of a large batch start to differ from its batch-1 run?  (GPU box)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from oracle.params import make_inputs, make_state_dict
from tests.util import spec_R4, load_golden, rel_err
import rtfs_net_amd as R
from rtfs_net_amd.configs import audionet_config

sd = make_state_dict(spec_R4(), 0)
m = R.AVNet(print_macs=False, **audionet_config(4))
m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
m = m.cuda().eval()
L, Tv = 32000, 50
wav0, emb0 = make_inputs(1, L, Tv, 2)
gold = load_golden("e2e_R4_L32000_B1")["out"]
dev = lambda x: torch.from_numpy(np.ascontiguousarray(x)).cuda()


def stages(wav, emb):
    with torch.no_grad():
        a0, st = m.encoder(wav, return_stats=True)
        a1 = m.audio_bottleneck(a0, st)
        rm = m.refinement_module
        blk = rm.audio_net.get_block(0)
        vp = rm.video_net.get_block(0)(m.video_bottleneck(emb))
        b0 = blk(a1)
        caf, _ = rm.crossmodal_fusion.get_fusion_block(0)(b0, vp)
        b1 = blk(caf, a1)
        b2 = blk(b1, a1)
        b3 = blk(b2, a1)
        sep = m.mask_generator(b3, a0)
        out = m.decoder(sep, wav.shape)
    torch.cuda.synchronize()
    return dict(a0=a0, a1=a1, vp=vp, b0=b0, caf=caf, b1=b1, b2=b2, b3=b3, sep=sep, out=out)


ref = {k: v[:1].cpu().numpy() for k, v in stages(dev(wav0), dev(emb0)).items()}
print("B=1 modular vs golden", rel_err(ref["out"], gold))
for B in (2, 4, 8, 16, 32):
    wr, er = make_inputs(B - 1, L, Tv, 1002)
    wav, emb = dev(np.concatenate([wav0, wr])), dev(np.concatenate([emb0, er]))
    with torch.no_grad():
        of = m(wav, emb)
    torch.cuda.synchronize()
    print(f"B={B} fused   out[0] vs golden {rel_err(of[:1].cpu().numpy(), gold):.3e}")
    st = stages(wav, emb)
    print(f"B={B} modular " + "  ".join(f"{k} {rel_err(v[:1].cpu().numpy(), ref[k]):.2e}" for k, v in st.items()))
    if B == 32:
        # isolate inside the block at B=32: each sub-module on the batch vs on mixture 0 alone
        blk = m.refinement_module.audio_net.get_block(0)
        x = st["a1"]
        with torch.no_grad():
            for i, g in enumerate(blk.globalatt):
                z = torch.randn(B, 64, 125, 64, device="cuda", generator=torch.Generator(device="cuda").manual_seed(5))
                yb = g(z); y1 = g(z[:1].contiguous())
                torch.cuda.synchronize()
                print(f"  globalatt[{i}] batch vs alone {rel_err(yb[:1].cpu().numpy(), y1.cpu().numpy()):.2e}   last {rel_err(yb[-1:].cpu().numpy(), g(z[-1:].contiguous()).cpu().numpy()):.2e}")
