"""Times one RTFS block inside a training step (forward with saved state + backward, all HIP) at the reference's training
shape (batch 4 per GPU, 2 s: (B, 256, 251, 129)), next to the fused inference forward.
python tools/bench_block_train.py [--batch 4] [--iters 5]   (GPU box)"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import yaml
import rtfs_net_amd as R


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=4)
    ap.add_argument("--iters", type=int, default=5)
    a = ap.parse_args()
    from rtfs_net_amd.configs import RTFS4_AUDIONET
    torch.manual_seed(0)
    m = R.AVNet(print_macs=False, **RTFS4_AUDIONET).cuda()
    blk = m.refinement_module.audio_net.get_block(0)
    x = torch.randn(a.batch, 256, 251, 129, device="cuda", requires_grad=True)
    dout = torch.randn_like(x)
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
    tf = tb = 0.0
    blk.train()
    for it in range(a.iters + 1):
        for p in blk.parameters():
            p.grad = None
        ev[0].record()
        out = blk(x)
        ev[1].record()
        out.backward(dout)
        ev[2].record()
        torch.cuda.synchronize()
        if it >= 1:
            tf += ev[0].elapsed_time(ev[1])
            tb += ev[1].elapsed_time(ev[2])
    blk.eval()
    with torch.no_grad():
        for _ in range(2):
            blk(x)
        ev[0].record()
        for _ in range(a.iters):
            blk(x)
        ev[1].record()
        torch.cuda.synchronize()
        ti = ev[0].elapsed_time(ev[1]) / a.iters
    print(f"RTFS block, batch {a.batch}, (256, 251, 129): training forward {tf / a.iters:.2f} ms, backward {tb / a.iters:.2f} ms, "
          f"inference forward {ti:.2f} ms; peak memory {torch.cuda.max_memory_allocated() / 2**30:.2f} GiB")


if __name__ == "__main__":
    main()
