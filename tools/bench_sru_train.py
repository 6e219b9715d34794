"""Times the training-side SRU operator (forward with saved state, backward) at the path's sweep shapes.
python tools/bench_sru_train.py [--batch 32]   (GPU box)"""
import argparse
import sys, os

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import rtfs_net_amd as R


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--iters", type=int, default=10)
    a = ap.parse_args()
    torch.manual_seed(0)
    sru = R.layers.SRU(512, 32, num_layers=4, bidirectional=True).cuda().train()
    for name, L, N in (("F-sweep", 57, 125 * a.batch), ("T-sweep", 118, 64 * a.batch)):
        x = torch.randn(L, N, 512, device="cuda", requires_grad=True)
        dh = torch.randn(L, N, 64, device="cuda")
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
        tf = tb = 0.0
        for it in range(a.iters + 2):
            ev[0].record()
            h, _ = sru(x)
            ev[1].record()
            h.backward(dh)
            ev[2].record()
            torch.cuda.synchronize()
            if it >= 2:
                tf += ev[0].elapsed_time(ev[1])
                tb += ev[1].elapsed_time(ev[2])
        gflop_f = 2.0 * L * N * (512 * 256 + 3 * 64 * 192) / 1e9
        print(f"{name}: L {L} N {N}: forward {tf / a.iters:.3f} ms ({gflop_f / (tf / a.iters):.1f} TFLOP/s eff), "
              f"backward {tb / a.iters:.3f} ms ({2 * gflop_f / (tb / a.iters):.1f} TFLOP/s eff)")


if __name__ == "__main__":
    main()
