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


def dualpath(batch, iters):
    """DualPathRNN module (LN + unfold windows + SRU + ConvTranspose1d + residual) at the block's G-level shape."""
    for name, dim in (("F-sweep", 4), ("T-sweep", 3)):
        mod = R.layers.DualPathRNN(64, 32, dim, kernel_size=8, stride=1, rnn_type="SRU", num_layers=4, bidirectional=True).cuda().train()
        x = torch.randn(batch, 64, 125, 64, device="cuda", requires_grad=True)
        dout = torch.randn_like(x)
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
        tf = tb = 0.0
        for it in range(iters + 2):
            ev[0].record()
            out = mod(x)
            ev[1].record()
            out.backward(dout)
            ev[2].record()
            torch.cuda.synchronize()
            if it >= 2:
                tf += ev[0].elapsed_time(ev[1])
                tb += ev[1].elapsed_time(ev[2])
        with torch.no_grad():
            mod.eval()
            for _ in range(3):
                mod(x)
            ev[0].record()
            for _ in range(iters):
                mod(x)
            ev[1].record()
            torch.cuda.synchronize()
            ti = ev[0].elapsed_time(ev[1]) / iters
        print(f"DualPathRNN {name} (B {batch}): training forward {tf / iters:.3f} ms, backward {tb / iters:.3f} ms, inference forward {ti:.3f} ms")


if __name__ == "__main__":
    main()
    dualpath(32, 10)
