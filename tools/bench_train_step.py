"""One training step of RTFS-Net-R on the HIP path (AVNet.forward_train + PIT loss + backward + AdamW), at the reference's
training shape: batch 4 per GPU, 2 s segments (config/lrs2_RTFSNet_4_layer.yaml: batch_size 4, segment 2.0).
python tools/bench_train_step.py [--batch 4] [--repeats 4] [--iters 5]   (GPU box)"""
import argparse
import copy
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import rtfs_net_amd as R


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=4)
    ap.add_argument("--repeats", type=int, default=4)
    ap.add_argument("--iters", type=int, default=5)
    ap.add_argument("--seconds", type=float, default=2.0)
    ap.add_argument("--full", action="store_true", help="train everything (VP block with dropout, BatchNorm on batch statistics) instead of "
                    "the fine-tuning configuration (frozen BatchNorm statistics and VP block)")
    a = ap.parse_args()
    from rtfs_net_amd.configs import RTFS4_AUDIONET
    conf = copy.deepcopy(RTFS4_AUDIONET)
    conf["audio_params"]["repeats"] = a.repeats
    torch.manual_seed(0)
    m = R.AVNet(print_macs=False, **conf).cuda()
    m = m.train() if a.full else m.freeze_for_finetune()
    loss_mod = R.losses.PITLossWrapper(R.losses.PairwiseNegSDR("snr"), pit_from="pw_mtx")
    opt = torch.optim.AdamW([p for p in m.parameters() if p.requires_grad], lr=1e-3)
    system = R.System(audio_model=m, loss_func={"train": loss_mod, "val": loss_mod}, optimizer=opt)
    g = torch.Generator().manual_seed(1234)
    L = int(16000 * a.seconds)
    s1, s2 = 0.05 * torch.randn(a.batch, L, generator=g), 0.05 * torch.randn(a.batch, L, generator=g)
    wav, tgt, emb = (s1 + s2).cuda(), s1.cuda(), torch.randn(a.batch, 512, int(25 * a.seconds), generator=g).cuda()
    batch = (wav, tgt, emb, None)
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    losses = []
    for it in range(a.iters + 2):
        if it == 2:
            torch.cuda.synchronize()
            ev[0].record()
        losses.append(float(system.optimization_step(batch)))
    ev[1].record()
    torch.cuda.synchronize()
    ms = ev[0].elapsed_time(ev[1]) / a.iters
    print(f"RTFS-Net-{a.repeats} {'full' if a.full else 'fine-tune'} training step, batch {a.batch} x {a.seconds:g} s: {ms:.1f} ms/step = {a.batch / ms * 1e3:.1f} mixtures/s trained; "
          f"peak memory {torch.cuda.max_memory_allocated() / 2**30:.2f} GiB; loss {losses[0]:.3f} -> {losses[-1]:.3f}")


if __name__ == "__main__":
    main()
