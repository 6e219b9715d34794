"""Which host-side line launches the small torch kernels of a training step?  One profiled step (torch.profiler, Python stacks) of the
`bench.py --train` workload; prints, per aten op of interest, the call sites by launch count.

    python tools/profile_train_ops.py [--ops aten::fill_,aten::add_,...]
"""
import argparse
import collections
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ops", default="aten::fill_,aten::zero_,aten::add_,aten::add,aten::cat,aten::copy_,aten::mul")
    ap.add_argument("--depth", type=int, default=3)
    args = ap.parse_args()
    import rtfs_net_amd as R
    from bench import audionet_config, rank_inputs
    dev = torch.device("cuda", 0)
    torch.manual_seed(0)
    model = R.AVNet(print_macs=False, **audionet_config(4)).to(dev).train()
    loss = R.losses.PITLossWrapper(R.losses.PairwiseNegSDR("snr"), pit_from="pw_mtx")
    opt = torch.optim.AdamW(model.parameters(), lr=1e-3, weight_decay=0.1)
    system = R.System(audio_model=model, loss_func={"train": loss, "val": loss}, optimizer=opt)
    wav, emb = rank_inputs(0, 4, 32000, 50)
    tgt = wav - 0.05 * torch.randn(4, 32000)
    batch = (wav.to(dev), tgt.unsqueeze(1).to(dev), emb.to(dev), None)
    for _ in range(2):
        system.optimization_step(batch)
    torch.cuda.synchronize()
    from torch.profiler import ProfilerActivity, profile
    with profile(activities=[ProfilerActivity.CPU], with_stack=True) as prof:
        system.optimization_step(batch)
        torch.cuda.synchronize()
    want = set(args.ops.split(","))
    sites = {op: collections.Counter() for op in want}
    for ev in prof.events():
        if ev.name in want:
            frames = [f for f in ev.stack if ROOT in f or "torch/optim" in f or "autograd" in f][: args.depth]
            sites[ev.name][" <- ".join(f.replace(ROOT + "/", "") for f in frames) or "(no python frame: autograd engine thread)"] += 1
    for op in sorted(want):
        print(f"== {op}: {sum(sites[op].values())} calls")
        for site, n in sites[op].most_common(14):
            print(f"   {n:5d}  {site}")


if __name__ == "__main__":
    main()
