"""A minimal data-parallel training loop on the HIP path - what the reference's train.py (135-160) asks of Lightning, spelled out:
AVNet from the unchanged yaml section, AdamW, SyncBatchNorm, System.optimization_step (forward_train -> PIT loss -> HIP backward ->
one flattened gradient all-reduce -> clip 5.0 -> step), best_model.pth written in the reference's format.  Data are synthetic mixtures
(no dataset in this repo): the point is the step, not the recipe.

1 GPU :  python tools/train_synthetic.py --steps 20
N GPUs:  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port 29511 tools/train_synthetic.py
"""
import argparse
import copy
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.distributed as dist
import rtfs_net_amd as R


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--batch", type=int, default=4, help="per GPU (yaml training.batch_size)")
    ap.add_argument("--repeats", type=int, default=4)
    ap.add_argument("--seconds", type=float, default=2.0)
    ap.add_argument("--lr", type=float, default=1e-3)
    ap.add_argument("--out", default="")
    a = ap.parse_args()
    world, rank, local = int(os.environ.get("WORLD_SIZE", 1)), int(os.environ.get("RANK", 0)), int(os.environ.get("LOCAL_RANK", 0))
    torch.cuda.set_device(local)
    if world > 1:
        dist.init_process_group("nccl", rank=rank, world_size=world)  # RCCL
    from rtfs_net_amd.configs import RTFS4_AUDIONET
    conf = copy.deepcopy(RTFS4_AUDIONET)
    conf["audio_params"]["repeats"] = a.repeats
    torch.manual_seed(0)  # identical initial weights on every rank
    model = R.AVNet(print_macs=False, **conf).cuda().train()
    loss = R.losses.PITLossWrapper(R.losses.PairwiseNegSDR("snr"), pit_from="pw_mtx")
    opt = torch.optim.AdamW(model.parameters(), lr=a.lr, weight_decay=0.1)
    system = R.System(audio_model=model, loss_func={"train": loss, "val": loss}, optimizer=opt)
    if world > 1:
        system.convert_sync_batchnorm()  # train.py:145 sync_batchnorm=True
    g = torch.Generator().manual_seed(1234 + rank)  # every rank its own shard
    L, Tv = int(16000 * a.seconds), int(25 * a.seconds)
    t0 = time.time()
    for step in range(a.steps):
        s1, s2 = 0.05 * torch.randn(a.batch, L, generator=g), 0.05 * torch.randn(a.batch, L, generator=g)
        batch = ((s1 + s2).cuda(), s1.unsqueeze(1).cuda(), torch.randn(a.batch, 512, Tv, generator=g).cuda(), None)
        value = system.optimization_step(batch, step)
        if rank == 0 and (step % 5 == 0 or step == a.steps - 1):
            torch.cuda.synchronize()
            print(f"step {step:4d}  loss {float(value):8.4f}  {(time.time() - t0) / (step + 1) * 1e3:7.1f} ms/step  world {world}", flush=True)
    if rank == 0 and a.out:
        torch.save(system.audio_model.serialize(), a.out)  # best_model.pth format (base_av_model.py:36-51)
        print("wrote", a.out)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
