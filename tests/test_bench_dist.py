"""CPU, world_size 2 over gloo: the N > 1 leg of bench.py (per-rank shards, barrier, max-over-ranks timing, whole-job
throughput).  The forward itself has no collective (utterances are independent), so this is all the multi-GPU logic there is."""
import json
import os
import socket
import subprocess
import sys

from tests.util import ROOT

WORKER = r"""
import json, os, sys, torch
import torch.distributed as dist
sys.path.insert(0, os.environ["RTFS_ROOT"])
import bench
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", rank=rank, world_size=world)
wav, emb = bench.rank_inputs(rank, 2, 4096, 7)
dist.barrier()
dt = bench.max_over_ranks(0.5 + rank, dist, torch.device("cpu"))  # rank 1 is the slow one
gathered = [torch.zeros(1) for _ in range(world)]
dist.all_gather(gathered, wav[0, :1].clone())
dist.barrier()
print("RESULT " + json.dumps({"rank": rank, "dt": dt, "wav": list(wav.shape), "emb": list(emb.shape), "g": [float(g) for g in gathered]}), flush=True)
dist.destroy_process_group()
"""


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_two_rank_gloo_harness():
    world, port = 2, _free_port()
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(world), LOCAL_RANK=str(r), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RTFS_ROOT=ROOT)
        procs.append(subprocess.Popen([sys.executable, "-c", WORKER], env=env, cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    res = []
    for p in procs:
        out, err = p.communicate(timeout=180)
        assert p.returncode == 0, err[-2000:]
        res.append(json.loads([l for l in out.splitlines() if l.startswith("RESULT ")][0][7:]))
    res.sort(key=lambda d: d["rank"])
    assert res[0]["dt"] == res[1]["dt"] == 1.5  # every rank reports the slowest rank's time
    assert res[0]["wav"] == [2, 4096] and res[0]["emb"] == [2, 512, 7]
    assert res[0]["g"] == res[1]["g"] and res[0]["g"][0] != res[0]["g"][1]  # ranks hold different shards (different seeds)
    import bench
    assert bench.throughput(2, 32, 10, 1.5) == 2 * 32 * 10 / 1.5  # whole-job aggregate, weak scaling
    assert bench.sweep_bytes(64, 4000) == 20.0 * 57 * 4000 * 64 * 4


GRAD_WORKER = r"""
import json, os, sys, torch
import torch.distributed as dist
sys.path.insert(0, os.environ["RTFS_ROOT"])
import rtfs_net_amd as R
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", rank=rank, world_size=world)
torch.manual_seed(0)
net = torch.nn.Sequential(torch.nn.Linear(5, 3), torch.nn.PReLU(), torch.nn.Linear(3, 2))   # stand-in for the audio model's parameters
net[2].bias.requires_grad_(False)                                                           # a frozen parameter is left alone
opt = torch.optim.SGD([p for p in net.parameters() if p.requires_grad], lr=0.1)
system = R.System(audio_model=net, optimizer=opt)
for i, p in enumerate(system.trainable_parameters()):
    p.grad = torch.full_like(p, float((rank + 1) * (i + 1)))
system.trainable_parameters()[1].grad = None                                                # a parameter that got no gradient on this rank
n = system.allreduce_gradients()
g = [float(p.grad.reshape(-1)[0]) for p in system.trainable_parameters()]
print("RESULT " + json.dumps({"rank": rank, "n": n, "g": g, "frozen": net[2].bias.grad is None}), flush=True)
dist.destroy_process_group()
"""


def test_gradient_allreduce_two_ranks_gloo():
    """System.allreduce_gradients: one flattened all-reduce, mean over ranks, missing gradients count as zero, frozen parameters untouched."""
    world, port = 2, _free_port()
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(world), LOCAL_RANK=str(r), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RTFS_ROOT=ROOT)
        procs.append(subprocess.Popen([sys.executable, "-c", GRAD_WORKER], env=env, cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    res = []
    for p in procs:
        out, err = p.communicate(timeout=180)
        assert p.returncode == 0, err[-2000:]
        res.append(json.loads([l for l in out.splitlines() if l.startswith("RESULT ")][0][7:]))
    res.sort(key=lambda d: d["rank"])
    assert res[0]["n"] == res[1]["n"] == 5 * 3 + 3 + 1 + 3 * 2  # every trainable float in ONE buffer
    # parameter i carried (rank + 1) * (i + 1): mean over ranks 1.5 * (i + 1); parameter 1 had no gradient anywhere -> 0
    assert res[0]["g"] == res[1]["g"] == [1.5, 0.0, 4.5, 6.0]
    assert res[0]["frozen"] and res[1]["frozen"]
