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
# DDP's construction-time broadcast: ranks that start from different values end on rank 0's, parameters and buffers alike
bn = torch.nn.BatchNorm1d(3)
net2 = torch.nn.Sequential(torch.nn.Linear(5, 3), bn)
with torch.no_grad():
    for t in list(net2.parameters()) + [bn.running_mean, bn.running_var]:
        t.fill_(float(rank + 2))
    bn.num_batches_tracked.fill_(rank + 7)
sys2 = R.System(audio_model=net2)
nb = sys2.broadcast_parameters()
after = sorted({float(t.reshape(-1)[0]) for t in list(net2.parameters()) + list(net2.buffers())})
print("RESULT " + json.dumps({"rank": rank, "n": n, "g": g, "frozen": net[2].bias.grad is None, "nb": nb, "after": after}), flush=True)
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
    # broadcast: 5*3 + 3 + 3 + 3 + 3 + 3 floats and one integer buffer, every value now rank 0's (2.0; num_batches_tracked 7)
    assert res[0]["nb"] == res[1]["nb"] == 15 + 3 + 3 + 3 + 3 + 3 + 1
    assert res[0]["after"] == res[1]["after"] == [2.0, 7.0]


def _bench(*argv, env=None):
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *argv], cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                          text=True, timeout=300)


def test_bench_self_launches_its_ranks():
    """`python bench.py --gpus 2 ...` exactly as the driver types it (no torch.distributed.run, no WORLD_SIZE): the parent spawns the ranks,
    relays rank 0's ONE JSON line and exits 0.  --dist-dry-run swaps the forward for a stand-in step over gloo so the launcher, the
    rendezvous and the timing protocol run end to end where there is no GPU."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    pr = _bench("--gpus", "2", "--steps", "3", "--warmup", "1", "--dist-dry-run", env=env)
    assert pr.returncode == 0, pr.stderr[-2000:]
    lines = [l for l in pr.stdout.splitlines() if l.startswith("{")]  # (gloo itself prints a "[Gloo] Rank 0 is connected ..." line)
    assert len(lines) == 1, pr.stdout
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 2 and rec["steps"] == 3 and rec["warmup"] == 1 and rec["config"]["global_batch"] == 64
    assert rec["value"] == round(2 * 32 * 3 / (rec["ms_per_step"] * 3e-3), 3) or abs(rec["value"] * rec["ms_per_step"] * 1e-3 - 64) < 0.05
    # the line carries its own evidence of what the process group saw: backend, world size, one identity per rank gathered over that group
    d = rec["dist"]
    assert d["backend"] == "gloo" and d["world"] == 2 and len(d["devices"]) == 2 and d["distinct_devices"] == 2
    assert sorted(x["rank"] for x in d["devices"]) == [0, 1]
    # the same through torch.distributed.run (the driver's other launch form)
    port = _free_port()
    pr = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                         "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--dist-dry-run"],
                        cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=300)
    assert pr.returncode == 0, pr.stderr[-2000:]
    recs = [json.loads(l) for l in pr.stdout.splitlines() if l.startswith("{")]
    assert len(recs) == 1 and recs[0]["n_gpus"] == 2 and recs[0]["dist"]["world"] == 2 and recs[0]["dist"]["distinct_devices"] == 2


def test_bench_launcher_reports_a_failed_rank():
    """A rank that dies must fail the whole job (non-zero exit, no JSON line), not leave the others at a barrier: here --gpus disagrees
    with a pre-set WORLD_SIZE inside the children (the mismatch check), and the training leg is covered by the same launcher."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["WORLD_SIZE"] = "3"
    pr = _bench("--gpus", "2", "--steps", "1", "--warmup", "0", "--dist-dry-run", env=env)
    assert pr.returncode != 0 and "{" not in pr.stdout
    env.pop("WORLD_SIZE")
    env["RTFS_BENCH_FAIL_RANK"] = "1"
    pr = _bench("--gpus", "2", "--steps", "1", "--warmup", "0", "--dist-dry-run", env=env)
    assert pr.returncode != 0 and "{" not in pr.stdout and "rank 1 failed" in pr.stderr


import pytest  # noqa: E402


@pytest.mark.gpu
def test_bench_self_launch_with_the_real_forward_on_one_gpu():
    """The N-rank protocol with the REAL forward: `python bench.py --gpus 2` spawns its two ranks; RTFS_BENCH_SHARE_GPU=1 lets both use cuda:0
    (time-sliced) with a gloo process group, because a test box has one GPU and RCCL wants a device per rank.  Checks the launcher, the
    per-rank forward, the max-over-ranks timing and rank 0's single JSON line (roofline object included) - not a scaling number."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env["RTFS_BENCH_SHARE_GPU"] = "1"
    pr = _bench("--gpus", "2", "--steps", "2", "--warmup", "1", "--batch", "2", env=env)
    assert pr.returncode == 0, pr.stderr[-2000:]
    recs = [json.loads(l) for l in pr.stdout.splitlines() if l.startswith("{")]
    assert len(recs) == 1
    r = recs[0]
    assert r["n_gpus"] == 2 and r["config"]["global_batch"] == 4 and r["value"] > 0 and "cpu_baseline" not in r
    # 2 steps x 8 sweep launches, every third one timed (bench.py: rtfs_sweep_timing_enable(3))
    assert r["roofline"]["launches_timed"] == len(range(0, 2 * 8, 3)) and 0 < r["roofline"]["frac"] < 2
