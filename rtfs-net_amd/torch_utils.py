"""ROCm replacement for the reference's GPU polling (``src/utils/torch_utils.py:53-75``; ``train.py:61-68`` busy-waits on it until
as many GPUs are free as the yaml's ``training.gpus`` asks for).  The reference shells out to ``nvidia-smi`` twice (memory listing for
the bus ids, ``--query-compute-apps`` for the busy ones); an MI355X box has ``amd-smi``: one ``amd-smi process --json`` call lists every
GPU with its compute processes.  Same contract: ``get_free_gpu_indices() -> [int]`` = indices of the GPUs with no compute process.
"""
from __future__ import annotations

import json
import subprocess


def run_cmd(cmd):
    """torch_utils.py:53-55."""
    return subprocess.check_output(cmd, shell=True).decode("utf-8")[:-1]


def _gpu_records(doc):
    """The per-GPU records of an ``amd-smi ... --json`` document: a top-level list, or (newer releases) a dict holding that list."""
    if isinstance(doc, list):
        return [d for d in doc if isinstance(d, dict) and "gpu" in d]
    if isinstance(doc, dict):
        if "gpu" in doc:
            return [doc]
        for v in doc.values():
            recs = _gpu_records(v)
            if recs:
                return recs
    return []


def parse_amd_smi_process(text):
    """``amd-smi process --json`` -> (all GPU indices, indices with at least one compute process).  A GPU without processes carries
    one placeholder entry whose ``process_info`` is the string "No running processes detected" (or "N/A")."""
    start = min([i for i in (text.find("["), text.find("{")) if i >= 0], default=-1)
    if start < 0:
        raise ValueError("amd-smi printed no JSON document")
    recs = _gpu_records(json.loads(text[start:]))
    if not recs:
        raise ValueError("amd-smi JSON holds no per-GPU records")
    all_ids, busy = [], []
    for rec in recs:
        gpu = int(rec["gpu"])
        all_ids.append(gpu)
        procs = rec.get("process_list", [])
        if isinstance(procs, dict):
            procs = [procs]
        live = [p for p in procs if isinstance(p, dict) and isinstance(p.get("process_info", p), dict)]
        if live:
            busy.append(gpu)
    return sorted(set(all_ids)), sorted(set(busy))


def get_free_gpu_indices(run=run_cmd):
    """torch_utils.py:58-75 on ROCm: the GPUs no compute process is using right now."""
    all_ids, busy = parse_amd_smi_process(run("amd-smi process --json"))
    return [i for i in all_ids if i not in busy]
