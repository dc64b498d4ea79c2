"""Data parallelism the MI355X way: one process per GPU, replicas hold identical weights, one RCCL all-reduce of the
flat trainable-gradient buffer per step over xGMI (torch.distributed backend "nccl" is RCCL on ROCm; "gloo" on CPU).

Replaces torch.nn.DataParallel at il_modules/base.py:68, il_modules/mrn.py:106,133 (single process: per-iteration
broadcast of ALL parameters -- 1.26 GB for TRBA+MRN-6 -- scatter, gather, reduce on GPU 0).  Frozen experts need no
traffic at all; BatchNorm statistics stay per replica exactly as under DataParallel.
"""
import os

import torch
import torch.distributed as dist
import torch.nn as nn


class ReplicaDataParallel(nn.Module):
    """Keeps the reference's `self.model.module` access pattern and the `module.` state_dict key prefix."""

    def __init__(self, module):
        super().__init__()
        self.module = module

    def forward(self, *args, **kwargs):
        return self.module(*args, **kwargs)


def init_distributed(backend=None):
    """Initialise torch.distributed from the torchrun environment (RANK / WORLD_SIZE / LOCAL_RANK / MASTER_*).
    Returns (rank, world, local_rank).  Single-process runs need no initialisation."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:
            backend = os.environ.get("MRN_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        if os.environ.get("MRN_SHARE_DEVICE"):      # test hook: every rank on cuda:0 (gloo only) -- exercises the N > 1
            local = 0                               # control flow of bench.py / the learners on a one-GPU box
        if backend == "nccl":
            torch.cuda.set_device(local)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local


def world_size():
    return dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1


def average_gradients(flat_grad):
    """One collective over the flat gradient buffer (sum, then divide by world inside the collective)."""
    if world_size() == 1:
        return flat_grad
    if dist.get_backend() == "nccl":
        dist.all_reduce(flat_grad, op=dist.ReduceOp.AVG)
    else:                                   # gloo has no AVG
        dist.all_reduce(flat_grad, op=dist.ReduceOp.SUM)
        flat_grad.div_(world_size())
    return flat_grad


def broadcast_parameters(flat_param, src=0):
    if world_size() > 1:
        dist.broadcast(flat_param, src=src)
    return flat_param


def barrier():
    if world_size() > 1:
        dist.barrier()
