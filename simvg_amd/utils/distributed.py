"""Process-group helpers with the reference's names (`simvg/utils/distributed.py:9-27`: init_dist / get_dist_info /
is_main / reduce_mean): one process per GPU over RCCL (`backend="nccl"` is RCCL on ROCm).  `SIMVG_DIST_BACKEND=gloo`
selects gloo (CPU tests, and the two-processes-on-one-GPU test)."""
import os
from datetime import timedelta

import torch
import torch.distributed as dist

_TIMEOUT = timedelta(minutes=3)


def _live():
    return dist.is_available() and dist.is_initialized()


def get_dist_info():
    """(rank, world_size); (0, 1) outside a process group"""
    return (dist.get_rank(), dist.get_world_size()) if _live() else (0, 1)


def is_main():
    rank, _ = get_dist_info()
    return rank == 0


def init_dist():
    """Join the group the launcher (torchrun: RANK / LOCAL_RANK / WORLD_SIZE / MASTER_*) describes."""
    backend = os.environ.get("SIMVG_DIST_BACKEND", "nccl")
    kwargs = dict(backend=backend, timeout=_TIMEOUT)
    if backend == "nccl":
        local = int(os.environ["LOCAL_RANK"])
        torch.cuda.set_device(local)
        kwargs["device_id"] = torch.device("cuda", local)       # binds the RCCL communicator to this GPU up front
    dist.init_process_group(**kwargs)


def reduce_mean(tensor):
    """mean over the ranks (a copy; the argument is left alone); identity outside a process group"""
    if not _live():
        return tensor
    out = tensor.detach().clone() / dist.get_world_size()
    dist.all_reduce(out, op=dist.ReduceOp.SUM)
    return out
