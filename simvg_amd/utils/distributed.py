"""`simvg/utils/distributed.py:9-27` of the reference: init_dist / is_main / reduce_mean, one process per GPU over
RCCL (`backend="nccl"` is RCCL on ROCm).  `SIMVG_DIST_BACKEND=gloo` selects gloo for the CPU tests."""
import os
from datetime import timedelta

import torch
from torch import distributed as dist


def get_dist_info():
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


def init_dist():
    backend = os.environ.get("SIMVG_DIST_BACKEND", "nccl")
    if backend == "nccl":
        torch.cuda.set_device(int(os.environ["LOCAL_RANK"]))
        dist.init_process_group(backend="nccl", timeout=timedelta(minutes=3),
                                device_id=torch.device("cuda", int(os.environ["LOCAL_RANK"])))
    else:
        dist.init_process_group(backend=backend, timeout=timedelta(minutes=3))


def is_main():
    return get_dist_info()[0] == 0


def reduce_mean(tensor):
    if not (dist.is_available() and dist.is_initialized()):
        return tensor
    tensor = tensor.clone()
    dist.all_reduce(tensor.div_(dist.get_world_size()), op=dist.ReduceOp.SUM)
    return tensor
