"""Rank plumbing shared by bench.py and the multi-process tests: one process per GPU, independent
windows (replicas) per rank, no data-path collective -- only the timing reduction of the bench
contract (max over ranks) and the count of completed LM iterations (min over ranks)."""
import os


def rank_info():
    return int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))


def rank_seed(base_seed, config, rank):
    """Each rank solves its own window: distinct deterministic seed per (config, rank)."""
    return base_seed + config + 100 * rank


def reduce_timing(dist, device, seconds, steps_done):
    """(max seconds over ranks, min completed steps over ranks); dist None -> identity."""
    if dist is None:
        return seconds, steps_done
    import torch
    t = torch.tensor([seconds], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    s = torch.tensor([float(steps_done)], dtype=torch.float64, device=device)
    dist.all_reduce(s, op=dist.ReduceOp.MIN)
    return float(t.item()), int(s.item())


def aggregate_throughput(world, steps_done, seconds):
    """Whole-job LM iterations per second: every rank completed `steps_done` iterations in `seconds`."""
    return world * steps_done / seconds


class _DeviceArray:
    """Zero-copy view of a device buffer handed out by the C ABI (obvi_ba_set_allreduce callback)."""
    def __init__(self, ptr, n):
        self.__cuda_array_interface__ = {"data": (int(ptr), False), "shape": (int(n),), "typestr": "<f8", "version": 2}


def device_tensor(ptr, n):
    import torch
    return torch.as_tensor(_DeviceArray(ptr, n), device="cuda")


def torch_allreduce(dist):
    """obvi_ba all-reduce hook on top of torch.distributed (backend nccl == RCCL over xGMI): the collective is enqueued
    behind the library's own HIP stream, no host synchronisation."""
    import torch

    def fn(ptr, count, op, stream):
        with torch.cuda.stream(torch.cuda.ExternalStream(stream)):
            dist.all_reduce(device_tensor(ptr, count), op=dist.ReduceOp.MAX if op else dist.ReduceOp.SUM)
        return 0
    return fn
