"""Multi-GPU plumbing for the data-parallel path (one process per GPU, torch.distributed; backend "nccl" is
RCCL on ROCm, "gloo" in the CPU tests).

The path shards by utterance: utterance i -> rank i mod G, no per-step communication.
The only collective is a one-time broadcast of the finished device weight arena (weights + precomputed
cross K/V) from the rank that parsed the GGUF; timing is reduced with MAX, work counters with SUM.
"""
import os

import torch
import torch.distributed as dist


def env_world():
    return int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))


def init(backend, rank, world, device=None):
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29511")
    kw = {}
    if backend == "nccl" and device is not None:
        kw["device_id"] = device
    dist.init_process_group(backend, rank=rank, world_size=world, **kw)


def shard_utterances(n_total, rank, world):
    """utterance i -> GPU (i mod G)"""
    return list(range(rank, n_total, world))


def broadcast_arena(arena: torch.Tensor, src=0, chunk_bytes=1 << 30):
    """Broadcast a uint8 arena in <=1 GiB pieces (xGMI ring broadcast is per-link bound either way; chunking
    only bounds the collective's staging)."""
    assert arena.dtype == torch.uint8 and arena.is_contiguous()
    n = arena.numel()
    for off in range(0, n, chunk_bytes):
        dist.broadcast(arena[off:min(n, off + chunk_bytes)], src=src)


def reduce_timing(elapsed_s, units, device="cpu"):
    """returns (max elapsed over ranks, sum of processed units over ranks)"""
    t = torch.tensor([elapsed_s], dtype=torch.float64, device=device)
    u = torch.tensor([float(units)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dist.all_reduce(u, op=dist.ReduceOp.SUM)
    return float(t.item()), float(u.item())
