"""Clip-sharded multi-GPU execution (one process per GPU; RCCL via torch.distributed backend "nccl").

The forward shards by whole clips with no data-path collective: rank r owns clips
[lo, hi) of the global batch, weights are replicated.  Collectives appear only where results meet:
the max-over-ranks step time of bench.py and the gather of per-clip predictions for evaluation.
The helpers take an explicit process group / backend so the same code runs under gloo on CPU (tests).
"""
import os

import torch
import torch.distributed as dist


def env_world():
    """(rank, local_rank, world_size) from the torchrun environment (1 process if unset)."""
    return (int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")),
            int(os.environ.get("WORLD_SIZE", "1")))


def init(backend=None, device=None):
    """Initialise the default process group when WORLD_SIZE > 1 (nccl = RCCL on ROCm, gloo on CPU)."""
    rank, local, world = env_world()
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        kw = {}
        if backend == "nccl" and device is not None:
            kw["device_id"] = device
        dist.init_process_group(backend, **kw)
    return rank, local, world


def shard_range(n_items, rank, world):
    """Contiguous, balanced split of n_items clips: the first n_items % world ranks get one extra."""
    base, rem = divmod(n_items, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def max_over_ranks(seconds, device="cpu"):
    """Slowest rank's elapsed time (the step time of a synchronous data-parallel job)."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return float(seconds)
    t = torch.tensor([float(seconds)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def barrier():
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.barrier()


def gather_clips(local, n_total):
    """All-gather per-clip results (first dim = this rank's clips, shard_range order) into the global order.
    Ranks may hold different counts; pads to the largest shard for the collective."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return local
    world, rank = dist.get_world_size(), dist.get_rank()
    counts = [shard_range(n_total, r, world)[1] - shard_range(n_total, r, world)[0] for r in range(world)]
    mx = max(counts)
    pad = torch.zeros((mx,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    pad[:local.shape[0]] = local
    bufs = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(bufs, pad)
    return torch.cat([b[:c] for b, c in zip(bufs, counts)], dim=0)


def all_reduce_mean_(flat, async_op=False):
    """Gradient all-reduce of one flat buffer (sum over ranks; the 1/world is folded into the optimizer's
    grad_scale by the caller when async, applied here otherwise).  RCCL over xGMI under backend "nccl"."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return None
    if async_op:
        return dist.all_reduce(flat, op=dist.ReduceOp.SUM, async_op=True)
    dist.all_reduce(flat, op=dist.ReduceOp.SUM)
    flat.div_(dist.get_world_size())
    return None
