"""The N>1 host logic (clip sharding, max-over-ranks timing, prediction gather) under gloo, world_size 2, CPU."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from helpers import ROOT  # noqa: F401
from tdeed_amd import dist as tdist


def test_shard_range_partitions_exactly():
    for n in (0, 1, 7, 8, 64, 65):
        for world in (1, 2, 3, 8):
            spans = [tdist.shard_range(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [hi - lo for lo, hi in spans]
            assert max(sizes) - min(sizes) <= 1


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n_clips, q):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    r, l, w = tdist.init(backend="gloo")
    assert (r, w) == (rank, world)
    lo, hi = tdist.shard_range(n_clips, rank, world)
    # every rank "predicts" its own clips: scores carry the global clip index so that order is checkable
    local = torch.arange(lo, hi, dtype=torch.float32).view(-1, 1, 1).expand(-1, 4, 3).contiguous()
    full = tdist.gather_clips(local, n_clips)
    grad = torch.full((10,), float(rank + 1))            # flat gradient buffer: mean over ranks = 1.5
    tdist.all_reduce_mean_(grad)
    assert torch.allclose(grad, torch.full((10,), 1.5))
    slow = tdist.max_over_ranks(0.5 + rank)            # rank 1 is the slow one
    tdist.barrier()
    q.put((rank, full[:, 0, 0].tolist(), slow))
    dist.destroy_process_group()


@pytest.mark.parametrize("n_clips", [8, 5])
def test_two_rank_gather_and_timing(n_clips):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, n_clips, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, order, slow in res:
        assert order == [float(i) for i in range(n_clips)]      # global clip order restored on every rank
        assert slow == 1.5                                        # max over ranks


def _reducer_worker(rank, world, port, q):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    tdist.init(backend="gloo")
    flat = torch.arange(100, dtype=torch.float32) * (rank + 1)          # rank 0: i, rank 1: 2i
    red = tdist.GradReducer(flat, [(60, 100), (0, 60)])
    assert red.backend == "torch" and not red.capturable and red.scale == 0.5
    red.reduce_bucket(0)                                               # the late bucket first, like the backward
    red.reduce_bucket(1)
    red.join()
    q.put((rank, flat.tolist()))
    tdist.barrier()
    dist.destroy_process_group()


def test_bucketed_grad_reducer_sums_every_bucket_over_two_ranks():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_reducer_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
    want = [3.0 * i for i in range(100)]
    assert res[0][1] == want and res[1][1] == want
