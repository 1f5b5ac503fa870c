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


def _measure_worker(rank, world, port, q):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    tdist.init(backend="gloo")
    flat = torch.ones(4096, dtype=torch.float32)
    red = tdist.GradReducer(flat, [(1024, 4096), (0, 1024)], rs_ag_min_bytes=8192)
    m = red.measure(reps=2)
    q.put((rank, m, float(flat[0]), float(flat[-1])))
    tdist.barrier()
    dist.destroy_process_group()


def test_reducer_self_diagnosis_times_every_bucket_on_two_ranks():
    """GradReducer.measure (what `bench.py --mode train --gpus N` prints as dp_diag.buckets): per bucket the collective it
    goes out as, its size, a positive time and the bus bandwidth that implies; every rank runs the same collectives."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_measure_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, m, first, last in res:
        assert [b["bucket"] for b in m] == [0, 1]
        assert m[0]["collective"] == "rs_ag" and m[1]["collective"] == "all_reduce"      # 12 KB >= 8 KB, 4 KB below
        assert m[0]["mb"] == round(3072 * 4 / 2 ** 20, 2) and all(b["ms"] > 0 and b["busbw_GBps"] >= 0 for b in m)   # (KB-sized buckets over gloo: the bandwidth can round to 0.00)
        assert first == 4.0 and last == 4.0          # two repetitions of a sum over two ranks: 1 -> 2 -> 4 in both buckets


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


# ---------------------------------------------------------------------------------------------- world = 8
CFG3 = dict(feature_arch="rny008_gsf", clip_len=100, crop_dim=224, n_layers=3, sgp_ks=7, sgp_r=4, num_classes=4,
            radi_displacement=2)
CFG5 = dict(feature_arch="rny008_gsf", clip_len=250, crop_dim=None, n_layers=2, sgp_ks=9, sgp_r=4, num_classes=12,
            radi_displacement=4)
CFG2 = dict(feature_arch="rny002_gsf", clip_len=100, crop_dim=224, n_layers=2, sgp_ks=7, sgp_r=4, num_classes=4,
            radi_displacement=2)


def _flat_layout(cfg):
    """(numel, first element of the temporal-stack bucket) of the real flat parameter buffer, computed from the shapes
    alone (optim.FlatParams' rule; no tensors of that size are allocated)."""
    from tdeed_amd import state_layout
    from tdeed_amd.optim import FlatParams

    class Meta:                       # stands in for a tensor: FlatParams only needs numel / shape / copy semantics
        pass
    shapes = state_layout.model_state_shapes(cfg)
    state = {k: torch.empty(sh, dtype=torch.float32 if dt == "float32" else torch.int64, device="meta")
             for k, (sh, dt) in shapes.items()}
    fp = FlatParams.layout_only(state)
    first = min(o for k, (o, n) in fp.index.items() if k.startswith(("_temp_fine.", "_pred_")))
    return fp.numel, first, fp.index


@pytest.mark.parametrize("cfg", [CFG2, CFG3, CFG5], ids=["cfg2_200MF", "cfg3_800MF", "cfg5_snb_t250"])
def test_real_flat_buffer_buckets_divide_over_eight_ranks(cfg):
    """Both gradient buckets of the real models split evenly over 2, 4 and 8 ranks with 16-byte-aligned shards, so the
    large bucket takes the reduce-scatter + all-gather path at world 8 (with 4-element padding only it was 4 mod 8)."""
    numel, first, index = _flat_layout(cfg)
    for lo, hi in [(first, numel), (0, first)]:
        n = hi - lo
        for world in (2, 4, 8):
            assert n % world == 0, (lo, hi, world)
            assert (n // world) % 4 == 0 and lo % 4 == 0                # fp32 shards start on 16-byte boundaries
    # every tensor still starts on a 16-byte boundary and nothing overlaps
    spans = sorted(index.values())
    assert all(o % 4 == 0 for o, n in spans)
    assert all(a[0] + a[1] <= b[0] for a, b in zip(spans, spans[1:])) and spans[-1][0] + spans[-1][1] <= numel


def _reducer8_worker(rank, world, port, first, numel, q):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    tdist.init(backend="gloo")
    # a flat buffer shaped like the real one scaled down by 512 (same residues mod 8 as the real offsets would have
    # without the padding fix are covered by the odd-bucket case below); rank r holds (r + 1) * i
    base = (torch.arange(numel) % 1021).float()                  # small integers: every partial sum is exact in fp32
    flat = base * (rank + 1)
    red = tdist.GradReducer(flat, [(first, numel), (0, first)], rs_ag_min_bytes=1024)
    plans = [red.plan(0), red.plan(1)]
    red.reduce_bucket(0)
    red.reduce_bucket(1)
    red.join()
    d = red.describe()
    # an unpadded bucket (length 4 mod 8): RS+AG over the prefix that divides, plain all-reduce of the 4-element tail
    odd = torch.ones(8 * 37 + 4) * (rank + 1)
    red2 = tdist.GradReducer(odd, [(0, odd.numel())], rs_ag_min_bytes=16)
    p_odd = red2.plan(0)
    red2.reduce_bucket(0)
    red2.join()
    q.put((rank, float((flat - base * 36.0).abs().max()), plans, d,
           p_odd, float((odd - 36.0).abs().max()), red.launched))
    tdist.barrier()
    dist.destroy_process_group()


def test_eight_rank_reducer_takes_rs_ag_on_the_real_bucket_layout():
    """world = 8 over gloo: GradReducer on a buffer with the real model's bucket boundaries (scaled) must plan AND launch
    reduce-scatter + all-gather for the large bucket, sum correctly, and report the collective per bucket."""
    numel, first, _ = _flat_layout(CFG3)
    assert numel % 64 == 0 and first % 64 == 0
    # scale the 58 M-element buffer down, keeping both boundaries multiples of 64
    s_first, s_numel = first // 64 // 64 * 64, numel // 64 // 64 * 64
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_reducer8_worker, args=(r, 8, port, s_first, s_numel, q)) for r in range(8)]
    for p in procs:
        p.start()
    res = sorted((q.get(timeout=300) for _ in procs), key=lambda x: x[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, err, plans, d, p_odd, err_odd, launched in res:
        assert err == 0.0 and err_odd == 0.0
        assert plans[0]["collective"] == "rs_ag" and plans[0]["tail"] == 0 and plans[0]["shard_aligned"]
        assert d["bucket_collectives"][0] == "rs_ag" and d["world"] == 8 and d["rs_ag_tail_elems"] == [0, 0]
        assert launched[0] == (0, "rs_ag")
        assert p_odd["collective"] == "rs_ag" and p_odd["tail"] == 4 and p_odd["rs_ag_numel"] == 8 * 37


def _bench(args, env_extra, timeout=300):
    import subprocess
    import sys
    env = dict(os.environ, **env_extra)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        if k not in env_extra:
            env.pop(k, None)
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, cwd=ROOT, env=env, capture_output=True,
                          text=True, timeout=timeout)


@pytest.mark.parametrize("n", [1, 2, 3, 8])
def test_bench_gpus_flag_starts_that_many_ranks_by_itself(n):
    """`python bench.py --gpus N` without torchrun around it (the flag was parsed and never used): the
    launch path alone (TDEED_BENCH_LAUNCH_PROBE: rendezvous + a counting all-reduce, no GPU) must come up as N ranks and
    print one line from rank 0."""
    import json
    r = _bench(["--gpus", str(n), "--steps", "3"], dict(TDEED_BENCH_LAUNCH_PROBE="1"))
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == n and rec["world_env"] == n and rec["self_launched"] == (n > 1)


def test_bench_gpus_flag_under_torchrun_and_mismatch_is_refused():
    """The driver's launch line keeps working (WORLD_SIZE from torchrun == --gpus), and a --gpus that disagrees with the
    launching environment's WORLD_SIZE exits non-zero before anything is measured."""
    import json
    import subprocess
    import sys
    env = dict(os.environ, TDEED_BENCH_LAUNCH_PROBE="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2"]
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    rec = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][0])
    assert rec["n_gpus"] == 2 and rec["self_launched"] is False
    r = _bench(["--gpus", "2"], dict(WORLD_SIZE="4", RANK="0", LOCAL_RANK="0"), timeout=120)
    assert r.returncode == 2 and "disagrees with WORLD_SIZE" in r.stderr
