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


def count_ranks(device=None):
    """How many ranks the default process group REALLY has: a one-element SUM all-reduce of ones (on `device` when the group
    is RCCL).  {"backend": "none" | "nccl" | "gloo", "ranks": n}; a single process reports ("none", 1)."""
    if not (dist.is_available() and dist.is_initialized()):
        return dict(backend="none", ranks=1)
    backend = dist.get_backend()
    one = torch.ones(1, dtype=torch.float32, device=(device if backend == "nccl" else "cpu"))
    dist.all_reduce(one, op=dist.ReduceOp.SUM)
    return dict(backend=backend, ranks=int(round(float(one.item()))))


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


class RcclComm:
    """The C-ABI communicator (csrc/comm.hip): RCCL on its own high-priority HIP stream.  The 128-byte unique id is made
    on rank 0 and broadcast through the default torch.distributed group (any backend), which is also what launched us."""

    def __init__(self, device):
        import ctypes
        from . import _lib
        self._lib, self._ct = _lib, ctypes
        world, rank = dist.get_world_size(), dist.get_rank()
        idbuf = torch.zeros(128, dtype=torch.uint8)
        if rank == 0:
            raw = (ctypes.c_ubyte * 128)()
            _lib.call("tdeed_comm_unique_id", raw)
            idbuf = torch.tensor(list(raw), dtype=torch.uint8)
        on_dev = dist.get_backend() == "nccl"
        t = idbuf.to(device) if on_dev else idbuf
        dist.broadcast(t, src=0)
        raw = (ctypes.c_ubyte * 128)(*t.cpu().tolist())
        self.handle = ctypes.c_void_p()
        with torch.cuda.device(device):
            _lib.call("tdeed_comm_init", ctypes.byref(self.handle), raw, world, rank)
        self.world, self.rank = world, rank

    def info(self):
        """(world, rank) as the communicator itself reports them (tdeed_comm_info)."""
        w, r = self._ct.c_int(0), self._ct.c_int(0)
        self._lib.call("tdeed_comm_info", self.handle, self._ct.byref(w), self._ct.byref(r))
        return int(w.value), int(r.value)

    def all_reduce(self, t, rs_ag=False):
        """In-place sum over ranks.  rs_ag: reduce-scatter + all-gather over the largest prefix that divides evenly over
        the ranks, a plain all-reduce of the (< world elements) rest -- tdeed_comm_all_reduce_rs_ag does both."""
        from ._lib import dtype_code, ptr, stream_ptr
        fn = "tdeed_comm_all_reduce_rs_ag" if rs_ag else "tdeed_comm_all_reduce"
        self._lib.call(fn, self.handle, ptr(t), t.numel(), dtype_code(t.dtype), stream_ptr())

    def join(self):
        from ._lib import stream_ptr
        self._lib.call("tdeed_comm_join", self.handle, stream_ptr())

    def close(self):
        if self.handle:
            self._lib.call("tdeed_comm_destroy", self.handle)
            self.handle = None


class GradReducer:
    """Bucketed gradient reduction of ONE flat buffer, overlapped with the backward.

    `buckets` = [(lo, hi), ...] element ranges in the order the backward completes them (temporal stack + heads first:
    92 % of the 800MF gradient bytes, then the trunk).  reduce_bucket(i) enqueues the SUM over ranks of that range behind
    the work already queued on the current stream and returns; join() makes the current stream wait for all of them.  The
    mean's 1/world is NOT applied here: the fused AdamW launch takes it as grad_scale (`scale`).
    backend "rccl": the C-ABI communicator (own stream, capturable into a HIP graph); backend "torch": torch.distributed
    async all_reduce on a side stream (gloo works with GPU tensors: the 2-process single-GPU tests; also CPU tensors)."""

    RS_AG_MIN_BYTES = 32 << 20      # buckets at least this large go out as reduce-scatter + all-gather

    def __init__(self, flat, buckets, backend=None, device=None, rs_ag_min_bytes=None):
        if not (dist.is_available() and dist.is_initialized()):
            raise RuntimeError("GradReducer needs an initialised torch.distributed process group")
        self.flat, self.buckets = flat, [(int(a), int(b)) for a, b in buckets]
        self.world = dist.get_world_size()
        self.rank = dist.get_rank()
        self.scale = 1.0 / self.world
        self.rs_ag_min_bytes = self.RS_AG_MIN_BYTES if rs_ag_min_bytes is None else int(rs_ag_min_bytes)
        from collections import deque
        self.launched = deque(maxlen=4 * max(1, len(self.buckets)))  # the last collectives enqueued (describe(), tests): bounded
        if backend is None:
            backend = "rccl" if (flat.is_cuda and dist.get_backend() == "nccl") else "torch"
        self.backend = backend
        self.capturable = backend == "rccl"
        self._works = []
        self._side = None
        if backend == "torch" and flat.is_cuda:
            from .streams import new_stream
            self._side = new_stream(flat.device)
        self._comm = RcclComm(device if device is not None else flat.device) if backend == "rccl" else None

    def plan(self, i):
        """The collective(s) bucket i goes out as: {"collective": "rs_ag" | "all_reduce", "numel", "rs_ag_numel" (the prefix
        that divides evenly over the ranks), "tail" (the rest, < world elements, plain all-reduce), "shard_aligned" (every
        rank's shard starts on a 16-byte boundary)}.  Both transports follow it."""
        lo, hi = self.buckets[i]
        n = hi - lo
        es = self.flat.element_size()
        if n * es >= self.rs_ag_min_bytes and self.world > 1 and n >= self.world:
            main = n // self.world * self.world
            per = main // self.world
            return dict(collective="rs_ag", numel=n, rs_ag_numel=main, tail=n - main,
                        shard_aligned=(lo * es) % 16 == 0 and (per * es) % 16 == 0)
        return dict(collective="all_reduce", numel=n, rs_ag_numel=0, tail=0, shard_aligned=(lo * es) % 16 == 0)

    def _torch_collectives(self, view, pl):
        """torch.distributed form of plan(): returns what join() finishes -- async works, or (for RS+AG) a closure that
        waits for the reduce-scatter and then gathers the shards (gloo runs independent async works concurrently, so the
        all-gather must not be enqueued before the reduce-scatter has produced its shard; the shard lives in a buffer of
        its own because gloo's reduce-scatter does not support an output aliasing its input)."""
        if pl["collective"] != "rs_ag":
            return [dist.all_reduce(view, op=dist.ReduceOp.SUM, async_op=True)]
        main, per = pl["rs_ag_numel"], pl["rs_ag_numel"] // self.world
        shard = torch.empty(per, dtype=view.dtype, device=view.device)
        rs = dist.reduce_scatter_tensor(shard, view[:main], op=dist.ReduceOp.SUM, async_op=True)
        tail = dist.all_reduce(view[main:], op=dist.ReduceOp.SUM, async_op=True) if pl["tail"] else None

        class _Chain:
            def wait(_self):
                rs.wait()
                dist.all_gather_into_tensor(view[:main], shard)
                if tail is not None:
                    tail.wait()
        return [_Chain()]

    def reduce_bucket(self, i):
        lo, hi = self.buckets[i]
        view = self.flat[lo:hi]
        if self.world == 1:
            return
        pl = self.plan(i)
        self.launched.append((i, pl["collective"]))
        if self._comm is not None:
            self._comm.all_reduce(view, rs_ag=pl["collective"] == "rs_ag")
        elif self._side is not None:
            self._side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(self._side):
                self._works += self._torch_collectives(view, pl)
        else:
            self._works += self._torch_collectives(view, pl)

    def describe(self):
        """What the bench line prints about the reduction path of a data-parallel run: per bucket its size and the
        collective it goes out as (plan()), and -- from the communicator itself -- how many ranks RCCL really has."""
        d = dict(backend=self.backend, world=self.world, buckets_mb=[round((b - a) * 4 / 2 ** 20, 1) for a, b in self.buckets],
                 bucket_collectives=[self.plan(i)["collective"] if self.world > 1 else "none" for i in range(len(self.buckets))],
                 rs_ag_tail_elems=[self.plan(i)["tail"] for i in range(len(self.buckets))],
                 shards_16B_aligned=[self.plan(i)["shard_aligned"] for i in range(len(self.buckets))],
                 capturable=self.capturable)
        if self._comm is not None:
            w, r = self._comm.info()
            d["rccl_ranks"], d["rccl_rank"] = w, r
        return d

    def reduce_all(self):
        for i in range(len(self.buckets)):
            self.reduce_bucket(i)

    def measure(self, reps=3):
        """Self-diagnosis of a data-parallel run (bench.py --mode train, N > 1): each bucket's collective ALONE on an
        otherwise idle device / process group -- median wall ms over `reps` (sync, enqueue, join, sync), the bus bandwidth it
        implies for a ring (2 (N-1)/N x bytes / time) -- so that an 8-GPU line shows at once whether a bucket runs at xGMI
        rates and how much of it the backward has to hide.  The buffer's VALUES are summed `reps` times over the ranks: call it
        on gradients nobody needs any more (the bench does, after its timed regions).  All ranks must call it together."""
        import statistics
        import time
        sync = torch.cuda.synchronize if self.flat.is_cuda else (lambda: None)
        out = []
        for i, (lo, hi) in enumerate(self.buckets):
            pl = self.plan(i) if self.world > 1 else dict(collective="none")
            ts = []
            for _ in range(reps):
                sync()
                barrier()
                t0 = time.perf_counter()
                self.reduce_bucket(i)
                self.join()
                sync()
                ts.append(time.perf_counter() - t0)
            ms = statistics.median(ts) * 1e3
            nbytes = (hi - lo) * self.flat.element_size()
            out.append(dict(bucket=i, collective=pl["collective"], mb=round(nbytes / 2 ** 20, 2), ms=round(ms, 4),
                            busbw_GBps=round(2 * (self.world - 1) / max(self.world, 1) * nbytes / max(ms, 1e-9) / 1e6, 2)))
        return out

    def join(self):
        if self._comm is not None:
            self._comm.join()
            return
        if self._side is not None:
            with torch.cuda.stream(self._side):
                for w in self._works:
                    w.wait()
            self._works = []
            torch.cuda.current_stream().wait_stream(self._side)
            return
        for w in self._works:
            w.wait()
        self._works = []

    def close(self):
        if self._comm is not None:
            self._comm.close()


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
