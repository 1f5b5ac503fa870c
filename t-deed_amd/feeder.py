"""Input pipeline of the hot path: JPEG frames -> uint8 clips -> device, overlapped with compute.

The reference decodes every frame with `torchvision.io.read_image` inside DataLoader workers (/root/reference/dataset/
frame.py:263-382 `FrameReader.load_frames`, 546-626 `FrameReaderVideo`), stacks them to a uint8 (T,3,H,W) clip and lets
`epoch()` / `predict()` move the batch with a blocking `.to(device).float()` (model/model.py:216, 342): 4 bytes per pixel
over PCIe, no overlap with the forward.  Here

  * `read_frame` / `load_clip` decode JPEGs (Pillow's libjpeg, the same decoder family as torchvision's) straight into a
    caller-provided uint8 buffer -- a slot of a pinned staging ring -- with the reference's zero padding of clips that
    start before / end after the video;
  * `PinnedRing` owns `depth` pinned host slots and matching device buffers; `upload()` issues ONE asynchronous H2D copy
    of a uint8 batch (1 byte per pixel: 15 MB per clip of 100 x 224 x 224 frames, ~0.24 ms at PCIe 5 x16) on a dedicated
    copy stream and returns the device tensor with the event that marks its arrival;
  * `prefetch(loader, ...)` wraps any iterable of batch dicts (the reference's loaders) so that batch i+1 is staged and
    copied while batch i is being processed: `TDEEDModel.epoch()` uses it for both the validation and the training loop.

Only plumbing lives here (host memory, streams, events); nothing in this file computes on the device.
"""
import os

import numpy as np
from types import SimpleNamespace
import torch


def read_frame(path, out=None):
    """JPEG (or any Pillow-readable image) -> uint8 (3,H,W) RGB, like torchvision.io.read_image (frame.py:271, 555).
    out: optional uint8 array/tensor view (3,H,W) to decode into."""
    from PIL import Image
    with Image.open(path) as im:
        a = np.asarray(im.convert("RGB"))                       # (H, W, 3) uint8
    chw = np.moveaxis(a, 2, 0)
    if out is None:
        return torch.from_numpy(np.ascontiguousarray(chw))
    dst = out.numpy() if isinstance(out, torch.Tensor) else out
    if tuple(dst.shape) != tuple(chw.shape):
        raise ValueError(f"frame {path}: {chw.shape} does not fit the buffer {tuple(dst.shape)}")
    np.copyto(dst, chw)
    return out


def load_clip_video(frame_dir, dataset, video_name, start, end, pad=False, stride=1, source_info=None, out=None):
    """`FrameReaderVideo.load_frames(video_name, start, end, pad, stride, source_info)` (frame.py:558-626) with the
    reference's per-dataset file naming (`frame_locator`)."""
    _, _, _, path_fn = frame_locator(frame_dir, dataset, video_name, source_info)
    return load_clip(path_fn, start, end, stride=stride, pad=pad, out=out)


def load_clip(frame_path_fn, start, end, stride=1, pad=False, out=None):
    """`FrameReaderVideo.load_frames` (frame.py:558-626) for one clip: frames start, start+stride, ... < end through
    `frame_path_fn(frame_num) -> path`; frames before 0 pad the start with zeros, missing files pad the end (kept only
    when `pad`).  Returns a uint8 (T,3,H,W) tensor (a view of `out` when given), or -1 when no frame exists."""
    nums = list(range(start, end, stride))
    n_pad_start = sum(1 for n in nums if n < 0)
    todo = [n for n in nums if n >= 0]
    frames = []
    n_pad_end = 0
    for j, n in enumerate(todo):
        p = frame_path_fn(n)
        if not os.path.exists(p):
            n_pad_end += 1
            continue
        dst = None if out is None else out[n_pad_start + len(frames)]
        frames.append(read_frame(p, out=dst))
    if not frames:
        return -1
    T_real = len(frames)
    if out is not None:
        if n_pad_start:
            out[:n_pad_start].zero_()
        tail = n_pad_start + T_real
        if pad and n_pad_end:
            out[tail:tail + n_pad_end].zero_()
            tail += n_pad_end
        return out[:tail]
    clip = torch.stack(frames, 0)
    if n_pad_start > 0 or (pad and n_pad_end > 0):
        clip = torch.nn.functional.pad(clip, (0, 0, 0, 0, 0, 0, n_pad_start, n_pad_end if pad else 0))
    return clip


def frame_locator(frame_dir, dataset, video_name, source_info=None):
    """Where the frames of one video live, per dataset, as both of the reference's readers lay it out
    (frame.py:274-292 / 563-578 for the directory and the first frame number, 303-338 / 586-609 for the file names).
    Returns (base_path, frame0, ndigits, path_fn): frame index n of the video (0 = its first frame) is the file
    path_fn(n); ndigits is the zero-padded width of FineDiving's file names, -1 for the 'frame<N>.jpg' datasets."""
    ndigits = -1
    frame0 = 0
    if dataset == "finediving":
        base = os.path.join(frame_dir, video_name.replace("__", "/"))
        first = sorted(os.listdir(base))[0]
        ndigits, frame0 = len(first[:-4]), int(first[:-4])
        return base, frame0, ndigits, (lambda n: os.path.join(base, str(frame0 + n).zfill(ndigits) + ".jpg"))
    if dataset == "tennis":
        parts = video_name.split("_")
        frame0 = int(parts[-2])
        base = os.path.join(frame_dir, "_".join(parts[:-2]))
    elif dataset == "finegym":
        frame0 = source_info["start_frame"] - source_info["pad"][0]
        base = os.path.join(frame_dir, video_name.split("_")[0])
    elif dataset in ("soccernetball", "soccernet", "fs_comp", "fs_perf"):
        base = os.path.join(frame_dir, video_name)
    else:
        raise ValueError(f"unknown dataset {dataset!r}")
    return base, frame0, ndigits, (lambda n: os.path.join(base, "frame" + str(frame0 + n) + ".jpg"))


def load_paths(frame_dir, dataset, video_name, start, end, stride=1, source_info=None):
    """`FrameReader.load_paths` (frame.py:273-351), the training reader's clip descriptor:
    [base_path, first existing frame NUMBER (file numbering, -1 if none), pad_start, pad_end, ndigits, length] with
    length = (end - start) // stride.  Frames before 0 count as start padding; from the first missing file on every
    remaining step counts as end padding (the reference stops looking: a later file that exists is not read)."""
    base, frame0, ndigits, path_fn = frame_locator(frame_dir, dataset, video_name, source_info)
    found_start, pad_start, pad_end = -1, 0, 0
    for n in range(start, end, stride):
        if n < 0:
            pad_start += 1
            continue
        if pad_end > 0:
            pad_end += 1
            continue
        if os.path.exists(path_fn(n)):
            if found_start == -1:
                found_start = frame0 + n
        else:
            pad_end += 1
    return [base, found_start, pad_start, pad_end, ndigits, (end - start) // stride]


def load_frames(paths, pad=False, stride=1, out=None, pool=None):
    """`FrameReader.load_frames` (frame.py:353-382): the `length - pad_start - pad_end` frames numbered paths[1],
    paths[1] + stride, ... decoded into a uint8 (T,3,H,W) clip, zero frames in front for pad_start, behind for pad_end when
    `pad`.  out: optional uint8 buffer (>= T frames: a pinned ring slot) to decode into; pool: a DecodePool to decode the
    frames concurrently."""
    base, start, pad_start, pad_end, ndigits, length = paths
    n_real = length - pad_start - pad_end
    if ndigits == -1:
        names = [os.path.join(base, "frame") + str(start + j * stride) + ".jpg" for j in range(n_real)]
    else:
        names = [base + "/" + str(start + j * stride).zfill(ndigits) + ".jpg" for j in range(n_real)]
    return _assemble(names, pad_start, pad_end if pad else 0, out, pool)


def _assemble(names, n_pad_start, n_pad_end, out, pool):
    """Decode `names` into [zeros * n_pad_start | frames | zeros * n_pad_end]."""
    if out is None:
        first = read_frame(names[0])
        total = n_pad_start + len(names) + n_pad_end
        out = torch.zeros((total,) + tuple(first.shape), dtype=torch.uint8)
        out[n_pad_start].copy_(first)
        rest = [(nm, out[n_pad_start + 1 + i]) for i, nm in enumerate(names[1:])]
    else:
        total = n_pad_start + len(names) + n_pad_end
        if out.shape[0] < total:
            raise ValueError(f"staging buffer holds {out.shape[0]} frames, the clip needs {total}")
        if n_pad_start:
            out[:n_pad_start].zero_()
        if n_pad_end:
            out[n_pad_start + len(names):total].zero_()
        rest = [(nm, out[n_pad_start + i]) for i, nm in enumerate(names)]
    if pool is not None:
        pool.decode(rest)
    else:
        for nm, dst in rest:
            read_frame(nm, out=dst)
    return out[:total]


class DecodePool:
    """Decode threads of the input pipeline (the reference uses 4-8 DataLoader worker PROCESSES, train_tdeed.py:131-139;
    Pillow releases the GIL inside libjpeg, so threads scale and decode straight into pinned ring slots without a copy
    between processes).  decode([(path, uint8 (3,H,W) destination), ...]) returns when all frames are in place."""

    def __init__(self, threads=None):
        from concurrent.futures import ThreadPoolExecutor
        self.threads = threads if threads else min(32, (os.cpu_count() or 8))
        self.ex = ThreadPoolExecutor(max_workers=self.threads, thread_name_prefix="tdeed-decode")

    def decode(self, jobs):
        futs = [self.ex.submit(read_frame, nm, dst) for nm, dst in jobs]
        for f in futs:
            f.result()

    def submit(self, jobs):
        """Asynchronous form: returns the futures (wait with .result())."""
        return [self.ex.submit(read_frame, nm, dst) for nm, dst in jobs]

    def close(self):
        self.ex.shutdown(wait=True)


class _Ack:
    """Future of one decode job of a ProcessDecodePool."""
    __slots__ = ("ev", "err", "worker")

    def __init__(self, worker=-1):
        import threading
        self.ev, self.err, self.worker = threading.Event(), None, worker

    def result(self, timeout=120.0):
        if not self.ev.wait(timeout):
            raise TimeoutError("decode worker did not answer")
        if self.err:
            raise RuntimeError(self.err)


class ProcessDecodePool:
    """Decode worker PROCESSES (what the reference's DataLoader workers are, train_tdeed.py:131-139): JPEG decode in Python
    scales to ~8 threads (DecodePool: Pillow releases the GIL only inside libjpeg), processes scale with the cores.  The
    workers (`_decode_worker.py`: numpy + Pillow, no torch) write straight into staging slots that live in POSIX shared
    memory and are page-locked in this process (`make_slots`), so a decoded batch is uploaded without a copy between
    processes.  Same interface as DecodePool."""

    def __init__(self, procs=None):
        import subprocess
        import sys
        import threading
        self.procs = procs if procs else min(32, (os.cpu_count() or 8))
        root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
        boot = f"import sys; sys.path.insert(0, {root!r}); import tdeed_amd._decode_worker as w; w.main()"
        self._w, self._pending, self._lock, self._next, self._rr = [], {}, threading.Lock(), 0, 0
        self._shms, self._slots, self._slot_cache = [], [], {}
        self._alive = []
        self._closing = False
        # one decode thread per worker: numpy's BLAS / OpenMP pools otherwise start a thread per host core in EVERY worker
        # (128 workers on a 256-thread host = 32 k threads: the decode rate fell from 412 to 32 clips/s)
        wenv = dict(os.environ, OMP_NUM_THREADS="1", OPENBLAS_NUM_THREADS="1", MKL_NUM_THREADS="1", NUMEXPR_NUM_THREADS="1")
        for wi in range(self.procs):
            pr = subprocess.Popen([sys.executable, "-c", boot], stdin=subprocess.PIPE, stdout=subprocess.PIPE, text=True, bufsize=1,
                                  env=wenv)
            self._w.append(pr)
            self._alive.append(True)
            th = threading.Thread(target=self._reader, args=(pr, wi), daemon=True)
            th.start()

    def _reader(self, pr, wi):
        try:
            for line in pr.stdout:
                parts = line.rstrip("\n").split(" ", 2)
                try:
                    jid = int(parts[0])
                    status = parts[1]
                except (ValueError, IndexError):
                    continue                    # not a protocol line (a library's warning on the worker's stdout): skip it
                with self._lock:
                    ack = self._pending.pop(jid, None)
                if ack is not None:
                    if status != "ok":
                        ack.err = parts[2] if len(parts) > 2 else "decode worker error"
                    ack.ev.set()
        finally:
            # EOF on the worker's pipe (or this thread dying): the worker exited (crash, OOM kill, or close()).  Fail the
            # jobs it still owed right away instead of letting every result() run into its timeout, and take it out of
            # the rotation.
            try:
                rc = pr.wait(timeout=2.0)
            except Exception:           # noqa: BLE001  (still running: the pipe closed first)
                rc = pr.poll()
            with self._lock:
                self._alive[wi] = False
                owed = [j for j, a in self._pending.items() if a.worker == wi]
                for j in owed:
                    ack = self._pending.pop(j)
                    ack.err = f"decode worker {wi} exited (rc={rc}) with the job outstanding"
                    ack.ev.set()

    def _pick_worker(self):
        """next live worker of the rotation (caller holds the lock)"""
        for _ in range(self.procs):
            wi = self._rr % self.procs
            self._rr += 1
            if self._alive[wi]:
                return wi
        raise RuntimeError("ProcessDecodePool: no live decode worker left")

    def _send(self, wi, line, ack, jid):
        try:
            self._w[wi].stdin.write(line)
        except (BrokenPipeError, ValueError, OSError) as e:
            with self._lock:
                self._alive[wi] = False
                self._pending.pop(jid, None)
            ack.err = f"decode worker {wi} is gone ({type(e).__name__})"
            ack.ev.set()

    def make_slots(self, depth, shape):
        """`depth` uint8 staging tensors of `shape` in shared memory, page-locked when a GPU runtime is there.  Cached per
        (depth, shape): a second loader over the same pool reuses the blocks instead of piling up shared memory."""
        from multiprocessing import shared_memory
        key = (int(depth), tuple(int(v) for v in shape))
        if key in self._slot_cache:
            return self._slot_cache[key][0]
        n = int(np.prod(shape))
        out = []
        for _ in range(depth):
            shm = shared_memory.SharedMemory(create=True, size=n)
            t = torch.frombuffer(shm.buf, dtype=torch.uint8, count=n).view(*shape)
            if torch.cuda.is_available():
                rc = torch.cuda.cudart().cudaHostRegister(t.data_ptr(), n, 0)
                if int(rc) != 0:
                    raise RuntimeError(f"cudaHostRegister failed: {rc}")
            self._shms.append((shm, t.data_ptr(), n, torch.cuda.is_available()))
            self._slots.append(t)
            out.append(t)
        # the upload-event holders that guard a slot's reuse live WITH the slots: a second loader over the same pool and
        # shape (train + val, or a new epoch while an upload from the old generator is still in flight) waits on the same
        # events before it decodes into a slot
        from types import SimpleNamespace
        self._slot_cache[key] = (out, [SimpleNamespace(event=None) for _ in range(depth)])
        return out

    def slot_holders(self, depth, shape):
        """the per-slot upload-event holders of make_slots(depth, shape) (shared by every loader that uses those slots)"""
        self.make_slots(depth, shape)
        return self._slot_cache[(int(depth), tuple(int(v) for v in shape))][1]

    def _locate(self, dst):
        p = dst.data_ptr()
        for shm, base, n, _ in self._shms:
            if base <= p < base + n:
                return shm.name, p - base
        raise ValueError("ProcessDecodePool: destination is not inside a slot from make_slots()")

    def submit(self, jobs):
        acks = []
        for path, dst in jobs:
            name, off = self._locate(dst)
            with self._lock:
                wi = self._pick_worker()
                ack = _Ack(wi)
                jid = self._next
                self._next += 1
                self._pending[jid] = ack
            self._send(wi, f"{jid} {name} {off} {dst.shape[-2]} {dst.shape[-1]} {path}\n", ack, jid)
            acks.append(ack)
        self._flush(range(self.procs))
        return acks

    def _flush(self, workers):
        for wi in workers:
            if self._alive[wi]:
                try:
                    self._w[wi].stdin.flush()
                except (BrokenPipeError, ValueError, OSError):
                    pass                # the reader thread fails the worker's jobs at EOF

    def submit_rows(self, paths, dst, chunk=25):
        """Consecutive frames: paths[i] -> dst[i] (dst: a contiguous (n,3,H,W) view inside a slot), `chunk` frames per job
        line -- the parent's per-frame work (formatting, pipe write, acknowledgement) otherwise caps the pool near 30 000
        frames/s however many workers there are."""
        import json
        if not dst.is_contiguous():
            raise ValueError("submit_rows: the destination rows must be contiguous")
        name, off = self._locate(dst)
        H, W = dst.shape[-2], dst.shape[-1]
        acks, touched = [], set()
        for lo in range(0, len(paths), chunk):
            with self._lock:
                wi = self._pick_worker()
                ack = _Ack(wi)
                jid = self._next
                self._next += 1
                self._pending[jid] = ack
            self._send(wi, "J " + json.dumps(dict(id=jid, shm=name, off=off + lo * 3 * H * W, H=H, W=W,
                                                  paths=list(paths[lo:lo + chunk]))) + "\n", ack, jid)
            touched.add(wi)
            acks.append(ack)
        self._flush(touched)
        return acks

    def decode(self, jobs):
        for a in self.submit(jobs):
            a.result()

    def close(self):
        for pr in self._w:
            try:
                pr.stdin.write("quit\n")
                pr.stdin.flush()
                pr.stdin.close()
            except Exception:       # noqa: BLE001
                pass
        for pr in self._w:
            try:
                pr.wait(timeout=10)
            except Exception:       # noqa: BLE001
                pr.kill()
        self._w = []
        self._slots = []
        self._slot_cache = {}
        for shm, ptr_, n, pinned in self._shms:
            try:
                if pinned:
                    torch.cuda.cudart().cudaHostUnregister(ptr_)
            except Exception:       # noqa: BLE001
                pass
            try:
                shm.close()
            except BufferError:
                pass                # a tensor still views the block: the mapping goes with the process
            try:
                shm.unlink()
            except FileNotFoundError:
                pass
        self._shms = []

    def __del__(self):
        try:
            if self._w:
                self.close()
        except Exception:           # noqa: BLE001
            pass


def clip_batches(clips, batch_size, frame_shape, clip_len, pool=None, depth=3, pad=True):
    """Batches of decoded clips in pinned staging slots, ready for `prefetch`: `clips` is a list of descriptors
    dict(paths=<load_paths result>, stride=..) (+ any label entries, passed through per batch as lists); every batch dict
    holds 'frame' = a PINNED uint8 (B,T,3,H,W) tensor filled by `pool` (all frames of the batch are decode jobs of one
    pool call, so B * T JPEGs decode concurrently).  The slots rotate: a batch must be consumed (copied to the device)
    before `depth` further batches are produced.  `prefetch` uploads a pinned batch straight from its slot, asynchronously:
    it leaves the arrival event in the batch's `_src` holder, and a slot is not decoded into again before that event has
    completed (a consumer that never syncs -- graph replays -- would otherwise see frames overwritten mid-copy).  Rows of
    a clip behind its real frames are always zeroed (pad=False only drops the END padding from the clip's length in the
    reference, dataset/frame.py:355-382; a fixed-length slot must not show an earlier batch's frames there)."""
    own = pool is None
    pool = pool if pool is not None else DecodePool()
    shape = (batch_size, clip_len) + tuple(frame_shape)
    if hasattr(pool, "make_slots"):          # worker processes decode into shared, page-locked slots
        slots = pool.make_slots(depth, shape)
        holders = pool.slot_holders(depth, shape) if hasattr(pool, "slot_holders") else None
    else:
        holders = None
        # (page-locked when a GPU runtime is there, like make_slots: the decode logic itself also runs on a GPU-less host)
        slots = [torch.zeros(shape, dtype=torch.uint8) for _ in range(depth)]
        if torch.cuda.is_available():
            slots = [t_.pin_memory() for t_ in slots]
    if holders is None:
        holders = [SimpleNamespace(event=None) for _ in range(depth)]
    try:
        for bi, lo in enumerate(range(0, len(clips) - batch_size + 1, batch_size)):
            slot = slots[bi % depth]
            hold = holders[bi % depth]
            if hold.event is not None:       # the upload that last read this slot (prefetch) must have left it
                hold.event.synchronize()
                hold.event = None
            group = clips[lo:lo + batch_size]
            futs = []
            for i, c in enumerate(group):
                base, start, pad_start, pad_end, ndigits, length = c["paths"]
                stride = c.get("stride", 1)
                n_real = length - pad_start - pad_end
                dst = slot[i]
                if pad_start:
                    dst[:pad_start].zero_()
                if pad_start + n_real < clip_len:
                    dst[pad_start + n_real:].zero_()
                names = []
                for j in range(n_real):
                    num = start + j * stride
                    names.append((os.path.join(base, "frame") + str(num) + ".jpg") if ndigits == -1
                                 else (base + "/" + str(num).zfill(ndigits) + ".jpg"))
                if hasattr(pool, "submit_rows"):
                    futs += pool.submit_rows(names, dst[pad_start:pad_start + n_real])
                else:
                    futs += pool.submit([(nm, dst[pad_start + j]) for j, nm in enumerate(names)])
            for f in futs:
                f.result()
            extra = {k: [c[k] for c in group] for k in group[0] if k not in ("paths", "stride")}
            yield dict(frame=slot, _src=hold, **extra)
    finally:
        if own:
            pool.close()


class PinnedRing:
    """`depth` pinned host slots + device buffers of one batch geometry (B,T,3,H,W) uint8 and a copy stream."""

    def __init__(self, shape, device="cuda", depth=3):
        self.shape, self.device, self.depth = tuple(shape), device, depth
        self.host = [torch.empty(self.shape, dtype=torch.uint8).pin_memory() for _ in range(depth)]
        self.dev = [torch.empty(self.shape, dtype=torch.uint8, device=device) for _ in range(depth)]
        self.copied = [None] * depth            # event: H2D copy of the slot finished
        self.consumed = [None] * depth          # event: the consumer is done with the device buffer
        from .streams import new_stream
        self.stream = new_stream(device)                  # never the consumer's own stream (torch's pool wraps around)
        self.i = 0

    def next_slot(self):
        """Slot to fill next; blocks (host) only if its previous copy has not left the pinned buffer yet."""
        j = self.i % self.depth
        self.i += 1
        if self.copied[j] is not None:
            self.copied[j].synchronize()
        return j

    def upload(self, j, src=None):
        """Async H2D of slot j on the copy stream (src: an already pinned tensor to copy from instead of the slot's own
        host buffer).  Returns (device tensor, arrival event)."""
        host = self.host[j] if src is None else src
        with torch.cuda.stream(self.stream):
            if self.consumed[j] is not None:
                self.stream.wait_event(self.consumed[j])        # do not overwrite a buffer the forward still reads
            self.dev[j].copy_(host, non_blocking=True)
            ev = torch.cuda.Event()
            ev.record(self.stream)
        self.copied[j] = ev
        return self.dev[j], ev

    def release(self, j, stream=None):
        """The consumer's stream no longer needs slot j after everything queued on it so far."""
        ev = torch.cuda.Event()
        ev.record(stream if stream is not None else torch.cuda.current_stream())
        self.consumed[j] = ev


def wait(batch, stream=None):
    """Explicit mode of `prefetch`: make `stream` (default: the current one) wait for the batch's frames to have arrived."""
    slot = batch.get("_slot")
    if slot is not None:
        (stream if stream is not None else torch.cuda.current_stream()).wait_event(slot[2])


def done(batch, stream=None):
    """Explicit mode of `prefetch`: everything queued on `stream` so far was the last use of the batch's device frames."""
    slot = batch.pop("_slot", None)
    if slot is not None:
        slot[0].release(slot[1], stream)


def prefetch(loader, device="cuda", key="frame", depth=3, auto=True):
    """Iterate `loader` (batch dicts) one batch ahead: `batch[key]` (uint8 host tensor) is staged in a pinned ring and
    copied to the device on a copy stream while the previous batch is processed; the yielded dict holds the device tensor.
    Tensors that are already pinned are copied without the staging memcpy; device tensors pass through.
    auto=True: the stream that is current when the batch is yielded waits for the arrival, and the buffer is released on
    that same stream when the next batch is requested (single-stream consumers).  auto=False: the consumer brackets its use
    with `wait(batch)` / `done(batch)` inside its own stream context (consumers that alternate streams)."""
    ring = None
    pending = None

    def start(batch):
        nonlocal ring
        fr = batch[key]
        if not isinstance(fr, torch.Tensor):
            fr = torch.as_tensor(np.asarray(fr))
        if fr.is_cuda:
            return dict(batch)
        if fr.dtype != torch.uint8:
            fr = fr.round().clamp_(0, 255).to(torch.uint8)
        if ring is None or ring.shape != tuple(fr.shape):
            ring = PinnedRing(fr.shape, device, depth)
        j = ring.next_slot()
        if fr.is_pinned():
            dev, ev = ring.upload(j, src=fr.contiguous())
            src = batch.get("_src")
            if src is not None:              # the producer's staging slot is busy until this copy has left it
                src.event = ev
        else:
            ring.host[j].copy_(fr)
            dev, ev = ring.upload(j)
        out = dict(batch)
        out.pop("_src", None)
        out[key] = dev
        out["_slot"] = (ring, j, ev)
        return out

    def emit(b):
        if auto:
            wait(b)
            slot = b.pop("_slot", None)
            yield b
            if slot is not None:
                slot[0].release(slot[1])
        else:
            yield b

    for batch in loader:
        nxt = start(batch)
        if pending is not None:
            yield from emit(pending)
        pending = nxt
    if pending is not None:
        yield from emit(pending)
