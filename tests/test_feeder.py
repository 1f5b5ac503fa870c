"""Input pipeline helpers (row f4): JPEG decode into staging buffers with the reference's padding rules (CPU), and the
pinned ring / prefetch wrapper on the GPU."""
import os

import numpy as np
import pytest
import torch

from helpers import ROOT  # noqa: F401
from tdeed_amd import feeder


def _write_frames(d, n, h=24, w=32):
    from PIL import Image
    rs = np.random.RandomState(0)
    for i in range(n):
        Image.fromarray(rs.randint(0, 256, (h, w, 3), dtype=np.uint8)).save(os.path.join(d, f"frame{i}.jpg"), quality=95)


def test_load_clip_decodes_and_pads_like_the_reference(tmp_path):
    """frame.py:558-626: frames before 0 pad the start (always), missing frames pad the end only when pad=True."""
    from PIL import Image
    d = str(tmp_path)
    _write_frames(d, 5)
    path = lambda n: os.path.join(d, f"frame{n}.jpg")        # noqa: E731
    ref = [torch.from_numpy(np.moveaxis(np.asarray(Image.open(path(i)).convert("RGB")), 2, 0).copy()) for i in range(5)]
    clip = feeder.load_clip(path, 0, 5)
    assert clip.dtype == torch.uint8 and clip.shape == (5, 3, 24, 32) and all(torch.equal(clip[i], ref[i]) for i in range(5))
    c2 = feeder.load_clip(path, -2, 4)                       # two frames before the video: zero padded at the start
    assert c2.shape[0] == 6 and int(c2[:2].sum()) == 0 and torch.equal(c2[2], ref[0]) and torch.equal(c2[5], ref[3])
    c3 = feeder.load_clip(path, 3, 8)                        # runs past the end
    assert c3.shape[0] == 2
    c4 = feeder.load_clip(path, 3, 8, pad=True)
    assert c4.shape[0] == 5 and int(c4[2:].sum()) == 0 and torch.equal(c4[1], ref[4])
    c5 = feeder.load_clip(path, 0, 6, stride=2)
    assert c5.shape[0] == 3 and torch.equal(c5[1], ref[2])
    assert feeder.load_clip(path, 10, 12) == -1
    # decoding straight into a staging buffer gives the same bytes
    buf = torch.zeros((6, 3, 24, 32), dtype=torch.uint8)
    out = feeder.load_clip(path, -2, 4, out=buf)
    assert torch.equal(out, c2)


@pytest.mark.gpu
def test_prefetch_delivers_the_same_batches_on_the_device():
    batches = [dict(frame=torch.randint(0, 256, (2, 4, 3, 16, 16), dtype=torch.uint8), label=torch.full((2, 4), i))
               for i in range(5)]
    batches[3]["frame"] = batches[3]["frame"].pin_memory()
    got = []
    for b in feeder.prefetch(batches, "cuda"):
        assert b["frame"].is_cuda
        got.append((b["frame"].cpu(), int(b["label"][0, 0])))
    assert [g[1] for g in got] == list(range(5))
    assert all(torch.equal(g[0], b["frame"].cpu()) for g, b in zip(got, batches))
    # explicit mode with alternating consumer streams
    sts = [torch.cuda.Stream(), torch.cuda.Stream()]
    outs = []
    for i, b in enumerate(feeder.prefetch(batches, "cuda", auto=False, depth=2)):
        with torch.cuda.stream(sts[i % 2]):
            feeder.wait(b)
            outs.append(b["frame"].float().sum())
            feeder.done(b)
    torch.cuda.synchronize()
    assert [float(o) for o in outs] == [float(b["frame"].float().sum()) for b in batches]


def _golden_frames():
    import json
    g = np.load(os.path.join(ROOT, "tests", "golden", "frame_reader.npz"))
    return json.loads(str(g["meta"])), g, os.path.join(ROOT, "tests", "golden", "frames")


def test_clip_readers_match_the_reference_on_the_committed_jpeg_directories():
    """The fixture holds what the reference's FrameReaderVideo.load_frames / FrameReader.load_paths + load_frames returned
    (tools/make_goldens.py:frame_reader, dataset/frame.py:263-382, 546-626) for four dataset layouts x eight
    (start, end, stride, pad) spans on tests/golden/frames/: file naming, start / end padding, stride, missing files."""
    meta, g, base = _golden_frames()
    assert len(meta["cases"]) == 32
    pool = feeder.DecodePool(4)
    for c in meta["cases"]:
        ds, si = c["dataset"], meta["source_info"].get(c["dataset"])
        fdir = os.path.join(base, ds)
        want = g[c["video_key"]]
        got = feeder.load_clip_video(fdir, ds, c["video"], c["start"], c["end"], pad=c["pad"], stride=c["stride"], source_info=si)
        if want.shape == ():
            assert got == -1, c
        else:
            assert got.dtype == torch.uint8 and np.array_equal(got.numpy(), want), c
        paths = feeder.load_paths(fdir, ds, c["video"], c["start"], c["end"], stride=c["stride"], source_info=si)
        assert [os.path.relpath(paths[0], base)] + paths[1:] == c["paths"], (c, paths)
        if "train_key" in c:
            tr = feeder.load_frames(paths, pad=c["pad"], stride=c["stride"])
            assert np.array_equal(tr.numpy(), g[c["train_key"]]), c
            # decoding through the thread pool into a (larger) staging buffer gives the same clip
            buf = torch.full((12, 3, meta["h"], meta["w"]), 7, dtype=torch.uint8)
            tr2 = feeder.load_frames(paths, pad=c["pad"], stride=c["stride"], out=buf, pool=pool)
            assert np.array_equal(tr2.numpy(), g[c["train_key"]]), c
    pool.close()


@pytest.mark.gpu
def test_clip_batches_decodes_whole_batches_into_pinned_slots():
    meta, g, base = _golden_frames()
    cs = [c for c in meta["cases"] if "train_key" in c and g[c["train_key"]].shape[0] == 7 and c["stride"] == 1]
    assert len(cs) >= 4
    clips = [dict(paths=[os.path.join(base, c["paths"][0])] + c["paths"][1:], stride=1, tag=i) for i, c in enumerate(cs[:4])]
    out = list(feeder.clip_batches(clips, 2, (3, meta["h"], meta["w"]), 7, depth=2))
    assert len(out) == 2 and out[0]["tag"] == [0, 1] and out[1]["frame"].is_pinned()
    for b, lo in zip(out, (0, 2)):
        for i in range(2):
            assert np.array_equal(b["frame"][i].numpy(), g[cs[lo + i]["train_key"]])


def test_process_decode_pool_matches_read_frame(tmp_path):
    """feeder.ProcessDecodePool (decode worker PROCESSES writing into shared-memory staging slots: the role of the reference's
    DataLoader workers, train_tdeed.py:131-139) delivers the same bytes as read_frame, reports a missing file as an error,
    and leaves neither processes nor shared-memory blocks behind."""
    import numpy as np
    import torch
    from PIL import Image
    from tdeed_amd import feeder
    rs = np.random.RandomState(3)
    for i in range(12):
        Image.fromarray(rs.randint(0, 256, (48, 64, 3), dtype=np.uint8)).save(tmp_path / f"frame{i}.jpg", quality=92)
    ref = torch.stack([feeder.read_frame(str(tmp_path / f"frame{i}.jpg")) for i in range(12)])
    pool = feeder.ProcessDecodePool(2)
    try:
        clips = [dict(paths=[str(tmp_path), 0, 0, 0, -1, 12], stride=1, label=7)] * 4
        n = 0
        for batch in feeder.clip_batches(clips, 2, (3, 48, 64), 12, pool=pool, depth=2):
            assert batch["label"] == [7, 7]
            assert torch.equal(batch["frame"][0], ref) and torch.equal(batch["frame"][1], ref)
            n += 1
        assert n == 2
        slot = pool.make_slots(1, (1, 1, 3, 48, 64))[0]
        with pytest.raises(RuntimeError, match="FileNotFoundError"):
            pool.decode([(str(tmp_path / "missing.jpg"), slot[0, 0])])
        names = [shm.name for shm, *_ in pool._shms]
        procs = list(pool._w)
    finally:
        pool.close()
    assert all(p.poll() is not None for p in procs)
    import os
    assert not any(os.path.exists("/dev/shm/" + nm.lstrip("/")) for nm in names)


def test_a_dead_decode_worker_fails_its_jobs_at_once_and_leaves_the_rotation(tmp_path):
    """when a worker process dies, the jobs it still owed fail immediately with a clear error (no 120 s timeout per
    result), the worker leaves the round-robin, the remaining workers keep serving, and with none left submit raises."""
    import time
    from PIL import Image
    rs = np.random.RandomState(5)
    for i in range(4):
        Image.fromarray(rs.randint(0, 256, (16, 16, 3), dtype=np.uint8)).save(tmp_path / f"frame{i}.jpg", quality=92)
    pool = feeder.ProcessDecodePool(2)
    try:
        slot = pool.make_slots(1, (4, 3, 16, 16))[0]
        pool.decode([(str(tmp_path / f"frame{i}.jpg"), slot[i]) for i in range(4)])          # both workers alive
        pool._w[0].kill()
        pool._w[0].wait(timeout=10)
        t0 = time.time()
        while pool._alive[0] and time.time() - t0 < 10:                                      # the reader thread sees EOF
            time.sleep(0.05)
        assert not pool._alive[0] and pool._alive[1]
        ref = torch.stack([feeder.read_frame(str(tmp_path / f"frame{i}.jpg")) for i in range(4)])
        slot.zero_()
        t0 = time.time()
        pool.decode([(str(tmp_path / f"frame{i}.jpg"), slot[i]) for i in range(4)])          # all four go to worker 1
        assert time.time() - t0 < 30 and torch.equal(slot, ref)
        # a job outstanding on a worker that dies is failed by the reader thread, at once
        pool._w[1].stdin.write("")                                                          # (pipe still open)
        acks = pool.submit([(str(tmp_path / "frame0.jpg"), slot[0])])
        pool._w[1].kill()
        t0 = time.time()
        try:
            acks[0].result(timeout=30)
            finished = True                                                                  # it had answered before the kill
        except RuntimeError as e:
            finished = False
            assert "exited" in str(e) or "gone" in str(e)
        assert time.time() - t0 < 20, finished
        t0 = time.time()
        while pool._alive[1] and time.time() - t0 < 10:
            time.sleep(0.05)
        with pytest.raises(RuntimeError, match="no live decode worker"):
            pool.submit([(str(tmp_path / "frame0.jpg"), slot[0])])
    finally:
        pool.close()


def test_clip_batches_waits_for_the_upload_that_last_read_a_slot(tmp_path):
    """a staging slot is decoded into again only after the event the consumer (prefetch) left in the batch's `_src`
    holder has completed; rows behind a clip's real frames are zeroed even with pad=False."""
    from PIL import Image
    rs = np.random.RandomState(6)
    for i in range(3):
        Image.fromarray(rs.randint(1, 256, (16, 16, 3), dtype=np.uint8)).save(tmp_path / f"frame{i}.jpg", quality=92)

    class Ev:
        def __init__(self):
            self.waited = 0

        def synchronize(self):
            self.waited += 1

    # every clip: 3 real frames + 1 missing at the end (pad_end = 1), clip_len 4
    clips = [dict(paths=[str(tmp_path), 0, 0, 1, -1, 4], stride=1)] * 6
    evs = []
    gen = feeder.clip_batches(clips, 1, (3, 16, 16), 4, pool=feeder.DecodePool(2), depth=2, pad=False)
    for bi, b in enumerate(gen):
        assert int(b["frame"][0, 3].sum()) == 0 and int(b["frame"][0, 2].sum()) > 0         # tail row zeroed, real rows decoded
        b["frame"][0, 3].fill_(9)                                                           # stale content a later batch must not show
        ev = Ev()
        b["_src"].event = ev                                                                # what prefetch does after ring.upload
        evs.append(ev)
    assert len(evs) == 6
    # slots rotate with depth 2: the events of batches 0..3 were waited for exactly once (before batches 2..5 decoded)
    assert [e.waited for e in evs] == [1, 1, 1, 1, 0, 0]
