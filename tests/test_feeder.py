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
