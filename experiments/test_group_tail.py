"""Parked with experiments/group_tail.py (run with: python -m pytest experiments/test_group_tail.py on a GPU box)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "experiments")]
import numpy as np
import torch
from helpers import load_golden, model_state, t, max_abs
from tdeed_amd import synth
from tdeed_amd.engine import ForwardEngine
import group_tail
group_tail.attach(ForwardEngine)
DEV = "cuda"


def test_one_temporal_stage_per_group_of_batches_in_flight_equals_single_clip_forwards():
    """The default shape of bench.py (ForwardEngine.plan_group): three batches of 8 clips in flight, each trunk a HIP graph of
    its own on its own stream, ONE SGP encoder-decoder + heads per group of 24 clips on a fourth stream.  Over several groups
    with changing clips every slot's logits equal the B=1 forwards of its clips (bf16 <= 2e-2) and the golden clip's the
    reference's logits; a region that stops in the middle of a group is flushed (model/model.py:105-149, modules.py:69-87)."""
    from tdeed_amd.engine import ForwardEngine
    meta, g = load_golden("finediving_small")
    cfg = meta["cfg"]
    sd = model_state(cfg, meta["seed_w"])
    T, H, W = cfg["clip_len"], meta["H"], meta["W"]
    B, depth = 8, 3
    K1 = cfg["num_classes"] + 1
    streams = [torch.cuda.Stream() for _ in range(depth)]
    mk = lambda s: np.concatenate([synth.uint8_clip(meta["seed_x"] + 100 * s + i, (1, T, 3, H, W)) for i in range(B)], 0)  # noqa: E731
    clips = [mk(s) for s in range(depth + 2)]
    with torch.cuda.stream(streams[0]):
        eng = ForwardEngine(cfg, sd, torch.bfloat16, DEV, n_split=1)
        grp = eng.plan_group(B, H, W, depth)
        for i in range(depth):
            eng.set_group_frames(grp, i, t(clips[i]).to(DEV))
    torch.cuda.synchronize()
    for rep in range(4):                                             # whole groups, as bench.py's run(n) issues them
        for i in range(depth):
            with torch.cuda.stream(streams[i]):
                eng.run_group_slot(grp, i)
    # new clips for slots 0 and 1, and a region that ends after slot 1: the flush runs the stage
    torch.cuda.synchronize()
    with torch.cuda.stream(streams[0]):
        eng.set_group_frames(grp, 0, t(clips[3]).to(DEV))
        eng.set_group_frames(grp, 1, t(clips[4]).to(DEV))
    torch.cuda.synchronize()
    for i in range(2):
        with torch.cuda.stream(streams[i]):
            eng.run_group_slot(grp, i)
    eng.flush_group(grp)
    torch.cuda.synchronize()
    heads = grp.head_out.float().cpu().view(depth, B, T, -1).clone()
    assert torch.isfinite(heads).all()
    want = [clips[3], clips[4], clips[2]]                            # slot 2 keeps the rows of its last issue
    eng1 = ForwardEngine(cfg, sd, torch.bfloat16, DEV, n_split=1)
    st = torch.cuda.Stream()
    for s in range(depth):
        for i in (0, 5):
            with torch.cuda.stream(st):
                h1, _ = eng1.forward(t(want[s][i:i + 1]).to(DEV))
                st.synchronize()
            assert max_abs(h1.float().cpu().view(T, -1), heads[s][i]) <= 2e-2, (s, i)
    # the golden clip sat in slot 0 of the first groups: run one more whole group with it back in place
    with torch.cuda.stream(streams[0]):
        eng.set_group_frames(grp, 0, t(clips[0]).to(DEV))
    torch.cuda.synchronize()
    for i in range(depth):
        with torch.cuda.stream(streams[i]):
            eng.run_group_slot(grp, i)
    torch.cuda.synchronize()
    h0 = grp.head_out.float().cpu().view(depth, B, T, -1)[0, 0, :, :K1]
    assert max_abs(h0, g["logits"][0]) < 0.08 * max(1.0, float(np.abs(g["logits"]).max()))
