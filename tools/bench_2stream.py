"""Experiment: one batch of 8 clips as two concurrent half-batches on two streams (graph replay each)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from tdeed_amd import synth, state_layout, ops
from tdeed_amd.engine import ForwardEngine
wl = bench.CONFIGS["rny002_b8"]
cfg, H, W = wl["cfg"], wl["H"], wl["W"]
sd = synth.make_state(state_layout.model_state_shapes(cfg), 0)
def setup(B, stream):
    with torch.cuda.stream(stream):
        eng = ForwardEngine(cfg, sd, torch.bfloat16, "cuda", n_split=1)
        plan = eng.plan(B, H, W)
        eng.set_frames(plan, ops.fill_u8_hash((B, 100, 3, H, W), 1000, "cuda"))
        for _ in range(3): eng.run_plan(plan)
        stream.synchronize()
    return eng, plan
for nsplit in (1, 2, 4):
    streams = [torch.cuda.Stream() for _ in range(nsplit)]
    eps = [setup(8 // nsplit, s) for s in streams]
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    steps = 20
    for _ in range(steps):
        for s, (eng, plan) in zip(streams, eps):
            with torch.cuda.stream(s):
                eng.run_plan(plan)
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    print(f"{nsplit} stream(s) x {8 // nsplit} clips: {el / steps * 1e3:.3f} ms/step  {8 * steps / el:.0f} clips/s")
