"""GPU tool: per-launch device time of a plan (HIP events on the launch stream), sorted by time.
    python tools/profile_steps.py [workload] [dtype]"""
import sys, os, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
from tdeed_amd import synth, state_layout, ops
from tdeed_amd.engine import ForwardEngine

wl = bench.CONFIGS[sys.argv[1] if len(sys.argv) > 1 else "rny002_b8"]
dt = torch.float32 if (len(sys.argv) > 2 and sys.argv[2] == "f32") else torch.bfloat16
cfg, B, H, W = wl["cfg"], wl["B"], wl["H"], wl["W"]
sd = synth.make_state(state_layout.model_state_shapes(cfg), 0)
st = torch.cuda.Stream()
with torch.cuda.stream(st):
    eng = ForwardEngine(cfg, sd, dt, "cuda", use_graph=False)
    plan = eng.plan(B, H, W)
    eng.set_frames(plan, ops.fill_u8_hash((B, cfg["clip_len"], 3, H, W), 1000, "cuda"))
    reps = 5
    acc = [0.0] * len(plan.steps)
    for r in range(reps + 1):
        for _ in range(3):            # keep the GPU busy so the host runs ahead: event gaps = device time only
            for s in plan.steps:
                s.fn()
        evs = []
        for s in plan.steps:
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(st); s.fn(); b.record(st)
            evs.append((a, b))
        st.synchronize()
        if r:
            for i, (a, b) in enumerate(evs):
                acc[i] += a.elapsed_time(b) / reps
tot = sum(acc)
print(f"total {tot:.3f} ms over {len(plan.steps)} launches; pool {plan.pool_bytes/2**30:.2f} GiB")
rows = sorted(zip(acc, plan.steps), key=lambda x: -x[0])
for ms, s in rows[:int(os.environ.get("TOP", "45"))]:
    print(f"{s.name:34s} {s.kernel:14s} {ms*1e3:9.1f} us  {s.bytes/1e6:9.1f} MB {s.bytes/ms/1e6 if ms else 0:8.1f} GB/s  {s.flops/ms/1e9 if ms else 0:8.2f} TF/s")
