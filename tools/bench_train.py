"""GPU tool: time one optimisation step (train-mode forward + loss + backward + AdamW) of a bench workload.
    python tools/bench_train.py [workload] [B] [steps]"""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
from tdeed_amd import synth, state_layout, ops
from tdeed_amd.trainer import TrainEngine
from tdeed_amd.regnet_spec import regnet_spec

wl = bench.CONFIGS[sys.argv[1] if len(sys.argv) > 1 else "rny002_b8"]
cfg, H, W = wl["cfg"], wl["H"], wl["W"]
B = int(sys.argv[2]) if len(sys.argv) > 2 else wl["B"]
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
T = cfg["clip_len"]
sd = {k: torch.from_numpy(v) for k, v in synth.make_state(state_layout.model_state_shapes(cfg), 0).items()}
eng = TrainEngine(cfg, sd, torch.bfloat16, "cuda", lr=1e-4)
frames = ops.fill_u8_hash((B, T, 3, H, W), 1000, "cuda")
lab_np, labD_np = synth.labels(5, B, T, cfg["num_classes"], max(cfg["radi_displacement"], 1))
lab = torch.from_numpy(lab_np).cuda()
labD = torch.from_numpy(labD_np).float().cuda() if cfg["radi_displacement"] else None
C = regnet_spec(cfg["feature_arch"]).feat_dim
nh = 2 if cfg["radi_displacement"] else 1
masks = [((torch.rand((B, T, C), device="cuda") >= 0.5).to(torch.bfloat16) * 2.0) for _ in range(nh)]
for i in range(steps + 1):
    torch.cuda.synchronize()
    t0 = time.time()
    loss = eng.step(frames, lab, labD, drop_masks=masks)
    torch.cuda.synchronize()
    dt = time.time() - t0
    print(f"step {i}: {dt*1e3:9.1f} ms  loss {float(loss[0]):.4f}  ({B/dt:.1f} clips/s)  mem {torch.cuda.max_memory_allocated()/2**30:.1f} GiB", flush=True)
