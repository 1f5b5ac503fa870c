"""GPU tool: where the torch-side device copies / elementwise launches of one eager training step come from (everything
that is not a C-ABI call): torch profiler, grouped by aten op and calling line of this package.
    python tools/find_copies.py [workload] [B]"""
import collections
import os
import sys
import traceback
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
from tdeed_amd import synth, state_layout, ops
from tdeed_amd.trainer import TrainEngine
from tdeed_amd.regnet_spec import regnet_spec

wname = sys.argv[1] if len(sys.argv) > 1 else "rny002_b8"
wl = bench.CONFIGS[wname]
cfg, H, W = wl["cfg"], wl["H"], wl["W"]
B = int(sys.argv[2]) if len(sys.argv) > 2 else wl["B"]
T = cfg["clip_len"]
sd = {k: torch.from_numpy(v) for k, v in synth.make_state(state_layout.model_state_shapes(cfg), 0).items()}
eng = TrainEngine(cfg, sd, torch.bfloat16, "cuda", lr=1e-4)
frames = ops.fill_u8_hash((B, T, 3, H, W), 1000, "cuda")
lab_np, labD_np = synth.labels(5, B, T, cfg["num_classes"], max(cfg["radi_displacement"], 1))
lab = torch.from_numpy(lab_np).cuda()
labD = torch.from_numpy(labD_np).float().cuda() if cfg["radi_displacement"] else None
C = regnet_spec(cfg["feature_arch"]).feat_dim
masks = [((torch.rand((B, T, C), device="cuda") >= 0.5).to(torch.bfloat16) * 2.0) for _ in range(2 if cfg["radi_displacement"] else 1)]
for _ in range(2):
    eng.step(frames, lab, labD, drop_masks=masks)
torch.cuda.synchronize()

sites = collections.Counter()


class Spy(torch.utils._python_dispatch.TorchDispatchMode):
    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        name = str(func)
        if any(isinstance(a, torch.Tensor) and a.is_cuda for a in args):
            where = "?"
            for fs in reversed(traceback.extract_stack()):
                if "t-deed_amd" in fs.filename and "_lib.py" not in fs.filename:
                    where = f"{os.path.basename(fs.filename)}:{fs.lineno}"
                    break
            sites[(name, where)] += 1
        return func(*args, **(kwargs or {}))


with Spy():
    eng.step(frames, lab, labD, drop_masks=masks)
torch.cuda.synchronize()
skip = ("aten.view", "aten._unsafe_view", "aten.t.", "aten.transpose", "aten.slice", "aten.select", "aten.detach", "aten.alias",
        "aten.as_strided", "aten.unsqueeze", "aten.squeeze", "aten.expand", "aten.permute", "aten.reshape", "aten._reshape_alias",
        "aten.unbind", "aten.split")
tot = 0
for (name, where), n in sorted(sites.items(), key=lambda kv: -kv[1]):
    if name.startswith(skip):
        continue
    tot += n
    print(f"{n:5d}  {name:40s} {where}")
print("device-launching torch ops in one eager step:", tot)
