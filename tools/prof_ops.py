"""GPU tool: which host-side torch calls of one eager training step launch device work, grouped by the source line
that issued them (finds the small casts / fills / copies that surround the HIP kernels).
    python tools/prof_ops.py [workload] [B] [top]"""
import sys, os, collections
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from torch.utils._python_dispatch import TorchDispatchMode
import traceback
import bench
from tdeed_amd import synth, state_layout, ops
from tdeed_amd.trainer import TrainEngine
from tdeed_amd.regnet_spec import regnet_spec

wl = bench.CONFIGS[sys.argv[1] if len(sys.argv) > 1 else "rny002_b8"]
cfg, H, W = wl["cfg"], wl["H"], wl["W"]
B = int(sys.argv[2]) if len(sys.argv) > 2 else wl["B"]
top = int(sys.argv[3]) if len(sys.argv) > 3 else 60
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
eng.step(frames, lab, labD, drop_masks=masks)

VIEW = ("view", "reshape", "_unsafe_view", "slice", "select", "transpose", "permute", "t.", "as_strided", "expand",
        "unsqueeze", "squeeze", "detach", "alias", "empty", "_local_scalar", "unbind", "split", "narrow", "size", "stride",
        "is_", "sym_", "_to_copy_noop", "lift_fresh", "empty_like", "empty_strided", "new_empty", "flatten", "unflatten")
counts = collections.Counter()


class Mode(TorchDispatchMode):
    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        name = func.__name__ if hasattr(func, "__name__") else str(func)
        full = str(func)
        if not any(full.split("aten.")[-1].startswith(v) for v in VIEW):
            st = traceback.extract_stack()
            where = "?"
            for fr in reversed(st):
                if "t-deed_amd" in fr.filename and "prof_ops" not in fr.filename:
                    where = f"{os.path.basename(fr.filename)}:{fr.lineno}"
                    break
            counts[(full, where)] += 1
        return func(*args, **(kwargs or {}))


with Mode():
    eng.step(frames, lab, labD, drop_masks=masks)
torch.cuda.synchronize()
tot = sum(counts.values())
print(f"{tot} non-view aten calls in one step")
for (f, w), n in counts.most_common(top):
    print(f"{n:5d}  {f:40s} {w}")
