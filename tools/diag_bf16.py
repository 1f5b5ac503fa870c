"""GPU diagnostic: per-layer relative error of the bf16 engine against the fp32 engine."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch, numpy as np
from helpers import load_golden, model_state, t
from tdeed_amd import synth
from tdeed_amd.engine import ForwardEngine
from tdeed_amd.regnet_spec import regnet_spec

name = sys.argv[1] if len(sys.argv) > 1 else "finediving_small"
meta, g = load_golden(name)
cfg = meta["cfg"]
sd = model_state(cfg, meta["seed_w"])
clip = synth.uint8_clip(meta["seed_x"], (meta["B"], cfg["clip_len"], 3, meta["H"], meta["W"]))
spec = regnet_spec(cfg["feature_arch"])
n = cfg["n_layers"]
names = ["_features.stem"] + ["_features." + b.name for b in spec.blocks] + \
        [f"_temp_fine._sgp.{i}" for i in range(2 * n + 1)] + [f"_temp_fine._sgpMixer.{i}" for i in range(n)]
st = torch.cuda.Stream()
res = {}
with torch.cuda.stream(st):
    for dt in (torch.float32, torch.bfloat16):
        eng = ForwardEngine(cfg, sd, dt, "cuda", use_graph=False)
        head, plan = eng.forward(t(clip).to("cuda"), taps=tuple(names))
        st.synchronize()
        res[dt] = {k: v.float().cpu() for k, v in plan.keep.items()}
        res[dt]["head"] = head.float().cpu()
for k in names + ["feat", "sgp_out", "head"]:
    a, b = res[torch.float32][k], res[torch.bfloat16][k]
    print(f"{k:28s} max|ref| {a.abs().max():10.4f} rms {a.pow(2).mean().sqrt():9.4f}  max err {(a-b).abs().max():9.4f}  rel-rms {((a-b).pow(2).mean().sqrt()/a.pow(2).mean().sqrt()):.4f}")
