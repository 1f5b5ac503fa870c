"""GPU diagnostic: relative error of every block output of the bf16 engine against the fp32 engine on the golden clip
(rms of the difference / rms of the fp32 map), to be read next to tools/diag_bf16_emul.py --taps (the same numbers for an
emulation that rounds to bf16 only where the engine's design says it does): a block where the engine's error jumps above
the emulation's has a rounding point, or an accumulation, the design does not need.
    python tools/diag_bf16_taps.py [golden name]"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
from helpers import load_golden, model_state, t
from tdeed_amd import synth
from tdeed_amd.engine import ForwardEngine
from tdeed_amd.regnet_spec import regnet_spec

name = sys.argv[1] if len(sys.argv) > 1 else "finediving_small"
meta, g = load_golden(name)
cfg = meta["cfg"]
sd = model_state(cfg, meta["seed_w"])
spec = regnet_spec(cfg["feature_arch"])
B, T = meta["B"], cfg["clip_len"]
clip = synth.uint8_clip(meta["seed_x"], (B, T, 3, meta["H"], meta["W"]))
names = ["_features.stem"] + ["_features." + b.name for b in spec.blocks]
st = torch.cuda.Stream()
maps = {}
with torch.cuda.stream(st):
    for dt in (torch.float32, torch.bfloat16):
        eng = ForwardEngine(cfg, sd, dt, "cuda", use_graph=False, n_split=1)
        head, plan = eng.forward(t(clip).to("cuda"), taps=tuple(names))
        st.synchronize()
        maps[dt] = {n: plan.keep[n].float().cpu() for n in names}
        maps[dt]["feat"] = plan.keep["feat"].float().cpu()
        maps[dt]["head"] = head.float().cpu()
for n in names + ["feat", "head"]:
    a, b = maps[torch.float32][n], maps[torch.bfloat16][n]
    d = (a - b)
    print(f"{n:28s} rel rms err {float(d.pow(2).mean().sqrt() / a.pow(2).mean().sqrt()):.3e}   max abs {float(d.abs().max()):.3e}  (map rms {float(a.pow(2).mean().sqrt()):.3f})")
