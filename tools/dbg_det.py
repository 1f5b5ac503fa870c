import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
from helpers import load_golden, model_state, t
from tdeed_amd import synth
from tdeed_amd.engine import ForwardEngine
meta, g = load_golden("tiny_rny002_gsf")
cfg = meta["cfg"]
sd = model_state(cfg, 0)
clip = synth.uint8_clip(5, (2, cfg["clip_len"], 3, 64, 64))
names = ["_features.s1.b1", "_features.s2.b1", "_features.s3.b1", "_features.s3.b2", "_features.s3.b3", "_features.s3.b4", "_features.s4.b1", "_features.s4.b2", "_features.s4.b7"]
def run(graph, taps):
    st = torch.cuda.Stream()
    with torch.cuda.stream(st):
        junk = torch.randn(50_000_000, device="cuda")
        eng = ForwardEngine(cfg, sd, torch.bfloat16, "cuda", use_graph=graph)
        head, plan = eng.forward(t(clip).to("cuda"), taps=taps)
        st.synchronize()
        return head.float().cpu(), {k: v.float().cpu() for k, v in plan.keep.items()}
a, ka = run(False, tuple(names))
b, kb = run(False, tuple(names))
print("eager vs eager head diff", float((a - b).abs().max()))
for k in names + ["feat", "sgp_out"]:
    d = (ka[k] - kb[k]).abs()
    print(f"{k:22s} maxdiff {float(d.max()):.5f}  nonzero frames {int((d.flatten(1).max(1).values > 0).sum())}")
