"""GPU diagnostic: where the bf16 engine's logit error comes from -- trunk (uint8 -> pooled features) vs temporal stage (SGP
encoder-decoder + heads): the four combinations of {fp32, bf16} trunk x {fp32, bf16} temporal stage on the golden clip.
    python tools/diag_bf16_split.py [golden name]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch, numpy as np
from helpers import load_golden, model_state, t
from tdeed_amd import synth, ops
from tdeed_amd.engine import ForwardEngine, SgpBuilder, _Pool

name = sys.argv[1] if len(sys.argv) > 1 else "finediving_small"
meta, g = load_golden(name)
cfg = meta["cfg"]
sd = model_state(cfg, meta["seed_w"])
B, T = meta["B"], cfg["clip_len"]
clip = synth.uint8_clip(meta["seed_x"], (B, T, 3, meta["H"], meta["W"]))
K1 = cfg["num_classes"] + 1
gold = torch.from_numpy(g["logits"]).float()
st = torch.cuda.Stream()
with torch.cuda.stream(st):
    engs = {dt: ForwardEngine(cfg, sd, dt, "cuda", use_graph=False, n_split=1) for dt in (torch.float32, torch.bfloat16)}
    feats = {}
    for dt, eng in engs.items():
        head, plan = eng.forward(t(clip).to("cuda"))
        st.synchronize()
        feats[dt] = plan.keep["feat"].clone()
        print(f"full {str(dt):16s} max-abs-err vs reference logits {(head.float().cpu().view(B, T, -1)[..., :K1] - gold).abs().max():.4e}")
    for dt_trunk in engs:
        for dt_tail, eng in engs.items():
            pw, Wt = eng.pw, eng.pw.W
            pool, steps, keep = _Pool("cuda"), [], {}
            feat = feats[dt_trunk].to(dt_tail).contiguous()
            sb = SgpBuilder(pool, steps, keep, set(), B, dt_tail)
            cur = sb.pyramid(feat, T, pw.n_layers, Wt.sgp, Wt.mixer)
            head_out = torch.empty((B * T, pw.n_out), dtype=torch.float32, device="cuda")
            for s in steps:
                s.fn()
            ops.heads(cur, Wt.head_w, Wt.head_b, out=head_out)
            st.synchronize()
            e = (head_out.cpu().view(B, T, -1)[..., :K1] - gold).abs()
            print(f"trunk {str(dt_trunk):15s} temporal {str(dt_tail):15s} max-abs-err {e.max():.4e}  rms {e.pow(2).mean().sqrt():.4e}")
