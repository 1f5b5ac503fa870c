import sys, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np, torch
from helpers import load_golden, model_state, t, cfg_ns
from tdeed_amd import synth, augment
from tdeed_amd.model import TDEEDModel
meta, g = load_golden("tiny_rny002_gsf")
cfg = meta["cfg"]
for seed in (0, 1, 2):
    for double in (True, False):
        torch.manual_seed(seed)
        m = TDEEDModel(device="cuda", args=cfg_ns(cfg))
        m.load({k: t(v) for k, v in model_state(cfg, meta["seed_w"]).items()})
        k1a, k1b = cfg["num_classes"] + 1, 6
        B, T = meta["B"], cfg["clip_len"]
        clip = synth.uint8_clip(meta["seed_x"], (B, T, 3, meta["H"], meta["W"]))
        if double:
            m._model.update_pred_head([k1a, k1b])
            ds = [1, 2][:B] if B >= 2 else [2]
            labs = [synth.labels(30 + i, 1, T, (k1a if ds[i] == 1 else k1b) - 1, cfg["radi_displacement"], fg_frac=0.3) for i in range(B)]
            lab = np.concatenate([l[0] for l in labs], 0); labD = np.concatenate([l[1] for l in labs], 0)
            loader = [dict(frame=t(clip), label=t(lab), labelD=t(labD), dataset=torch.tensor(ds))]
        else:
            lab, labD = synth.labels(3, B, T, cfg["num_classes"], cfg["radi_displacement"], fg_frac=0.3)
            loader = [dict(frame=t(clip), label=t(lab), labelD=t(labD))]
        m._model.augment_fn = augment.crop_only
        opt, _ = m.get_optimizer({"lr": 3e-4})
        losses = [m.epoch(loader, optimizer=opt) for _ in range(20)]
        print(seed, double, B, T, [round(x, 2) for x in losses], flush=True)
