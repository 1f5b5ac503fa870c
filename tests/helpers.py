"""Shared test plumbing: golden fixtures, synthetic states, oracle access."""
import json
import os
import sys
from collections import OrderedDict
from types import SimpleNamespace

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
import tdeed_amd  # noqa: E402,F401
from tdeed_amd import synth, state_layout  # noqa: E402
from tdeed_amd.regnet_spec import regnet_spec, sgp_up_size  # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")


def load_golden(name):
    z = np.load(os.path.join(GOLD, name + ".npz"))
    meta = json.loads(str(z["meta"]))
    return meta, {k: z[k] for k in z.files if k != "meta"}


def act(seed, name, shape, scale=1.0):
    n = int(np.prod(shape))
    return (synth.normalish(seed, name, n) * scale).reshape(shape).astype(np.float32)


def cfg_ns(cfg):
    return SimpleNamespace(modality="rgb", temporal_arch="ed_sgp_mixer", pretrain=None, **cfg)


def model_state(cfg, seed=0):
    shapes = state_layout.model_state_shapes(cfg)
    return synth.make_state(shapes, seed)


def module_state(mkind, prefix, mseed, **kw):
    kind, seed = mkind, mseed
    d = OrderedDict()
    if kind == "sgp_block":
        state_layout._sgp_block(d, prefix, kw["C"], kw["ks"], sgp_up_size(kw["ks"], kw["r"]))
    elif kind == "sgp_mixer":
        state_layout._sgp_mixer(d, prefix, kw["C"], kw["ks"], sgp_up_size(kw["ks"], kw["r"]))
    elif kind == "pyramid":
        up = sgp_up_size(kw["ks"], kw["r"])
        for i in range(2 * kw["n"] + 1):
            state_layout._sgp_block(d, f"{prefix}._sgp.{i}", kw["C"], kw["ks"], up)
        for i in range(kw["n"]):
            state_layout._sgp_mixer(d, f"{prefix}._sgpMixer.{i}", kw["C"], kw["ks"], up)
    elif kind == "gate_shift":
        state_layout._gate_shift(d, prefix, kw["F"], kw["mode"])
    elif kind == "ln":
        d[prefix + ".weight"] = ((1, kw["C"], 1), "float32")
        d[prefix + ".bias"] = ((1, kw["C"], 1), "float32")
    else:
        raise KeyError(kind)
    return synth.make_state(d, seed)


def t(x):
    x = np.asarray(x)
    return torch.from_numpy(x.copy() if x.ndim == 0 else np.ascontiguousarray(x))


def max_abs(a, b):
    a = a.detach().double().cpu().numpy() if isinstance(a, torch.Tensor) else np.asarray(a)
    b = b.detach().double().cpu().numpy() if isinstance(b, torch.Tensor) else np.asarray(b)
    return float(np.max(np.abs(a.astype(np.float64) - b.astype(np.float64)))) if a.size else 0.0


# ----------------------------------------------------------------------------- training reference on the CPU oracle
def drop_masks(seed, B, T, C, n):
    """the recorded dropout keep-masks of the train fixtures (tools/make_goldens.py drop_masks): 0 / 2, class head first"""
    return [t(((synth.normalish(seed + i, "dropmask", B * T * C) > 0).astype(np.float32) * 2.0).reshape(B, T, C))
            for i in range(n)]


def oracle_train_loss(frames, sd, cfg, spec, lab, labD, masks, crop, flip, soft=None):
    """the training-branch forward of TDEEDModel on the CPU oracle (batch-statistics BatchNorm, dropout masks = [class
    head, displacement head], one shared crop) and the loss of epoch() (model/model.py:208-211, 308-319).
    frames: uint8 or fp32 0..255 (mixup batches).  Returns (loss, logits, displ)."""
    from oracle import tdeed_oracle as O
    x = frames.float() / 255.0
    if crop is not None:
        top, left, ch, cw = crop
        x = x[..., top:top + ch, left:left + cw]
    if flip:
        x = x.flip(-1)
    mean = torch.tensor(O.IMAGENET_MEAN).view(1, 1, 3, 1, 1)
    std = torch.tensor(O.IMAGENET_STD).view(1, 1, 3, 1, 1)
    x = (x - mean) / std
    B, T = x.shape[:2]
    mode = "gsm" if cfg["feature_arch"].endswith("_gsm") else "gsf"
    f = O.regnet_features(x.reshape(B * T, *x.shape[2:]), sd, spec, T, mode, training=True)
    f = f.reshape(B, T, -1) + sd["temp_enc"][None]
    enc = O.ed_sgp_mixer(f, sd, cfg["n_layers"], cfg["clip_len"])
    dm = None if masks is None else (masks[1] if len(masks) > 1 else None, masks[0])
    cls, displ = O.heads(enc, sd, cfg["radi_displacement"], drop_mask=dm)
    return O.loss_fn(cls, lab if soft is None else soft, displ, labD), cls, displ


def sample_flat(a, cap=8192):
    """what a train fixture keeps of a tensor (tools/make_goldens.py sample_flat)"""
    a = np.asarray(a).reshape(-1)
    stride = 1 if a.size <= cap else a.size // 4096 + 1
    return a[::stride]


def chained_scheduler(opt, warm_steps, cos_steps):
    """train_tdeed.py:79-87"""
    from torch.optim.lr_scheduler import ChainedScheduler, LinearLR, CosineAnnealingLR
    return ChainedScheduler([LinearLR(opt, start_factor=0.01, end_factor=1.0, total_iters=warm_steps),
                             CosineAnnealingLR(opt, cos_steps)])
