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
