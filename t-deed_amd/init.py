"""Construction-time state of ``TDEEDModel.Impl`` as the reference builds it, and the mapping of a timm RegNetY
checkpoint onto it (VERDICT r2 "missing" 1).

The reference constructs (model/model.py:25-70)

* ``timm.create_model('regnety_002' | 'regnety_008', pretrained=True)`` -- ImageNet weights from the network; without
  them timm's own RegNet initialisation: Conv2d ``N(0, sqrt(2 / fan_out))`` with ``fan_out = kh*kw*out // groups``, conv
  biases 0 (the SE convs have biases), BatchNorm identity, and ``zero_init_last``: the last BatchNorm gain of every
  bottleneck (``conv3.bn.weight``) is 0 (timm 1.0.3 ``models/regnet.py`` ``_init_weights`` + ``Bottleneck.zero_init_last``;
  timm is not vendored under /root/reference and not installed here: restated from its published source, not pinned);
* ``make_temporal_shift`` (model/shift.py:46-59): ``_GSF`` keeps torch's default ``Conv3d`` / ``Conv2d`` initialisation
  (``U(+-1/sqrt(fan_in))`` for weight and bias, model/impl/gsf.py:17-24), ``_GSM`` zeroes its ``conv3D``
  (model/impl/gsm.py:73-76); ``BatchNorm3d`` identity;
* ``temp_enc ~ N(0, 1/clip_len)`` (model/model.py:65);
* ``EDSGPMIXERLayers`` (model/modules.py:58-66): LayerNorm / GroupNorm identity, every depthwise conv ``N(0, 0.1)`` with
  ZERO bias (``reset_params``, modules.py:146-157, 255-275), ``concat_fc`` ``N(0, 0.1)`` with zero bias (273-275), the
  ``mlp`` convs torch's default ``Conv1d`` initialisation;
* ``FCLayers`` heads: torch's default ``nn.Linear`` initialisation (modules.py:366-371).

Every BatchNorm buffer starts as running_mean 0, running_var 1, num_batches_tracked 0.  ``synth.make_state`` (perturbed
running statistics, non-zero depthwise biases) stays what the parity fixtures use; it is no longer the model's default.
"""
import math
import re

import torch

from . import state_layout

_DW = ("psi", "fc", "convw", "convkw", "global_fc")


def _uniform(shape, bound, g):
    return (torch.rand(shape, generator=g, dtype=torch.float32) * 2.0 - 1.0) * bound


def _normal(shape, std, g):
    return torch.randn(shape, generator=g, dtype=torch.float32) * std


def reference_init(shapes, cfg, generator=None):
    """name -> tensor for every entry of `shapes` (state_layout.model_state_shapes order), drawn like the reference's
    constructors draw them (distributions above; the draw ORDER is this file's own: the reference's exact stream would need
    timm).  generator: torch.Generator (CPU) or None for the global one."""
    g = generator
    arch = cfg["feature_arch"]
    gsm = arch.endswith("_gsm")
    out = {}
    weights = {}
    for key, (shape, dt) in shapes.items():
        shape = tuple(shape)
        last = key.rsplit(".", 1)[-1]
        mod = key.rsplit(".", 2)[-2] if key.count(".") >= 1 else ""
        mod_base = re.sub(r"\d+$", "", mod)
        if last == "num_batches_tracked":
            v = torch.zeros(shape, dtype=torch.int64)
        elif last == "running_mean":
            v = torch.zeros(shape)
        elif last == "running_var":
            v = torch.ones(shape)
        elif key == "temp_enc":
            v = _normal(shape, 1.0 / shape[0], g)
        elif re.search(r"(\.bn\.|\.ln\d?\.|\.gn\.)", "." + key):
            if last == "weight":
                v = torch.zeros(shape) if key.endswith(".conv3.bn.weight") else torch.ones(shape)   # zero_init_last
            else:
                v = torch.zeros(shape)
        elif key.startswith("_features.") and ".gs." not in key:
            # timm RegNet: convs N(0, sqrt(2/fan_out)), fan_out = kh*kw*out // groups; biases (SE) zero
            if last == "bias":
                v = torch.zeros(shape)
            else:
                cout, cin_g, kh, kw = shape
                groups = 1
                m = re.match(r"_features\.(s\d\.b\d+)\.conv2\.conv\.weight", key)
                if m:
                    groups = cout // cin_g
                v = _normal(shape, math.sqrt(2.0 / (kh * kw * cout // groups)), g)
        elif ".gs.conv3D." in key and gsm:
            v = torch.zeros(shape)                                     # gsm.py:75-76
        elif mod_base in _DW and (len(shape) == 1 or (len(shape) == 3 and shape[1] == 1)):
            v = torch.zeros(shape) if last == "bias" else _normal(shape, 0.1, g)           # reset_params
        elif mod == "concat_fc":
            v = torch.zeros(shape) if last == "bias" else _normal(shape, 0.1, g)
        else:
            # torch defaults of Conv1d / Conv2d / Conv3d / Linear: kaiming_uniform(a=sqrt(5)) = U(+-1/sqrt(fan_in)) for the
            # weight, U(+-1/sqrt(fan_in)) for the bias (fan_in of the module's weight)
            if last == "weight":
                fan_in = int(torch.tensor(shape[1:]).prod()) if len(shape) > 1 else shape[0]
                weights[key[:-len("weight")]] = fan_in
                v = _uniform(shape, 1.0 / math.sqrt(fan_in), g)
            else:
                fan_in = weights[key[:-len("bias")]]                    # the weight precedes its bias in the key order
                v = _uniform(shape, 1.0 / math.sqrt(fan_in), g)
        out[key] = v.to(torch.int64 if dt == "int64" else torch.float32)
    return out


def timm_key_map(cfg):
    """our `_features.*` key -> the key of a timm `regnety_002` / `regnety_008` state_dict that fills it.  Gate-shift
    wraps conv1 of every s3 / s4 block (shift.py:46-59: `blocks[i].conv1 = GatedShift(conv1)` -> `conv1.net.*`); the
    gate-shift's own tensors (`conv1.gs.*`) have no timm counterpart and timm's classifier `head.fc.*` has none here
    (model.py:45 replaces it by Identity)."""
    shapes = state_layout.model_state_shapes(cfg)
    m = {}
    for k in shapes:
        if not k.startswith("_features.") or ".gs." in k:
            continue
        m[k] = k[len("_features."):].replace(".conv1.net.", ".conv1.")
    return m


def map_timm_backbone(timm_sd, cfg):
    """-> {our key: tensor} for every trunk tensor, from a timm RegNetY state_dict (keys `stem.conv.weight`,
    `s3.b2.conv1.bn.running_var`, `s1.b1.se.fc1.bias`, ..., optionally `head.fc.*`, optionally prefixed `module.`).
    Raises on a missing or mis-shaped tensor and on unexpected trunk keys."""
    shapes = state_layout.model_state_shapes(cfg)
    sd = {(k[7:] if k.startswith("module.") else k): v for k, v in timm_sd.items()}
    used, out = set(), {}
    for ours, theirs in timm_key_map(cfg).items():
        if theirs not in sd:
            if theirs.endswith("num_batches_tracked"):
                continue                                                # older checkpoints lack the counter
            raise KeyError(f"timm checkpoint lacks {theirs} (wanted for {ours})")
        v = torch.as_tensor(sd[theirs])
        want = tuple(shapes[ours][0])
        if tuple(v.shape) != want:
            raise ValueError(f"{theirs}: shape {tuple(v.shape)} does not fit {ours} {want} -- wrong RegNetY size?")
        out[ours] = v
        used.add(theirs)
    extra = [k for k in sd if k not in used and not k.startswith(("head.fc.", "fc."))]
    if extra:
        raise KeyError(f"unexpected keys in the timm checkpoint: {extra[:5]}")
    return out
