"""CPU diagnostic (uses the oracle as a calculator, not a product path): emulate the bf16 engine's rounding points in the
RegNetY trunk on the reference's golden clip and ask where the logit error of the throughput mode comes from, and what an
fp32 residual stream in stage s4 / s3 would buy -- BEFORE building it.  Rounding points of the engine (bneck.hip header):
conv operands (activations and weights) bf16, y1, y2, y2 * gate and the block output bf16, everything else fp32.
    python tools/diag_bf16_emul.py [golden name]"""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
import torch.nn.functional as F
from helpers import load_golden, model_state
from oracle import tdeed_oracle as O
from tdeed_amd import synth
from tdeed_amd.regnet_spec import regnet_spec

torch.set_num_threads(8)
name = ([a for a in sys.argv[1:] if not a.startswith("--")] or ["finediving_small"])[0]
meta, g = load_golden(name)
cfg = meta["cfg"]
sd = O.as_torch_state(model_state(cfg, meta["seed_w"]))
spec = regnet_spec(cfg["feature_arch"])
T = cfg["clip_len"]
clip = torch.from_numpy(synth.uint8_clip(meta["seed_x"], (meta["B"], T, 3, meta["H"], meta["W"])))
gold = torch.from_numpy(g["logits"]).float()
K1 = cfg["num_classes"] + 1


def r(x):
    return x.to(torch.bfloat16).float()


def conv_bn(x, pre, stride=1, groups=1, relu=True, rnd_out=True):
    w = r(sd[pre + ".conv.weight"])
    y = F.conv2d(r(x), w, None, stride=stride, padding=w.shape[-1] // 2, groups=groups)
    y = O._bn(y, sd, pre + ".bn", False)
    y = torch.relu(y) if relu else y
    return r(y) if rnd_out else y


TAPS = {}


def trunk(x, f32_residual_stages=()):
    x = conv_bn(x, "_features.stem", stride=2)
    TAPS["_features.stem"] = x
    for blk in spec.blocks:
        pre = "_features." + blk.name
        keep32 = blk.name.split(".")[0] in f32_residual_stages
        short = x
        if blk.gsf_fold > 0:
            Fd = blk.gsf_fold
            gs = r(O.gate_shift(r(x[:, :Fd]), sd, pre + ".conv1.gs", T, "gsf", False, None))
            y = conv_bn(torch.cat([gs, x[:, Fd:]], 1), pre + ".conv1.net")
        else:
            y = conv_bn(x, pre + ".conv1")
        y = conv_bn(y, pre + ".conv2", stride=blk.stride, groups=blk.groups)
        s = y.mean(dim=(2, 3), keepdim=True)
        s = torch.relu(F.conv2d(s, sd[pre + ".se.fc1.weight"], sd[pre + ".se.fc1.bias"]))
        s = torch.sigmoid(F.conv2d(s, sd[pre + ".se.fc2.weight"], sd[pre + ".se.fc2.bias"]))
        y = r(y * s)
        y = conv_bn(y, pre + ".conv3", relu=False, rnd_out=False)
        if blk.has_downsample:
            short = conv_bn(short, pre + ".downsample", stride=blk.stride, relu=False, rnd_out=False)
        out = torch.relu(y + short)
        x = out if keep32 else r(out)
        TAPS[pre] = x
    return x.mean(dim=(2, 3))


with torch.no_grad():
    x01 = O.preprocess(clip, cfg["crop_dim"], False)
    Bn = x01.shape[0]
    frames = x01.reshape(Bn * T, *x01.shape[2:])
    for label, stages in (("bf16 block outputs everywhere (the engine)", ()), ("fp32 residual stream in s4", ("s4",)),
                          ("fp32 residual stream in s3 + s4", ("s3", "s4")), ("fp32 residual stream everywhere", ("s1", "s2", "s3", "s4"))):
        feat = trunk(frames, stages).view(Bn, T, -1) + sd["temp_enc"][None]
        out = O.ed_sgp_mixer(feat, sd, cfg["n_layers"], T)
        logits, displ = O.heads(out, sd, cfg["radi_displacement"])
        e = (logits[..., :K1] - gold).abs()
        print(f"{label:48s} max-abs-err {float(e.max()):.4e}  rms {float(e.pow(2).mean().sqrt()):.4e}", flush=True)
        if not stages and "--taps" in sys.argv:
            ref = {}
            O.regnet_features(O.preprocess(clip, cfg["crop_dim"], False).reshape(Bn * T, *x01.shape[2:]), sd, spec, T, taps=ref)
            for n, v in TAPS.items():
                a = ref[n]
                d = a - v
                print(f"   {n:28s} rel rms err {float(d.pow(2).mean().sqrt() / a.pow(2).mean().sqrt()):.3e}   max abs {float(d.abs().max()):.3e}")
