"""Pin the CPU oracle (oracle/tdeed_oracle.py) against vectors produced by the reference itself
(tools/make_goldens.py, run in the build container).  CPU only."""
import numpy as np
import pytest
import torch

from helpers import load_golden, act, model_state, module_state, t, max_abs, cfg_ns
from tdeed_amd import state_layout, synth
from tdeed_amd.regnet_spec import regnet_spec
from oracle import tdeed_oracle as O

TOL = 1e-4      # fp32 noise floor of the reference logits is ~3e-6


def test_misc_ops():
    meta, g = load_golden("misc_ops")
    sd = O.as_torch_state(module_state("ln", "ln", meta["seed"], C=48))
    x = t(act(meta["seed"], "ln:x", (2, 48, 25)))
    assert max_abs(O.channel_layernorm(x, sd["ln.weight"], sd["ln.bias"]), g["ln_y"]) < 1e-5
    for (L, o) in [(25, 13), (125, 63), (100, 50), (13, 7)]:
        x = t(act(meta["seed"], f"pool{L}:x", (2, 16, L)))
        assert np.array_equal(O.adaptive_max_pool(x, o).numpy(), g[f"pool_{L}_{o}"])
    for (lo, hi) in [(13, 25), (25, 50), (63, 125), (50, 100)]:
        x = t(act(meta["seed"], f"up{lo}:x", (2, 16, lo)))
        assert max_abs(O.upsample_linear(x, hi), g[f"up_{lo}_{hi}"]) < 1e-6


@pytest.mark.parametrize("name", ["sgp_block_c32_t25", "sgp_block_c368_t100", "sgp_block_c48_t13"])
def test_sgp_block(name):
    meta, g = load_golden(name)
    sd = O.as_torch_state(module_state("sgp_block", "blk", meta["seed"], **meta))
    x = t(act(meta["seed"], name + ":x", (meta["B"], meta["C"], meta["T"])))
    assert max_abs(O.sgp_block(x, sd, "blk"), g["y"]) < TOL


@pytest.mark.parametrize("name", ["sgp_mixer_c32_t25", "sgp_mixer_c368_t100"])
def test_sgp_mixer(name):
    meta, g = load_golden(name)
    sd = O.as_torch_state(module_state("sgp_mixer", "mix", meta["seed"], **meta))
    z = t(act(meta["seed"], name + ":z", (meta["B"], meta["C"], meta["T_hi"])))
    x = t(act(meta["seed"], name + ":x", (meta["B"], meta["C"], meta["T_lo"])))
    assert max_abs(O.sgp_mixer(x, z, sd, "mix", meta["T_hi"]), g["y"]) < TOL


@pytest.mark.parametrize("name", ["pyramid_c32_l25_n2", "pyramid_c64_l100_n3", "pyramid_c48_l250_n2"])
def test_pyramid(name):
    meta, g = load_golden(name)
    sd = O.as_torch_state(module_state("pyramid", "_temp_fine", meta["seed"], **meta))
    x = t(act(meta["seed"], name + ":x", (meta["B"], meta["L"], meta["C"])))
    y = O.ed_sgp_mixer(x, sd, meta["n"], meta["L"])
    assert max_abs(y, g["y"]) < TOL * max(1.0, float(np.abs(g["y"]).max()))


@pytest.mark.parametrize("name", ["gsf_f16", "gsf_f40", "gsf_f92", "gsm_f16"])
def test_gate_shift(name):
    meta, g = load_golden(name)
    sd = O.as_torch_state(module_state("gate_shift", "gs", meta["seed"], F=meta["F"], mode=meta["mode"]))
    x = t(act(meta["seed"], name + ":x", (meta["B"] * meta["T"], meta["F"], meta["h"], meta["w"])))
    assert max_abs(O.gate_shift(x, sd, "gs", meta["T"], meta["mode"]), g["y"]) < 1e-5


def test_loss_and_postproc():
    meta, g = load_golden("loss_postproc")
    B, T, K1, seed = meta["B"], meta["T"], meta["K1"], meta["seed"]
    logits = t(act(seed, "logits", (B, T, K1), 2.0))
    displ = t(act(seed, "displ", (B, T), 1.5))
    lab, labD = synth.labels(seed, B, T, K1 - 1, 2, fg_frac=0.3)
    hard = O.loss_fn(logits, t(lab))
    assert abs(float(hard) - float(g["ce_hard"])) < 1e-6
    soft = O.loss_fn(logits, t(g["soft_labels"]))
    assert abs(float(soft) - float(g["ce_soft"])) < 1e-6
    both = O.loss_fn(logits, t(lab), displ, t(labD))
    assert abs(float(both) - float(g["ce_hard"]) - float(g["mse"])) < 1e-6
    assert max_abs(O.process_prediction(logits, displ), g["process_prediction"]) < 1e-7
    assert max_abs(O.process_prediction(logits, t(g["d_half"])), g["process_prediction_half"]) < 1e-7


@pytest.mark.parametrize("arch", ["rny002", "rny008"])
def test_regnet_trunk_vs_hf(arch):
    """timm is absent (parity unpinned for the trunk); independent cross-check against HuggingFace RegNetY."""
    meta, g = load_golden("hf_regnet_" + arch)
    spec = regnet_spec(arch)
    shapes = {k: v for k, v in state_layout.model_state_shapes(
        dict(feature_arch=arch, clip_len=4, n_layers=1, sgp_ks=3, sgp_r=2, num_classes=1, radi_displacement=0)).items()
        if k.startswith("_features.")}
    shapes["_features.head.fc.weight"] = ((1000, spec.feat_dim), "float32")
    shapes["_features.head.fc.bias"] = ((1000,), "float32")
    n_params = sum(int(np.prod(s)) for k, (s, _) in shapes.items() if state_layout.is_parameter(k))
    assert n_params == meta["n_params_with_fc"] == {"rny002": 3162996, "rny008": 6263168}[arch]
    sd = O.as_torch_state(synth.make_state(shapes, meta["seed"]))
    x = t(act(meta["seed"], f"hf_regnet_{arch}:x", (2, 3, 64, 64)))
    y = O.regnet_features(x, sd, spec, T=1)
    assert max_abs(y, g["pooled"]) < 1e-4 * max(1.0, float(np.abs(g["pooled"]).max()))


FULL = ["tiny_rny002_gsf", "tiny_rny008_gsf", "tiny_rny002_gsm", "tiny_rny002_crop_flip"]


@pytest.mark.parametrize("name", FULL + [pytest.param("finediving_small", marks=pytest.mark.slow)])
def test_full_model(name):
    meta, g = load_golden(name)
    cfg = meta["cfg"]
    shapes = state_layout.model_state_shapes(cfg)
    assert state_layout.layout_digest(shapes) == meta["layout_sha1"]
    assert len(shapes) == meta["n_state"]
    assert sum(int(np.prod(s)) for k, (s, _) in shapes.items() if state_layout.is_parameter(k)) == meta["n_params"]
    sd = synth.make_state(shapes, meta["seed_w"])
    clip = synth.uint8_clip(meta["seed_x"], (meta["B"], cfg["clip_len"], 3, meta["H"], meta["W"]))
    taps = {}
    with torch.no_grad():
        logits, displ, feat = O.forward(t(clip), sd, cfg, regnet_spec(cfg["feature_arch"]),
                                        augment_inference=meta["augment"], taps=taps)
    pooled = feat - O.as_torch_state(sd)["temp_enc"][None]
    assert max_abs(pooled, g["pooled"]) < TOL * max(1.0, float(np.abs(g["pooled"]).max()))
    assert max_abs(logits, g["logits"]) < TOL * max(1.0, float(np.abs(g["logits"]).max()))
    if cfg["radi_displacement"] > 0:
        assert max_abs(displ, g["displ"]) < TOL * max(1.0, float(np.abs(g["displ"]).max()))
        scores = O.process_prediction(logits, displ)
        assert max_abs(scores, g["predict_scores"]) < 1e-4
    for k in g:
        if k.endswith(":frame1"):
            tap = taps[k[4:-7]]
            assert max_abs(tap[1], g[k]) < TOL * max(1.0, float(np.abs(g[k]).max()))
