"""Host-side logic that needs no GPU: synthetic generator determinism, state layout, spec tables,
weight packing, the TDEEDModel API surface."""
import numpy as np
import pytest
import torch

from helpers import model_state, cfg_ns
from tdeed_amd import synth, state_layout
from tdeed_amd.regnet_spec import regnet_spec, gsf_fold_dim, sgp_up_size, pyramid_lengths

CFG = dict(feature_arch="rny002_gsf", clip_len=16, crop_dim=None, n_layers=2, sgp_ks=5, sgp_r=2, num_classes=3,
           radi_displacement=2)


def test_synth_is_deterministic_and_well_scaled():
    a = synth.uint8_clip(7, (2, 3, 5, 11))
    assert np.array_equal(a, synth.uint8_clip(7, (2, 3, 5, 11))) and not np.array_equal(a, synth.uint8_clip(8, (2, 3, 5, 11)))
    # known-answer: pins the hash so that fixtures and the device-side twin cannot drift
    assert synth.hash_u64(0, "clip", 2).tolist() == synth.hash_u64(0, "clip", 3)[:2].tolist()
    n = synth.normalish(1, "x", 200000)
    assert abs(n.mean()) < 0.01 and abs(n.std() - 1.0) < 0.01
    u = synth.uniform01(1, "x", 200000)
    assert 0.0 <= u.min() and u.max() < 1.0 and abs(u.mean() - 0.5) < 0.01


def test_spec_tables_match_survey():
    s2, s8 = regnet_spec("rny002_gsf"), regnet_spec("rny008_gsf")
    assert [b.gsf_fold for b in s2.blocks if b.gsf_fold] == [16, 40, 40, 40, 40] + [92] * 6
    assert [b.gsf_fold for b in s8.blocks if b.gsf_fold] == [32] + [80] * 8 + [192]
    assert [b.se_rd for b in s2.blocks][:3] == [8, 6, 14] and s8.feat_dim == 768
    assert gsf_fold_dim(368) == 92 and sgp_up_size(7, 4) == 33 and sgp_up_size(9, 4) == 41
    assert pyramid_lengths(100, 3) == [100, 50, 25, 13] and pyramid_lengths(250, 2) == [250, 125, 63]
    assert all(b.gsf_fold == 0 for b in regnet_spec("rny002").blocks)


def test_param_counts_match_reference():
    for arch, n, total in [("rny002_gsf", 2, 12267634), ("rny008_gsf", 3, 64018958)]:
        cfg = dict(feature_arch=arch, clip_len=100, crop_dim=224, n_layers=n, sgp_ks=7, sgp_r=4, num_classes=4,
                   radi_displacement=2)
        sh = state_layout.model_state_shapes(cfg)
        assert sum(int(np.prod(s)) for k, (s, _) in sh.items() if state_layout.is_parameter(k)) == total


def test_weight_packing_layouts():
    from tdeed_amd.engine import PackedWeights
    sd = model_state(CFG)
    pw = PackedWeights(CFG, sd, torch.bfloat16, "cpu")
    b = pw.W.blocks[2]                       # s3.b1: gate-shift in front of conv1
    assert b.spec.gsf_fold == 16 and b.gs_wq.shape == (27, 16) and b.w1.w.dtype == torch.bfloat16 and not b.w1.ws
    w2 = sd["_features.s3.b1.conv2.conv.weight"]           # [C][gw][3][3]
    g, o, i, ky, kx = 3, 5, 2, 1, 2
    assert float(b.w2[g, ky * 3 + kx, i, o]) == float(w2[g * 8 + o, i, ky, kx])
    s = sd["_features.stem.bn.weight"] / np.sqrt(sd["_features.stem.bn.running_var"] + 1e-5)
    assert np.allclose(pw.W.stem_scale.numpy(), s, rtol=1e-6)
    assert pw.n_out == 5 and pw.displ_col == 4
    o0 = pw.W.sgp[0]
    assert o0.dw.shape == (368, 2 * 5 + 13 + 2) and o0.db.shape == (5, 368)
    assert np.array_equal(o0.dw[:, 5:10].numpy(), sd["_temp_fine._sgp.0.convw.weight"].reshape(368, 5))


def test_model_api_surface_on_cpu():
    """Construction, state_dict grammar, double head, loud failure without a GPU."""
    from tdeed_amd.model import TDEEDModel, update_labels_2heads
    m = TDEEDModel(device="cpu", args=cfg_ns(CFG))
    sd = m.state_dict()
    assert list(sd) == list(state_layout.model_state_shapes(CFG))
    assert m._num_classes == 4 and not m._model._double_head
    m.load({k: v.clone() for k, v in sd.items()})
    with pytest.raises(RuntimeError):
        m.load({"bogus": torch.zeros(1)})
    m._model.update_pred_head([4, 18])
    assert m._model._double_head and "_pred_fine._fc2._fc_out.weight" in m.state_dict()
    assert list(m.state_dict())[-2:] == ["_pred_displ._fc_out.weight", "_pred_displ._fc_out.bias"]
    assert len(m._get_params()) == sum(state_layout.is_parameter(k) for k in m.state_dict())
    lab = torch.zeros(2, 4, dtype=torch.int64)
    assert update_labels_2heads(lab, [1, 2], 3)[1, 0].item() == 4
    with pytest.raises(RuntimeError, match="GPU"):
        m._model(torch.zeros(1, 16, 3, 32, 32, dtype=torch.uint8), inference=True)


def test_warmup_cosine_lr_matches_torch_chained_scheduler():
    """train_tdeed.py:79-87: LinearLR(0.01->1, warmup) chained with CosineAnnealingLR(T_max)."""
    from torch.optim.lr_scheduler import ChainedScheduler, LinearLR, CosineAnnealingLR
    from tdeed_amd.optim import warmup_cosine_lr
    p = torch.nn.Parameter(torch.zeros(1))
    opt = torch.optim.AdamW([p], lr=8e-4)
    warm, cos = 30, 470
    sch = ChainedScheduler([LinearLR(opt, start_factor=0.01, end_factor=1.0, total_iters=warm), CosineAnnealingLR(opt, cos)])
    for step in range(0, 200):
        assert abs(opt.param_groups[0]["lr"] - 8e-4 * warmup_cosine_lr(step, warm, cos)) < 1e-9, step
        opt.step()
        sch.step()


def test_flat_params_alias_state_dict():
    from tdeed_amd.optim import FlatParams
    sd = {k: torch.from_numpy(v) for k, v in model_state(CFG).items()}
    before = {k: v.clone() for k, v in sd.items()}
    fp = FlatParams(sd, device="cpu")
    n_param = sum(v.numel() for k, v in before.items() if state_layout.is_parameter(k))
    assert n_param <= fp.numel < n_param + 4 * len(before)
    for k, v in before.items():
        assert torch.equal(sd[k], v)
    fp.flat.add_(1.0)                                   # an optimizer step on the flat buffer is seen by the state
    k0 = "_features.stem.conv.weight"
    assert torch.allclose(sd[k0], before[k0] + 1.0)
    assert sd["_features.stem.bn.running_mean"].data_ptr() != fp.flat.data_ptr()      # buffers stay outside
    assert all(o % 4 == 0 for o, _ in fp.index.values())


# ------------------------------------------------------------------ construction-time state (VERDICT r2 missing 1)
def _golden(name):
    import json
    import os
    from helpers import ROOT
    g = np.load(os.path.join(ROOT, "tests", "golden", name + ".npz"))
    return json.loads(str(g["meta"])), g


@pytest.mark.parametrize("tag", ["gsf", "gsm"])
def test_default_init_follows_the_reference_constructors(tag):
    """init.reference_init against statistics recorded from freshly constructed REFERENCE models (tools/make_goldens.py:
    init_stats; model.py:65, modules.py:146-157, 255-275, gsf.py:17-24, gsm.py:75-76): constants exactly, random tensors by
    mean / std / range within sampling tolerance.  The trunk convolutions follow timm's own init (unpinned: timm absent)."""
    from tdeed_amd import init as ref_init
    meta, g = _golden("init_stats")
    case = meta["cases"][tag]
    cfg = case["cfg"]
    shapes = state_layout.model_state_shapes(cfg)
    draws = [ref_init.reference_init(shapes, cfg, torch.Generator().manual_seed(100 + i)) for i in range(meta["n_models"])]
    stats = g[f"{tag}_stats"]
    assert set(case["keys"]) <= set(shapes)
    for k, (mean, std, lo, hi, n) in zip(case["keys"], stats):
        a = np.concatenate([d[k].double().reshape(-1).numpy() for d in draws])
        assert a.size == int(n), k
        if std == 0.0:                                   # constants: BatchNorm / LayerNorm identity, zero biases, GSM's conv3D
            assert np.all(a == mean), (k, mean, a[:4])
            continue
        span = max(abs(lo), abs(hi))
        if n < 30:                                       # 3..24 samples (conv biases of the gate-shift): order of magnitude only
            assert 0.0 < np.abs(a).max() < 5 * span, (k, np.abs(a).max(), span)
            continue
        se = std / np.sqrt(n)
        assert abs(a.mean() - mean) < 6 * np.sqrt(2) * se + 1e-12, (k, a.mean(), mean)
        if n >= 200:
            assert abs(a.std() / std - 1.0) < 0.15, (k, a.std(), std)
        # uniform draws are bounded by the same 1/sqrt(fan_in); normal ones only statistically
        assert np.abs(a).max() < 1.6 * span + 1e-12, (k, np.abs(a).max(), span)
    # timm's RegNet init for the trunk: zero_init_last, conv N(0, sqrt(2 / fan_out)), SE biases zero
    d0 = draws[0]
    assert float(d0["_features.s3.b1.conv3.bn.weight"].abs().max()) == 0.0
    assert float(d0["_features.s3.b1.conv1.net.bn.weight"].min()) == 1.0
    w = d0["_features.s4.b2.conv2.conv.weight"]          # grouped 3x3: fan_out = 9 * 368 // 46
    assert abs(float(w.std()) / np.sqrt(2.0 / (9 * 368 // 46)) - 1.0) < 0.05
    assert float(d0["_features.s2.b1.se.fc1.bias"].abs().max()) == 0.0


def test_model_constructor_uses_the_reference_init_not_the_fixture_generator():
    from tdeed_amd.model import TDEEDModel
    torch.manual_seed(3)
    m = TDEEDModel(device="cpu", args=cfg_ns(CFG))
    sd = m.state_dict()
    assert float(sd["_features.stem.bn.running_var"].min()) == 1.0 and float(sd["_features.stem.bn.running_mean"].abs().max()) == 0.0
    assert float(sd["_temp_fine._sgp.0.psi.bias"].abs().max()) == 0.0
    assert abs(float(sd["temp_enc"].std()) * CFG["clip_len"] - 1.0) < 0.1
    torch.manual_seed(3)
    m2 = TDEEDModel(device="cpu", args=cfg_ns(CFG))          # the global generator decides, like torch.nn constructors
    assert all(torch.equal(v, m2.state_dict()[k]) for k, v in sd.items())


@pytest.mark.parametrize("arch", ["rny002_gsf", "rny008_gsf", "rny002_gsm", "rny002"])
def test_timm_backbone_key_map_equals_the_references_wrapping(arch):
    """init.timm_key_map against the (reference key <- timm key) pairs recorded from the reference itself by tensor
    identity through make_temporal_shift (tools/make_goldens.py:timm_keymap; model/shift.py:46-59)."""
    from tdeed_amd import init as ref_init
    meta, _ = _golden("timm_keymap")
    rec = meta["archs"][arch]
    cfg = dict(CFG, feature_arch=arch, clip_len=8)
    assert ref_init.timm_key_map(cfg) == {a: b for a, b in rec["pairs"]}
    assert rec["dropped"] == ["head.fc.bias", "head.fc.weight"]


def test_load_timm_backbone_round_trip():
    from tdeed_amd import init as ref_init
    from tdeed_amd.model import TDEEDModel
    m = TDEEDModel(device="cpu", args=cfg_ns(CFG))
    before = {k: v.clone() for k, v in m.state_dict().items()}
    # a timm regnety_002 state dict = the plain-trunk layout without the `_features.` prefix, plus the classifier
    plain = state_layout.model_state_shapes(dict(CFG, feature_arch="rny002"))
    timm_sd = {k[len("_features."):]: torch.from_numpy(v) for k, v in synth.make_state(
        {k: v for k, v in plain.items() if k.startswith("_features.")}, 5).items()}
    timm_sd["head.fc.weight"], timm_sd["head.fc.bias"] = torch.zeros(1000, 368), torch.zeros(1000)
    filled = m._model.load_timm_backbone({"module." + k: v for k, v in timm_sd.items()})      # DataParallel prefix tolerated
    after = m.state_dict()
    trunk = [k for k in after if k.startswith("_features.") and ".gs." not in k]
    assert sorted(filled) == sorted(trunk)
    assert torch.equal(after["_features.s3.b2.conv1.net.conv.weight"], timm_sd["s3.b2.conv1.conv.weight"])
    assert torch.equal(after["_features.s1.b1.conv1.conv.weight"], timm_sd["s1.b1.conv1.conv.weight"])
    assert torch.equal(after["_features.s4.b7.se.fc2.bias"], timm_sd["s4.b7.se.fc2.bias"])
    for k in after:                                              # everything else keeps its construction-time state
        if k not in trunk:
            assert torch.equal(after[k], before[k]), k
    with pytest.raises(ValueError, match="shape"):
        bad = dict(timm_sd)
        bad["s4.b1.conv3.conv.weight"] = torch.zeros(768, 768, 1, 1)
        m._model.load_timm_backbone(bad)
    with pytest.raises(KeyError):
        m._model.load_timm_backbone({k: v for k, v in timm_sd.items() if k != "stem.conv.weight"})


def test_debug_flavour_compiles_with_device_asserts():
    """`python t-deed_amd/build.py --debug` (SURVEY §5 'race detection / sanitizers': an asserting build): one source is enough
    here -- the object must carry the TD_DEV_ASSERT message text, the release object must not."""
    import importlib.util
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("_tdeed_build_dbg", os.path.join(root, "t-deed_amd", "build.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    (obj,) = mod.build(flavour="debug", only=["bneck.hip"], verbose=False)
    assert os.path.exists(obj) and obj.endswith(".dbg.o")
    assert b"tdeed device assert" in open(obj, "rb").read()
    rel = os.path.join(os.path.dirname(obj), "bneck.o")
    if os.path.exists(rel):
        assert b"tdeed device assert" not in open(rel, "rb").read()
