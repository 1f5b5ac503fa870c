"""Whole-forward parity on the MI355X: HIP engine vs (a) logits produced by the reference itself
(committed golden vectors) and (b) the CPU oracle on the same seeded inputs.  -m gpu only."""
import numpy as np
import pytest
import torch

from helpers import load_golden, model_state, t, max_abs, cfg_ns
from tdeed_amd import synth, state_layout

pytestmark = pytest.mark.gpu
DEV = "cuda"
LOGIT_TOL_F32 = 1e-3      # BASELINE.json north_star: per-frame logits within 1e-3 in fp32


def _engine(cfg, sd, dtype, use_graph=True, n_split=1):
    from tdeed_amd.engine import ForwardEngine
    return ForwardEngine(cfg, sd, dtype, DEV, use_graph=use_graph, n_split=n_split)


def _run(eng, clip, flip=False, taps=()):
    st = torch.cuda.Stream()
    with torch.cuda.stream(st):
        head, plan = eng.forward(t(clip).to(DEV), augment_inference=flip, taps=taps)
        st.synchronize()
    return head.float().cpu(), plan


FULL = ["tiny_rny002_gsf", "tiny_rny008_gsf", "tiny_rny002_gsm", "tiny_rny002_crop_flip", "finediving_small",
        "finediving_big", "snb_t250"]


@pytest.mark.parametrize("name", FULL)
def test_forward_fp32_matches_reference_golden(name):
    meta, g = load_golden(name)
    cfg = meta["cfg"]
    sd = model_state(cfg, meta["seed_w"])
    clip = synth.uint8_clip(meta["seed_x"], (meta["B"], cfg["clip_len"], 3, meta["H"], meta["W"]))
    eng = _engine(cfg, sd, torch.float32)
    head, plan = _run(eng, clip, flip=meta["augment"])
    B, T = meta["B"], cfg["clip_len"]
    head = head.view(B, T, -1)
    K1 = cfg["num_classes"] + 1
    pooled = plan.keep["feat"].float().cpu() - t(sd["temp_enc"])[None]
    assert max_abs(pooled, g["pooled"]) < 1e-3 * max(1.0, float(np.abs(g["pooled"]).max()))
    assert max_abs(plan.keep["sgp_out"].float().cpu(), g["sgp_out"]) < 1e-3 * max(1.0, float(np.abs(g["sgp_out"]).max()))
    assert max_abs(head[..., :K1], g["logits"]) < LOGIT_TOL_F32
    if cfg["radi_displacement"] > 0:
        assert max_abs(head[..., K1], g["displ"]) < LOGIT_TOL_F32


@pytest.mark.parametrize("name", ["tiny_rny002_gsf", "tiny_rny008_gsf"])
def test_forward_fp32_matches_oracle_with_taps(name):
    from oracle import tdeed_oracle as O
    from tdeed_amd.regnet_spec import regnet_spec
    meta, g = load_golden(name)
    cfg = meta["cfg"]
    sd = model_state(cfg, meta["seed_w"] + 1)          # different weights than the golden run
    clip = synth.uint8_clip(77, (meta["B"], cfg["clip_len"], 3, meta["H"], meta["W"]))
    spec = regnet_spec(cfg["feature_arch"])
    taps = {}
    with torch.no_grad():
        logits, displ, feat = O.forward(t(clip), sd, cfg, spec, taps=taps)
    names = ["_features.stem"] + ["_features." + b.name for b in spec.blocks]
    eng = _engine(cfg, sd, torch.float32, use_graph=False)
    head, plan = _run(eng, clip, taps=tuple(names))
    for n in names:
        ref = taps[n].permute(0, 2, 3, 1)
        got = plan.keep[n].float().cpu()
        assert max_abs(got, ref) < 1e-4 * max(1.0, float(ref.abs().max())), n
    assert max_abs(plan.keep["feat"].float().cpu(), feat) < 1e-4 * max(1.0, float(feat.abs().max()))
    K1 = cfg["num_classes"] + 1
    head = head.view(meta["B"], cfg["clip_len"], -1)
    assert max_abs(head[..., :K1], logits) < LOGIT_TOL_F32
    assert max_abs(head[..., K1], displ) < LOGIT_TOL_F32


@pytest.mark.parametrize("name", ["tiny_rny002_gsf", "finediving_small", "finediving_big", "snb_t250"])
def test_forward_bf16_close_to_reference(name):
    """bf16 is the throughput mode: not held to 1e-3 (the reference's own bf16 autocast is 4.5e-2 off,
    SURVEY.md section 0) but must track the fp32 logits."""
    meta, g = load_golden(name)
    cfg = meta["cfg"]
    sd = model_state(cfg, meta["seed_w"])
    clip = synth.uint8_clip(meta["seed_x"], (meta["B"], cfg["clip_len"], 3, meta["H"], meta["W"]))
    head, _ = _run(_engine(cfg, sd, torch.bfloat16), clip)
    K1 = cfg["num_classes"] + 1
    head = head.view(meta["B"], cfg["clip_len"], -1)
    ref = t(g["logits"])
    err = (head[..., :K1] - ref).abs().max().item()
    scale = ref.abs().max().item()
    # measured 4.4e-2 .. 4.7e-2 on the FineDiving_small golden clip (logit range +-4.5) since the temporal stage keeps its
    # residual stream in fp32 (round 5; 7.6e-2 before: tools/diag_bf16_split.py attributed 4.6e-2 to the bf16 trunk alone);
    # the reference's own CPU bf16 autocast is 4.5e-2 off
    print(f"bf16 vs reference logits [{name}]: max abs err {err:.4f} on a range of +-{scale:.2f}")
    assert err < 0.06, (err, scale)
    assert (head[..., :K1].argmax(-1) == ref.argmax(-1)).float().mean().item() > 0.9


@pytest.mark.parametrize("name,reps", [("finediving_big", 16), ("snb_t250", 4), ("finediving_small", 8)])
def test_forward_bf16_at_the_timed_batch_size_tracks_reference_golden(name, reps):
    """The kernels a forward takes depend on the batch (register-stationary contractions from 60 000 rows up, the SGP
    tile forms, two-frames-per-workgroup gate-shift): the reference's one-clip golden replicated to the batch size bench.py
    times (cfg2: 8, 800MF: 16, T=250: 4), every copy held to the reference's logits."""
    meta, g = load_golden(name)
    cfg = meta["cfg"]
    sd = model_state(cfg, meta["seed_w"])
    clip = synth.uint8_clip(meta["seed_x"], (meta["B"], cfg["clip_len"], 3, meta["H"], meta["W"]))
    clip = np.concatenate([clip] * reps, axis=0)
    head, plan = _run(_engine(cfg, sd, torch.bfloat16), clip)
    K1 = cfg["num_classes"] + 1
    head = head.view(reps, cfg["clip_len"], -1)
    ref = t(g["logits"])
    errs = [(head[i:i + 1, :, :K1] - ref).abs().max().item() for i in range(reps)]
    print(f"bf16 at B={reps} vs reference logits [{name}]: max abs err per copy {min(errs):.4f} .. {max(errs):.4f}; "
          f"kernels: {sorted(set(s_.kernel for s_ in plan.steps))}")
    assert max(errs) < 0.06, errs
    assert (head[..., :K1].argmax(-1) == ref.argmax(-1)).float().mean().item() > 0.9


def test_wide_frames_fall_back_to_unfused_front():
    """A frame too wide for the fused front kernel's LDS band takes the unfused stem path: bf16 tracks fp32."""
    from tdeed_amd import ops
    cfg = dict(feature_arch="rny002_gsf", clip_len=4, crop_dim=None, n_layers=1, sgp_ks=5, sgp_r=2, num_classes=3,
               radi_displacement=2)
    sd = model_state(cfg, 0)
    H, W = 32, 512
    assert ops.s1_front_parts(H, W, 24) == 0 and ops.s1_front_parts(224, 224, 24) > 0
    clip = synth.uint8_clip(9, (2, 4, 3, H, W))
    h32, _ = _run(_engine(cfg, sd, torch.float32), clip)
    h16, plan = _run(_engine(cfg, sd, torch.bfloat16), clip)
    assert plan.steps[0].kernel == "stem"
    assert (h16 - h32).abs().max().item() < 0.08 * max(1.0, h32.abs().max().item())


def test_graph_replay_equals_eager_and_is_deterministic():
    meta, g = load_golden("tiny_rny002_gsf")
    cfg = meta["cfg"]
    sd = model_state(cfg, 0)
    clip = synth.uint8_clip(5, (2, cfg["clip_len"], 3, 64, 64))
    h_eager, _ = _run(_engine(cfg, sd, torch.bfloat16, use_graph=False), clip)
    eng = _engine(cfg, sd, torch.bfloat16, use_graph=True)
    h1, _ = _run(eng, clip)
    h2, _ = _run(eng, clip)
    clip2 = synth.uint8_clip(6, (2, cfg["clip_len"], 3, 64, 64))
    h3, _ = _run(eng, clip2)
    assert torch.equal(h1, h_eager) and torch.equal(h1, h2)
    assert not torch.equal(h1, h3)
    # the default engine cuts the batch into two sub-batches on forked streams inside one graph: same bits
    eng2 = _engine(cfg, sd, torch.bfloat16, use_graph=True, n_split=2)
    s1, plan2 = _run(eng2, clip)
    s2, _ = _run(eng2, clip)
    assert len(plan2.subs) == 2 and torch.equal(s1, h1) and torch.equal(s2, h1)
    e2, _ = _run(_engine(cfg, sd, torch.bfloat16, use_graph=False, n_split=2), clip)
    assert torch.equal(e2, h1)


def test_sub_batch_fork_survives_torchs_wrapping_stream_pool():
    """torch.cuda.Stream() hands out the next of 32 pooled streams: in a long-lived process the stream a sub-batch is forked
    onto can be the very stream that launches (and captures) the forward.  That sub-batch then runs inline; new_stream()
    never returns the current stream.  (The captured graph of such a plan used to crash the host in hipGraphLaunch.)"""
    from tdeed_amd.streams import new_stream
    meta, g = load_golden("tiny_rny002_gsf")
    cfg = meta["cfg"]
    sd = model_state(cfg, 0)
    clip = t(synth.uint8_clip(5, (2, cfg["clip_len"], 3, 64, 64))).to(DEV)
    ref, _ = _run(_engine(cfg, sd, torch.bfloat16, use_graph=True, n_split=2), clip.cpu().numpy())
    st = torch.cuda.Stream()
    with torch.cuda.stream(st):
        seen = {new_stream().cuda_stream for _ in range(80)}           # more than the pool holds
        assert st.cuda_stream not in seen and len(seen) > 8
        a = new_stream()
        assert new_stream(avoid=[a]).cuda_stream not in (a.cuda_stream, st.cuda_stream)
        eng = _engine(cfg, sd, torch.bfloat16, use_graph=True, n_split=2)
        plan = eng.plan(2, 64, 64)
        assert len(plan.subs) == 2 and plan.streams[1].cuda_stream != st.cuda_stream
        plan.streams[1] = st                                              # what a wrapped pool used to produce
        eng.set_frames(plan, clip)
        eng.run_plan(plan)                                                # warm-up + capture + launch
        eng.run_plan(plan)                                                # replay
        st.synchronize()
        assert torch.equal(plan.head_out.float().cpu(), ref)


def test_model_api_predict_and_epoch():
    """TDEEDModel drop-in surface: predict() against the reference's own predict() output, epoch() val loss
    against the oracle loss, state_dict round trip."""
    from tdeed_amd.model import TDEEDModel
    from oracle import tdeed_oracle as O
    meta, g = load_golden("tiny_rny002_gsf")
    cfg = meta["cfg"]
    m = TDEEDModel(device=DEV, args=cfg_ns(cfg))
    sd = model_state(cfg, meta["seed_w"])
    m.load({k: t(v) for k, v in sd.items()})
    got = m.state_dict()
    assert list(got.keys()) == list(state_layout.model_state_shapes(cfg).keys())
    clip = synth.uint8_clip(meta["seed_x"], (meta["B"], cfg["clip_len"], 3, meta["H"], meta["W"]))
    cls, scores = m.predict(t(clip).float(), use_amp=False)       # callers hand over .float() frames
    assert max_abs(scores, g["predict_scores"]) < 1e-3
    assert (cls == g["predict_cls"]).mean() > 0.99
    cls_b, scores_b = m.predict(t(clip), use_amp=True)
    # bf16: a displacement that crosses a .5 rounding boundary moves a frame's score to the neighbouring
    # frame (discontinuous post-proc), so compare in the mean and on the bulk of the entries
    d = np.abs(scores_b - g["predict_scores"])
    assert d.mean() < 0.01 and (d < 0.05).mean() > 0.95
    # validation epoch: loss of the HIP path == oracle loss on the golden logits
    lab, labD = synth.labels(3, meta["B"], cfg["clip_len"], cfg["num_classes"], cfg["radi_displacement"])
    loader = [dict(frame=t(clip), label=t(lab), labelD=t(labD))]
    loss = m.epoch(loader)
    ref = float(O.loss_fn(t(g["logits"]), t(lab), t(g["displ"]), t(labD)))
    assert abs(loss - ref) < 0.05 * max(1.0, abs(ref))      # bf16 forward
    # several batches: consecutive batches alternate between two buffer sets on two streams (two in flight)
    batches = []
    for i in range(3):
        c = synth.uint8_clip(meta["seed_x"] + 10 + i, clip.shape)
        l2, d2 = synth.labels(20 + i, meta["B"], cfg["clip_len"], cfg["num_classes"], cfg["radi_displacement"])
        batches.append(dict(frame=t(c), label=t(l2), labelD=t(d2)))
    singles = [m.epoch([b]) for b in batches]
    assert abs(m.epoch(batches) - float(np.mean(singles))) < 1e-5 * max(1.0, abs(float(np.mean(singles))))


def test_model_api_training_epochs_reduce_the_loss():
    """The reference's training loop against the drop-in API: get_optimizer -> torch LR schedulers (LinearLR + cosine,
    train_tdeed.py:79-87) -> epoch(loader, optimizer, scaler, lr_scheduler, acc_grad_iter).  Over-fitting one small
    batch must drive the loss down, BN running statistics must move, validation must see the updated weights."""
    from tdeed_amd.model import TDEEDModel
    meta, g = load_golden("tiny_rny002_gsf")
    cfg = meta["cfg"]
    m = TDEEDModel(device=DEV, args=cfg_ns(cfg))
    m.load({k: t(v) for k, v in model_state(cfg, meta["seed_w"]).items()})
    clip = synth.uint8_clip(meta["seed_x"], (meta["B"], cfg["clip_len"], 3, meta["H"], meta["W"]))
    lab, labD = synth.labels(3, meta["B"], cfg["clip_len"], cfg["num_classes"], cfg["radi_displacement"], fg_frac=0.3)
    loader = [dict(frame=t(clip), label=t(lab), labelD=t(labD))] * 2
    val0 = m.epoch(loader[:1])
    rm0 = m.state_dict()["_features.s2.b1.conv2.bn.running_mean"].clone()
    from tdeed_amd import augment
    m._model.augment_fn = augment.crop_only          # over-fitting ONE batch: keep it the same batch every step
    optimizer, scaler = m.get_optimizer({"lr": 3e-4})
    assert scaler is None and isinstance(optimizer, torch.optim.Optimizer)
    steps = 40
    sched = torch.optim.lr_scheduler.ChainedScheduler([
        torch.optim.lr_scheduler.LinearLR(optimizer, start_factor=0.01, end_factor=1.0, total_iters=4),
        torch.optim.lr_scheduler.CosineAnnealingLR(optimizer, steps)])
    losses = [m.epoch(loader, optimizer=optimizer, scaler=scaler, lr_scheduler=sched, acc_grad_iter=1) for _ in range(steps // 2)]
    assert all(np.isfinite(losses)) and np.mean(losses[-3:]) < 0.5 * losses[0], losses
    assert optimizer.param_groups[0]["lr"] < 3e-4                     # the schedulers drove the fused optimizer's lr
    sd = m.state_dict()
    assert not torch.equal(sd["_features.s2.b1.conv2.bn.running_mean"], rm0)
    assert int(sd["_features.stem.bn.num_batches_tracked"]) == steps
    val1 = m.epoch(loader[:1])                                         # eval-mode forward with the trained weights
    assert np.isfinite(val1) and val1 != val0
    # checkpoint round trip while the optimizer is alive: load() writes into the flat buffer the optimizer owns
    ck = {k: v.detach().cpu().clone() for k, v in m.state_dict().items()}
    m.epoch(loader, optimizer=optimizer)                               # move the weights away from the checkpoint
    m.load(ck)
    assert abs(m.epoch(loader[:1]) - val1) < 1e-6 * max(1.0, abs(val1))
    w_name = "_features.s2.b1.conv1.conv.weight"
    assert m.state_dict()[w_name].data_ptr() == optimizer.engine.state[w_name].data_ptr()
    assert np.isfinite(m.epoch(loader, optimizer=optimizer))
    # mixup batches (frame2 / label2 / labelD2 -> fp32 frames, soft labels)
    clip2 = synth.uint8_clip(meta["seed_x"] + 1, clip.shape)
    lab2, labD2 = synth.labels(4, meta["B"], cfg["clip_len"], cfg["num_classes"], cfg["radi_displacement"], fg_frac=0.3)
    mix_loader = [dict(frame=t(clip), label=t(lab), labelD=t(labD), frame2=t(clip2), label2=t(lab2), labelD2=t(labD2))]
    assert np.isfinite(m.epoch(mix_loader, optimizer=optimizer))
    # gradient accumulation: two half-weighted micro-batches == one batch (same data) up to rounding
    m2 = TDEEDModel(device=DEV, args=cfg_ns(cfg))
    m2.load({k: t(v) for k, v in model_state(cfg, meta["seed_w"]).items()})
    opt2, _ = m2.get_optimizer({"lr": 1e-3})
    assert np.isfinite(m2.epoch(loader, optimizer=opt2, acc_grad_iter=2))
    assert int(m2.state_dict()["_features.stem.bn.num_batches_tracked"]) == 2 and opt2.engine.opt.t == 1


def test_fused_adamw_on_flat_model_params_matches_torch():
    """FlatParams + FusedAdamW (one launch for all 431 tensors) == torch.optim.AdamW on the same tensors."""
    from tdeed_amd.optim import FlatParams, FusedAdamW, warmup_cosine_lr
    cfg = load_golden("tiny_rny002_gsf")[0]["cfg"]
    sd = {k: t(v).to(DEV) for k, v in model_state(cfg, 0).items()}
    ref = {k: v.clone().requires_grad_(True) for k, v in sd.items() if state_layout.is_parameter(k)}
    opt_ref = torch.optim.AdamW(list(ref.values()), lr=8e-4)
    fp = FlatParams(sd)
    opt = FusedAdamW(fp, lr=8e-4)
    g = torch.Generator(device="cpu").manual_seed(0)
    for step in range(3):
        for k, r in ref.items():
            gr = torch.randn(r.shape, generator=g).to(DEV)
            r.grad = gr.clone()
            fp.grad_view(k).copy_(gr.reshape(-1))
        f = warmup_cosine_lr(step, 2, 10)
        for grp in opt_ref.param_groups:
            grp["lr"] = 8e-4 * f
        opt_ref.step()
        opt.step(lr_factor=f)
    for k, r in ref.items():
        assert max_abs(sd[k], r.detach()) < 5e-6, k


def test_model_api_double_head_training_and_validation():
    """Joint-dataset mode (train_tdeed.py:147-148 -> update_pred_head; model.py:219-221, 278-306): training epochs and
    the validation loss run on the per-clip head selection."""
    from tdeed_amd.model import TDEEDModel
    meta, g = load_golden("tiny_rny002_gsf")
    cfg = meta["cfg"]
    m = TDEEDModel(device=DEV, args=cfg_ns(cfg))
    m.load({k: t(v) for k, v in model_state(cfg, meta["seed_w"]).items()})
    k1a, k1b = cfg["num_classes"] + 1, 6
    torch.manual_seed(0)                               # the new heads draw nn.Linear's default init from the global generator
    m._model.update_pred_head([k1a, k1b])
    B, T = meta["B"], cfg["clip_len"]
    clip = synth.uint8_clip(meta["seed_x"], (B, T, 3, meta["H"], meta["W"]))
    ds = [1, 2][:B] if B >= 2 else [2]
    labs = [synth.labels(30 + i, 1, T, (k1a if ds[i] == 1 else k1b) - 1, cfg["radi_displacement"], fg_frac=0.3) for i in range(B)]
    lab = np.concatenate([l[0] for l in labs], 0)
    labD = np.concatenate([l[1] for l in labs], 0)
    loader = [dict(frame=t(clip), label=t(lab), labelD=t(labD), dataset=torch.tensor(ds))]
    v0 = m.epoch(loader)
    from tdeed_amd import augment
    m._model.augment_fn = augment.crop_only
    torch.manual_seed(0)                               # dropout masks: Adam without warm-up makes the first steps noisy
    optimizer, _ = m.get_optimizer({"lr": 3e-4})
    losses = [m.epoch(loader, optimizer=optimizer) for _ in range(24)]
    # Adam without warm-up on one small batch is noisy step to step: the loss must come down, not monotonically
    assert np.isfinite(v0) and all(np.isfinite(losses)) and min(losses[-8:]) < 0.75 * losses[0], (v0, losses)
    assert np.isfinite(m.epoch(loader))
