"""Training pinned against the reference ITSELF: tests/golden/train_step_*.npz and
train_epoch_*.npz are written by tools/make_goldens.py from the imported /root/reference -- TDEEDModel in .train(),
the loss of epoch(), modules.step(), AdamW from get_optimizer(), the chained LinearLR + CosineAnnealingLR, and
the reference's own epoch() on a plain + a mixup batch.

  * CPU: the oracle's train-mode branch (batch-statistics BatchNorm, running-statistic updates, dropout masks,
    soft-label CE) against those fixtures -- this is what pins the checker the GPU gradient tests rely on;
  * GPU (-m gpu): the HIP fp32 engine through the TDEEDModel surface against the same fixtures.
"""
import random

import numpy as np
import pytest
import torch

from helpers import (load_golden, model_state, t, drop_masks, oracle_train_loss, sample_flat, chained_scheduler, cfg_ns)
from tdeed_amd import state_layout, synth
from tdeed_amd.regnet_spec import regnet_spec
from oracle import tdeed_oracle as O

STEP = "train_step_tiny_rny002"
EPOCH = "train_epoch_tiny_rny002"


def _inputs(meta, seed_x, with_second=False):
    cfg, B, H, W = meta["cfg"], meta["B"], meta["H"], meta["W"]
    T = cfg["clip_len"]
    fr = t(synth.uint8_clip(seed_x, (B, T, 3, H, W)))
    lab, labD = synth.labels(seed_x + 1, B, T, cfg["num_classes"], cfg["radi_displacement"], fg_frac=meta["fg_frac"])
    out = dict(frame=fr, label=t(lab).long(), labelD=t(labD))
    if with_second:
        lab2, labD2 = synth.labels(seed_x + 6, B, T, cfg["num_classes"], cfg["radi_displacement"], fg_frac=meta["fg_frac"])
        out.update(frame2=t(synth.uint8_clip(seed_x + 5, (B, T, 3, H, W))), label2=t(lab2).long(), labelD2=t(labD2))
    return out


def _oracle_state(cfg, seed_w):
    sd0 = {k: t(v) for k, v in model_state(cfg, seed_w).items()}
    par = [k for k in sd0 if state_layout.is_parameter(k)]
    sd = {k: (v.clone().requires_grad_(True) if k in par else v.clone()) for k, v in sd0.items()}
    return sd, par


def _rel(a, b):
    a, b = np.asarray(a, np.float64).reshape(-1), np.asarray(b, np.float64).reshape(-1)
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30))


def _check_grads(grad_of, g, meta, tol_sel, tol_norm, floor=1e-6):
    """selected tensors element-wise (sampled), every tensor through its norm and a fixed random projection; floor: the
    part of the WHOLE gradient's norm allowed on top (single-scalar biases are sums with heavy cancellation)"""
    names = meta["param_names"]
    gn = float(np.linalg.norm(g["grad_norm"]))
    for k in meta["keys"]:
        got = sample_flat(grad_of(k).detach().cpu().float().numpy(), meta["sample_cap"])
        want = g["grad:" + k]
        assert got.shape == want.shape, k
        d = float(np.linalg.norm(got.astype(np.float64) - want))
        assert d <= tol_sel * float(np.linalg.norm(want)) + floor * gn, (k, d, float(np.linalg.norm(want)))
    for i, k in enumerate(names):
        gk = grad_of(k).detach().cpu().double().numpy().reshape(-1)
        assert abs(np.linalg.norm(gk) - g["grad_norm"][i]) <= tol_norm * g["grad_norm"][i] + floor * gn, k
        proj = synth.normalish(meta["proj_seed"], "proj:" + k, gk.size).astype(np.float64)
        # the projection of an n-vector has magnitude ~ |g| (unit-variance weights): compare on that scale
        assert abs(float(gk @ proj) - g["grad_proj"][i]) <= tol_norm * g["grad_norm"][i] + floor * gn, k


def _check_bn(state, g, meta, tol):
    got = np.concatenate([state[k].detach().cpu().float().numpy().reshape(-1) for k in meta["bn_keys"]])
    assert _rel(got, g["bn_running"]) < tol
    trk = np.array([int(state[k]) for k in state if k.endswith("num_batches_tracked")])
    assert np.array_equal(trk, g["bn_tracked"])


# ----------------------------------------------------------------------------------------------- CPU: oracle vs reference
def test_oracle_train_mode_steps_match_the_reference():
    meta, g = load_golden(STEP)
    cfg = meta["cfg"]
    spec = regnet_spec(cfg["feature_arch"])
    sd, par = _oracle_state(cfg, meta["seed_w"])
    inp = _inputs(meta, meta["seed_x"])
    masks = drop_masks(meta["mask_seed"], meta["B"], cfg["clip_len"], spec.feat_dim, 2)
    opt = torch.optim.AdamW([sd[k] for k in par], lr=meta["lr"])          # modules.py:37-39 with opt_args = {'lr': lr}
    sched = chained_scheduler(opt, meta["warm_steps"], meta["cos_steps"])
    lrs = [opt.param_groups[0]["lr"]]
    for s_ in range(meta["n_steps"]):
        sd[O.BN_UPDATES] = {}
        loss, cls, displ = oracle_train_loss(inp["frame"], sd, cfg, spec, inp["label"], inp["labelD"].float(), masks,
                                             None, False)
        assert abs(float(loss) - g["losses"][s_]) < (2e-5 if s_ == 0 else 2e-3) * abs(g["losses"][s_]), s_
        loss.backward()
        if s_ == 0:
            assert float(np.abs(cls.detach().numpy() - g["logits0"]).max()) < 1e-4
            assert float(np.abs(displ.detach().numpy() - g["displ0"]).max()) < 1e-4
            _check_grads(lambda k: sd[k].grad, g, meta, tol_sel=2e-3, tol_norm=2e-3)
        opt.step()
        sched.step()
        opt.zero_grad()
        lrs.append(opt.param_groups[0]["lr"])
        for k, v in sd.pop(O.BN_UPDATES).items():
            sd[k] = v
        if s_ == 0:
            for k in meta["keys"]:
                got = sample_flat(sd[k].detach().numpy(), meta["sample_cap"])
                # one AdamW step at lr * 0.01: |update| <= lr0; equality to a fraction of that update
                assert float(np.abs(got - g["param1:" + k]).max()) < 0.05 * lrs[0] + 1e-7, k
    assert np.allclose(lrs, g["lrs"], rtol=1e-12, atol=0)
    _check_bn(sd, g, meta, 1e-4)
    for k in meta["keys"]:
        got = sample_flat(sd[k].detach().numpy(), meta["sample_cap"])
        step_sum = float(np.sum(g["lrs"][:-1]))
        assert float(np.abs(got - g["param:" + k]).max()) < 0.1 * step_sum + 1e-7, k


def test_oracle_epoch_with_mixup_matches_the_reference():
    """model.py:193-332 restated with the oracle: two micro-batches (the second one mixed up), acc_grad_iter=2, one
    AdamW + scheduler step; the loss epoch() returns and the parameters it leaves."""
    meta, g = load_golden(EPOCH)
    cfg = meta["cfg"]
    spec = regnet_spec(cfg["feature_arch"])
    sd, par = _oracle_state(cfg, meta["seed_w"])
    masks = drop_masks(meta["mask_seed"], meta["B"], cfg["clip_len"], spec.feat_dim, 2)
    opt = torch.optim.AdamW([sd[k] for k in par], lr=meta["lr"])
    sched = chained_scheduler(opt, meta["warm_steps"], meta["cos_steps"])
    random.seed(meta["random_seed"])
    K1 = cfg["num_classes"] + 1
    total = 0.0
    for i in range(2):
        b = _inputs(meta, meta["seed_x"] + 10 * i, with_second=(i == 1))
        frame, lab, labD, soft = b["frame"].float(), b["label"], b["labelD"].float(), None
        if "frame2" in b:
            lam = [random.betavariate(0.2, 0.2) for _ in range(meta["B"])]
            assert np.allclose(lam, g["lam"])
            lt = torch.tensor(lam, dtype=torch.float32)
            frame = lt.view(-1, 1, 1, 1, 1) * frame + (1 - lt).view(-1, 1, 1, 1, 1) * b["frame2"].float()
            oh = torch.nn.functional.one_hot
            soft = (lt.view(-1, 1, 1) * oh(lab, K1).float() + (1 - lt).view(-1, 1, 1) * oh(b["label2"], K1).float())
            labD = lt.view(-1, 1) * labD + (1 - lt).view(-1, 1) * b["labelD2"].float()
        sd[O.BN_UPDATES] = {}
        loss, _, _ = oracle_train_loss(frame, sd, cfg, spec, lab, labD, masks, None, False, soft=soft)
        (loss / meta["acc_grad_iter"]).backward()
        total += float(loss.detach())
        for k, v in sd.pop(O.BN_UPDATES).items():
            sd[k] = v
    opt.step()
    sched.step()
    assert abs(total / 2 - float(g["loss"])) < 2e-5 * abs(float(g["loss"]))
    assert abs(opt.param_groups[0]["lr"] - float(g["lr_after"])) < 1e-15
    _check_bn(sd, g, meta, 1e-4)
    for k in meta["keys"]:
        got = sample_flat(sd[k].detach().numpy(), meta["sample_cap"])
        assert float(np.abs(got - g["param:" + k]).max()) < 0.05 * meta["lr"] * 0.01 + 1e-7, k


# ----------------------------------------------------------------------------------------------- GPU: HIP engine vs reference
def _hip_model(meta):
    from tdeed_amd.model import TDEEDModel
    cfg = meta["cfg"]
    m = TDEEDModel(device="cuda", args=cfg_ns(cfg))
    m.load({k: t(v) for k, v in model_state(cfg, meta["seed_w"]).items()})
    m._train_dtype = torch.float32
    spec = regnet_spec(cfg["feature_arch"])
    mk = drop_masks(meta["mask_seed"], meta["B"], cfg["clip_len"], spec.feat_dim, 2)
    m._model.dropout_mask_fn = lambda B, T, C, n: mk[:n]
    m._model.augment_fn = lambda x, crop: x                      # the fixture's draw: no RandomApply fires, no flip
    return m


@pytest.mark.gpu
def test_hip_train_steps_match_the_reference():
    """fp32 HIP engine, three optimisation steps on the fixture's batch: loss per step, the first step's gradients (every
    tensor), the parameters after one and after three steps, BatchNorm buffers, the LR schedule."""
    meta, g = load_golden(STEP)
    cfg = meta["cfg"]
    m = _hip_model(meta)
    opt, scaler = m.get_optimizer({"lr": meta["lr"]})
    assert scaler is None
    sched = chained_scheduler(opt, meta["warm_steps"], meta["cos_steps"])
    eng = opt.engine
    inp = _inputs(meta, meta["seed_x"])
    B, T = meta["B"], cfg["clip_len"]
    fr = inp["frame"].cuda()
    lab, labD = inp["label"].cuda().reshape(-1).contiguous(), inp["labelD"].float().cuda().reshape(-1).contiguous()
    m._model.train()
    lrs = [opt.param_groups[0]["lr"]]
    for s_ in range(meta["n_steps"]):
        pred, _ = m._model(fr, inference=True)                  # train-mode module, centre-crop branch (model.py:119-129)
        head = pred["_head_out"].reshape(B * T, -1)
        loss, dhead = eng.temporal.loss_fwd_bwd(head, B, T, lab, labelD=labD, soft=None, fg_weight=5, dataset=None)
        eng.backward_and_write(m._model._train_ctx, dhead, scale=1.0, first=True, reduce=False)
        m._model._train_ctx = None
        torch.cuda.synchronize()
        assert abs(float(loss[0]) - g["losses"][s_]) < (1e-4 if s_ == 0 else 1e-2) * abs(g["losses"][s_]), (s_, float(loss[0]))
        if s_ == 0:
            assert float((pred["im_feat"].cpu() - t(g["logits0"])).abs().max()) < 1e-3
            assert float((pred["displ_feat"].cpu() - t(g["displ0"])).abs().max()) < 1e-3
            # per tensor within 2 % of its own norm (the bound tests/test_gpu_bwd.py holds the same engine to against autograd:
            # BatchNorm layers that see B*T*h*w = 32..8k samples at this size amplify fp32 summation-order differences)
            _check_grads(lambda k: eng.params.grad_view(k), g, meta, tol_sel=2e-2, tol_norm=2e-2, floor=2e-5)
        opt.step()
        sched.step()
        opt.zero_grad()
        lrs.append(opt.param_groups[0]["lr"])
        if s_ == 0:
            sd = m.state_dict()
            for k in meta["keys"]:
                got = sample_flat(sd[k].detach().cpu().numpy(), meta["sample_cap"])
                # the first Adam step moves every entry by lr * sign(g): an entry whose gradient sits in the fp32 noise may go
                # the other way (2 lr apart), the bulk must agree to a fraction of the step
                d = np.abs(got - g["param1:" + k])
                assert float(d.max()) <= 2.2 * lrs[0] + 1e-7 and float(np.median(d)) < 0.1 * lrs[0] + 1e-7, k
                assert float((d > 0.5 * lrs[0]).mean()) < 0.02, k
    assert np.allclose(lrs, g["lrs"], rtol=1e-9, atol=0)
    sd = m.state_dict()
    _check_bn(sd, g, meta, 2e-3)
    step_sum = float(np.sum(g["lrs"][:-1]))
    for k in meta["keys"]:
        got = sample_flat(sd[k].detach().cpu().numpy(), meta["sample_cap"])
        # Adam moves an entry by ~lr per step whatever its gradient's size: entries whose gradient sits in the fp32 noise
        # may differ by a whole step, the bulk must not
        d = np.abs(got - g["param:" + k])
        assert float(d.max()) <= 2.2 * step_sum and float(np.median(d)) < 0.05 * step_sum, (k, float(d.max()), float(np.median(d)))


@pytest.mark.gpu
def test_hip_epoch_with_mixup_matches_the_reference():
    """TDEEDModel.epoch(loader, optimizer, lr_scheduler, acc_grad_iter=2) on the HIP fp32 engine against the reference's own
    epoch() on the same two batches (plain + mixup): returned loss, parameters, BatchNorm buffers, LR."""
    meta, g = load_golden(EPOCH)
    m = _hip_model(meta)
    opt, _ = m.get_optimizer({"lr": meta["lr"]})
    sched = chained_scheduler(opt, meta["warm_steps"], meta["cos_steps"])
    loader = [_inputs(meta, meta["seed_x"] + 10 * i, with_second=(i == 1)) for i in range(2)]
    random.seed(meta["random_seed"])
    avg = m.epoch(loader, optimizer=opt, scaler=None, lr_scheduler=sched, acc_grad_iter=meta["acc_grad_iter"])
    assert abs(avg - float(g["loss"])) < 2e-4 * abs(float(g["loss"])), avg
    assert abs(opt.param_groups[0]["lr"] - float(g["lr_after"])) < 1e-12
    sd = m.state_dict()
    _check_bn(sd, g, meta, 2e-3)
    lr0 = meta["lr"] * 0.01
    for k in meta["keys"]:
        got = sample_flat(sd[k].detach().cpu().numpy(), meta["sample_cap"])
        d = np.abs(got - g["param:" + k])
        assert float(d.max()) <= 2.2 * lr0 and float(np.median(d)) < 0.05 * lr0, (k, float(d.max()), float(np.median(d)))
