"""Backward kernels of the training path (through the C ABI) against torch autograd on the CPU oracle's fp32
restatement of the same ops.  Needs an MI355X: run with -m gpu."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from helpers import act, module_state, t
from oracle import tdeed_oracle as O

pytestmark = pytest.mark.gpu

DEV = "cuda"
F32_TOL = 2e-4
BF16_TOL = 4e-2


@pytest.fixture(scope="module")
def bops():
    from tdeed_amd import ops_bwd as o, _lib
    _lib.load()
    return o


def rel_err(a, b):
    b = b.detach().cpu().double()
    a = a.detach().cpu().double()
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-6))


def rnd(seed, name, shape, scale=1.0):
    return t(act(seed, name, shape, scale))


def tol(dtype):
    return F32_TOL if dtype == torch.float32 else BF16_TOL


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("M,N,K", [(800, 1472, 368), (200, 368, 1472), (37, 24, 40), (1600, 368, 2208), (5000, 8, 368),
                                   (4100, 24, 40), (9001, 152, 56), (20000, 64, 32)])     # M >= 4096: the transposing-read kernel
def test_wgrad(bops, dtype, M, N, K):
    dY, X = rnd(201, "dy", (M, N)).to(dtype), rnd(202, "x", (M, K)).to(dtype)
    dW, db = bops.wgrad(dY.to(DEV), X.to(DEV))
    assert rel_err(dW, dY.float().T @ X.float()) < 1e-4
    assert rel_err(db, dY.float().sum(0)) < 1e-4
    dW2, _ = bops.wgrad(dY.to(DEV), X.to(DEV), with_bias=False, dW=dW.clone(), accumulate=True)
    assert rel_err(dW2, 2 * (dY.float().T @ X.float())) < 1e-4


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_transpose_and_eltwise(bops, dtype):
    x = rnd(203, "x", (75, 200)).to(dtype)
    assert torch.equal(bops.transpose(x.to(DEV)).cpu(), x.T.contiguous())
    h, g = rnd(204, "h", (64, 96), 2.0).to(dtype), rnd(205, "g", (64, 96)).to(dtype)
    hf = h.float().requires_grad_(True)
    y = F.gelu(hf)
    y.backward(g.float())
    assert rel_err(bops.eltwise(h.to(DEV), None, bops.GELU_FWD).float(), y) < tol(dtype)
    assert rel_err(bops.eltwise(h.to(DEV), g.to(DEV), bops.GELU_BWD).float(), hf.grad) < tol(dtype)
    assert rel_err(bops.eltwise(h.to(DEV), g.to(DEV), bops.ADD).float(), h.float() + g.float()) < tol(dtype)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("rows,C", [(200, 368), (77, 32), (800, 768)])
def test_layernorm_bwd(bops, dtype, rows, C):
    x, dy = rnd(206, "x", (rows, C), 2.0).to(dtype), rnd(207, "dy", (rows, C)).to(dtype)
    w, b = rnd(208, "w", (C,)) * 0.3 + 1.0, rnd(209, "b", (C,), 0.1)
    xr, wr, br = x.float().requires_grad_(True), w.clone().requires_grad_(True), b.clone().requires_grad_(True)
    y = O.channel_layernorm(xr.T[None], wr.view(1, C, 1), br.view(1, C, 1))[0].T      # oracle works on (B,C,T)
    y.backward(dy.float())
    dx, dw, db = bops.layernorm_bwd(x.to(DEV), dy.to(DEV), w.to(DEV))
    assert rel_err(dx.float(), xr.grad) < tol(dtype)
    assert rel_err(dw, wr.grad) < (1e-4 if dtype == torch.float32 else 2e-2)
    assert rel_err(db, br.grad) < 1e-4
    base = rnd(210, "base", (rows, C)).to(dtype)
    dx2, _, _ = bops.layernorm_bwd(x.to(DEV), dy.to(DEV), w.to(DEV), dx=base.clone().to(DEV), accumulate=True)
    assert rel_err(dx2.float(), base.float() + xr.grad) < tol(dtype)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("B,T,C", [(3, 25, 32), (2, 100, 368), (2, 50, 768), (1, 250, 768)])
def test_groupnorm_bwd(bops, dtype, B, T, C):
    x, dy = rnd(211, "x", (B, T, C), 2.0).to(dtype), rnd(212, "dy", (B, T, C)).to(dtype)
    w, b = rnd(213, "w", (C,)) * 0.3 + 1.0, rnd(214, "b", (C,), 0.1)
    xr, wr, br = x.float().requires_grad_(True), w.clone().requires_grad_(True), b.clone().requires_grad_(True)
    y = F.group_norm(xr.permute(0, 2, 1), 16, wr, br, 1e-5).permute(0, 2, 1)
    y.backward(dy.float())
    dx, dw, db = bops.groupnorm_bwd(x.to(DEV), dy.to(DEV), 16, w.to(DEV))
    assert rel_err(dx.float(), xr.grad) < tol(dtype)
    assert rel_err(dw, wr.grad) < (1e-4 if dtype == torch.float32 else 2e-2)
    assert rel_err(db, br.grad) < 1e-4


def _branch_ref(o, sd, pre):
    """the depthwise-branch part of SGPBlock.forward (modules.py:164-170) without the residual"""
    psi = O._dw(o, sd, pre + ".psi")
    fc = O._dw(o, sd, pre + ".fc")
    cw = O._dw(o, sd, pre + ".convw")
    ckw = O._dw(o, sd, pre + ".convkw")
    phi = torch.relu(O._dw(o.mean(dim=-1, keepdim=True), sd, pre + ".global_fc"))
    return fc * phi + (cw + ckw) * psi + o


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("B,C,T,ks,r", [(2, 32, 25, 7, 4), (2, 368, 100, 7, 4), (1, 48, 13, 5, 2), (1, 64, 250, 9, 4)])
def test_sgp_branch_bwd(bops, dtype, B, C, T, ks, r):
    from tdeed_amd.engine import _dwpack
    sd = {k: t(v).clone().requires_grad_(True) for k, v in module_state("sgp_block", "blk", 31, C=C, ks=ks, r=r).items()}
    names = ["psi", "convw", "convkw", "fc", "global_fc"]
    up = sd["blk.convkw.weight"].shape[-1]
    o = rnd(215, "o", (B, T, C)).to(dtype)
    dy = rnd(216, "dy", (B, T, C)).to(dtype)
    orq = o.float().requires_grad_(True)
    out = _branch_ref(orq.permute(0, 2, 1), sd, "blk").permute(0, 2, 1)
    out.backward(dy.float())
    dw, db = _dwpack({k: v.detach() for k, v in sd.items()}, "blk", names, C, DEV)
    d_o, ddw, ddb = bops.sgp_branch_bwd(o.to(DEV), dy.to(DEV), ks, up, dw, db)
    assert rel_err(d_o.float(), orq.grad) < tol(dtype)
    ref_w = torch.cat([sd[f"blk.{n}.weight"].grad.reshape(C, -1) for n in names], dim=1)
    ref_b = torch.stack([sd[f"blk.{n}.bias"].grad.reshape(C) for n in names], dim=0)
    assert rel_err(ddw, ref_w) < (2e-4 if dtype == torch.float32 else 2e-2)
    assert rel_err(ddb, ref_b) < (2e-4 if dtype == torch.float32 else 2e-2)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_upsample_and_maxpool_bwd(bops, dtype):
    for (T_hi, T_lo) in [(25, 13), (100, 50), (250, 125), (13, 13)]:
        g = rnd(217, "g", (2, T_hi, 32)).to(dtype)
        xr = rnd(218, "x", (2, T_lo, 32)).float().requires_grad_(True)
        O.upsample_linear(xr.permute(0, 2, 1), T_hi).permute(0, 2, 1).backward(g.float())
        assert rel_err(bops.upsample_bwd(g.to(DEV), T_lo).float(), xr.grad) < tol(dtype)
    for (T_in, T_out) in [(25, 13), (100, 50), (125, 63), (50, 25)]:
        x = (rnd(219, "x", (2, T_in, 32)) * 4).round().div(4).to(dtype)          # quantised -> ties inside windows
        g = rnd(220, "g", (2, T_out, 32)).to(dtype)
        xr = x.float().requires_grad_(True)
        # nn.AdaptiveMaxPool1d itself (the oracle's slice+amax restatement splits the gradient between tied maxima;
        # the real op, like the kernel, gives it to the first maximum of the window)
        y = F.adaptive_max_pool1d(xr.permute(0, 2, 1), T_out)
        assert torch.equal(y, O.adaptive_max_pool(xr.permute(0, 2, 1), T_out))
        y.permute(0, 2, 1).backward(g.float())
        got = bops.maxpool_bwd(x.to(DEV), g.to(DEV)).float().cpu()
        assert rel_err(got, xr.grad) < tol(dtype)


def _temporal_state(C, T, n, ks, r, K1, radi, seed=41):
    sd = {k: t(v) for k, v in module_state("pyramid", "_temp_fine", seed, C=C, ks=ks, r=r, n=n).items()}
    sd["_pred_fine._fc_out.weight"] = rnd(seed, "hw", (K1, C), 0.2)
    sd["_pred_fine._fc_out.bias"] = rnd(seed, "hb", (K1,), 0.1)
    if radi:
        sd["_pred_displ._fc_out.weight"] = rnd(seed, "dw", (1, C), 0.2)
        sd["_pred_displ._fc_out.bias"] = rnd(seed, "db", (1,), 0.1)
    return sd


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("geom", [dict(B=2, C=32, T=25, n=2, ks=7, r=4, K1=5, radi=2, soft=False, drop=False),
                                  dict(B=2, C=368, T=100, n=2, ks=7, r=4, K1=5, radi=2, soft=False, drop=True),
                                  dict(B=1, C=64, T=50, n=3, ks=9, r=4, K1=13, radi=0, soft=True, drop=True),
                                  dict(B=1, C=48, T=250, n=2, ks=9, r=4, K1=13, radi=4, soft=False, drop=False)])   # SoccerNetBall lengths
def test_temporal_stack_loss_and_grads_match_autograd(dtype, geom):
    """SGP encoder-decoder + heads (+dropout) + CE/MSE: loss, every parameter gradient and d loss / d features against
    torch autograd on the CPU oracle."""
    from tdeed_amd.temporal_train import TemporalStack
    from tdeed_amd import synth
    g = geom
    B, C, T, K1 = g["B"], g["C"], g["T"], g["K1"]
    cfg = dict(n_layers=g["n"], clip_len=T, num_classes=K1 - 1, radi_displacement=g["radi"])
    sd = _temporal_state(C, T, g["n"], g["ks"], g["r"], K1, g["radi"])
    feat = rnd(221, "feat", (B, T, C)).to(dtype)
    lab_np, labD_np = synth.labels(222, B, T, K1 - 1, max(g["radi"], 1), fg_frac=0.3)
    lab = t(lab_np).long()
    labD = t(labD_np).float() if g["radi"] else None
    soft = None
    if g["soft"]:
        a = torch.nn.functional.one_hot(lab, K1).float()
        b2 = torch.nn.functional.one_hot(t(synth.labels(224, B, T, K1 - 1, 1, fg_frac=0.3)[0]).long(), K1).float()
        soft = 0.7 * a + 0.3 * b2
    masks = None
    if g["drop"]:
        masks = [(rnd(225 + i, "m", (B, T, C)) > 0).float() * 2.0 for i in range(2 if g["radi"] else 1)]
    # ---- reference: autograd on the oracle (fp32, inputs rounded like the device tensors)
    sdr = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    fr = feat.float().requires_grad_(True)
    enc = O.ed_sgp_mixer(fr, sdr, g["n"], T)
    dm = None if masks is None else ((masks[1] if g["radi"] else None), masks[0])
    cls, displ = O.heads(enc, sdr, g["radi"], drop_mask=dm)
    ref_loss = O.loss_fn(cls, soft if g["soft"] else lab, displ, labD)
    ref_loss.backward()
    # ---- device
    sdd = {k: v.to(DEV) for k, v in sd.items()}
    ts = TemporalStack(sdd, cfg, act_dtype=dtype)
    loss, grads, d_feat = ts.loss_and_grads(
        feat.to(DEV), None if g["soft"] else lab.reshape(-1).to(DEV), labelD=None if labD is None else labD.reshape(-1).to(DEV),
        soft=None if soft is None else soft.reshape(-1, K1).contiguous().to(DEV),
        drop_masks=None if masks is None else [m.to(dtype).to(DEV) for m in masks])
    lt = 1e-4 if dtype == torch.float32 else 3e-2
    assert abs(float(loss[0]) - float(ref_loss.detach())) < lt * max(1.0, abs(float(ref_loss.detach())))
    assert set(grads) == set(sd), set(sd) ^ set(grads)
    if dtype == torch.float32:      # element-wise: max error relative to the tensor's largest entry
        assert rel_err(d_feat.float(), fr.grad) < 2e-3
        worst = max((rel_err(grads[k], sdr[k].grad), k) for k in sd)
        assert worst[0] < 2e-3, worst
    else:                           # bf16 activations and activation gradients: norm-wise
        l2 = lambda a, b: float((a.detach().cpu().double() - b.double()).norm() / b.double().norm().clamp_min(1e-12))  # noqa: E731
        assert l2(d_feat.float(), fr.grad) < 0.1        # ~10 modules deep, every gradient tensor rounded to bf16
        # per tensor (sums with heavy cancellation, e.g. conv biases, carry the bf16 noise of the whole chain) ...
        worst = max((l2(grads[k], sdr[k].grad), k) for k in sd)
        assert worst[0] < 0.3, worst
        # ... and the whole gradient vector
        ga = torch.cat([grads[k].detach().cpu().double().reshape(-1) for k in sd])
        gr = torch.cat([sdr[k].grad.double().reshape(-1) for k in sd])
        assert float((ga - gr).norm() / gr.norm()) < 3e-2


# ============================================================================= train-mode trunk pieces
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("M,C,relu,res", [(3000, 24, True, False), (5001, 152, True, True), (700, 368, False, False),
                                          (40000, 32, True, True)])
def test_bn_train_fwd_bwd(bops, dtype, M, C, relu, res):
    z = (rnd(231, "z", (M, C), 1.5) + rnd(232, "mu", (C,), 0.5)).to(dtype)
    dy = rnd(233, "dy", (M, C)).to(dtype)
    r = rnd(234, "r", (M, C)).to(dtype) if res else None
    w, b = rnd(235, "w", (C,)) * 0.3 + 1.0, rnd(236, "b", (C,), 0.2)
    rm, rv = rnd(237, "rm", (C,), 0.1), rnd(238, "rv", (C,)).abs() + 0.5
    zr, wr, br = z.float().requires_grad_(True), w.clone().requires_grad_(True), b.clone().requires_grad_(True)
    rr = r.float().requires_grad_(True) if res else None
    rm_ref, rv_ref = rm.clone(), rv.clone()
    y = F.batch_norm(zr, rm_ref, rv_ref, wr, br, True, 0.1, 1e-5)
    if res:
        y = y + rr
    if relu:
        y = torch.relu(y)
    y.backward(dy.float())
    rm_d, rv_d = rm.clone().to(DEV), rv.clone().to(DEV)
    out, ctx = bops.bn_train(z.to(DEV), w.to(DEV), b.to(DEV), run_mean=rm_d, run_var=rv_d, res=None if r is None else r.to(DEV),
                             relu=relu)
    assert rel_err(out.float(), y) < tol(dtype)
    assert rel_err(rm_d, rm_ref) < 1e-4 and rel_err(rv_d, rv_ref) < 1e-4
    dz, d_res, dw, db = bops.bn_train_bwd(z.to(DEV), dy.to(DEV), out, ctx, w.to(DEV), relu=relu, want_res=res)
    assert rel_err(dz.float(), zr.grad) < tol(dtype)
    assert rel_err(dw, wr.grad) < (2e-4 if dtype == torch.float32 else 2e-2)
    assert rel_err(db, br.grad) < (2e-4 if dtype == torch.float32 else 2e-2)
    if res:
        assert rel_err(d_res.float(), rr.grad) < tol(dtype)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("N,h,w,C,R", [(6, 7, 7, 368, 92), (3, 28, 28, 56, 6), (2, 14, 14, 152, 38), (2, 56, 56, 24, 8)])
def test_se_train_fwd_bwd(bops, dtype, N, h, w, C, R):
    """squeeze -> fc1/ReLU/fc2/sigmoid -> scale, and its backward, against autograd on the oracle's _se."""
    x = rnd(241, "x", (N, h, w, C)).to(dtype)
    dy = rnd(242, "dy", (N, h, w, C)).to(dtype)
    sd = {"se.fc1.weight": rnd(243, "w1", (R, C, 1, 1), 0.1), "se.fc1.bias": rnd(244, "b1", (R,), 0.1),
          "se.fc2.weight": rnd(245, "w2", (C, R, 1, 1), 0.2), "se.fc2.bias": rnd(246, "b2", (C,), 0.1)}
    sdr = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    xr = x.float().permute(0, 3, 1, 2).requires_grad_(True)
    y = O._se(xr, sdr, "se")
    y.backward(dy.float().permute(0, 3, 1, 2))
    w1, w2 = sd["se.fc1.weight"].reshape(R, C), sd["se.fc2.weight"].reshape(C, R)
    xd, dyd = x.to(DEV), dy.to(DEV)
    p = bops.pool_rows(xd)
    hid, gate = bops.se_train_fwd(p, w1.T.contiguous().to(DEV), sd["se.fc1.bias"].to(DEV), w2.T.contiguous().to(DEV),
                                  sd["se.fc2.bias"].to(DEV))
    ys = bops.scale_rows(xd, gate)
    assert rel_err(ys.float().permute(0, 3, 1, 2), y) < tol(dtype)
    d_gate = bops.pool_rows(dyd, xd)
    d_pre2, d_hid, d_p = bops.se_train_bwd(d_gate, gate, hid, w1.to(DEV), w2.to(DEV))
    dx = bops.scale_rows(dyd, gate, add=d_p, add_scale=1.0 / (h * w))
    dW2, db2 = bops.wgrad(d_pre2, hid)
    dW1, db1 = bops.wgrad(d_hid, p)
    assert rel_err(dx.float().permute(0, 3, 1, 2), xr.grad) < tol(dtype)
    gt = 5e-4 if dtype == torch.float32 else 3e-2
    assert rel_err(dW1, sdr["se.fc1.weight"].grad.reshape(R, C)) < gt
    assert rel_err(db1, sdr["se.fc1.bias"].grad) < gt
    assert rel_err(dW2, sdr["se.fc2.weight"].grad.reshape(C, R)) < gt
    assert rel_err(db2, sdr["se.fc2.bias"].grad) < gt


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("C,gw,stride,H,W", [(24, 8, 2, 20, 22), (56, 8, 1, 9, 7), (64, 16, 2, 16, 16), (152, 8, 2, 28, 28),
                                             (368, 8, 1, 7, 7), (128, 16, 1, 14, 14)])
def test_gconv3x3_raw_fwd_and_bwd(bops, dtype, C, gw, stride, H, W):
    from tdeed_amd import ops
    N = 3
    x = rnd(251, "x", (N, C, H, W)).to(dtype)
    wt = rnd(252, "w", (C, gw, 3, 3), 0.2)
    xr, wr = x.float().requires_grad_(True), wt.clone().requires_grad_(True)
    y = F.conv2d(xr, wr, stride=stride, padding=1, groups=C // gw)
    dy = rnd(253, "dy", tuple(y.shape)).to(dtype)
    y.backward(dy.float())
    G = C // gw
    wp = wt.reshape(G, gw, gw, 3, 3).permute(0, 3, 4, 2, 1).reshape(G, 9, gw, gw).contiguous().to(DEV)
    one, zero = torch.ones(C, device=DEV), torch.zeros(C, device=DEV)
    xd = x.permute(0, 2, 3, 1).contiguous().to(DEV)
    yraw, _ = ops.gconv3x3(xd, wp, one, zero, gw, stride, relu=False)          # VALU path, no activation
    assert rel_err(yraw.float().permute(0, 3, 1, 2), y) < tol(dtype)
    dx, dw = bops.gconv3x3_bwd(xd, dy.permute(0, 2, 3, 1).contiguous().to(DEV), wp, gw, stride)
    assert rel_err(dx.float().permute(0, 3, 1, 2), xr.grad) < tol(dtype)
    ref_dw = wr.grad.reshape(G, gw, gw, 3, 3).permute(0, 3, 4, 2, 1).reshape(G, 9, gw, gw)
    assert rel_err(dw, ref_dw) < (2e-4 if dtype == torch.float32 else 2e-2)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_stride2_rows(bops, dtype):
    x = rnd(261, "x", (3, 9, 12, 24)).to(dtype)
    g = bops.stride2_gather(x.to(DEV))
    assert torch.equal(g.cpu(), x[:, ::2, ::2].contiguous())
    big = rnd(262, "b", (3, 9, 12, 24)).to(dtype)
    ref = big.float().clone()
    ref[:, ::2, ::2] += g.cpu().float()
    out = bops.stride2_scatter_add(g, big.clone().to(DEV))
    assert rel_err(out.float(), ref) < tol(dtype)


def _block_state(blk, seed=51):
    from collections import OrderedDict
    from tdeed_amd import synth
    d = OrderedDict()
    pre = "blk"
    def conv_bn(n, co, ci, k):
        d[f"{pre}.{n}.conv.weight"] = ((co, ci, k, k), "float32")
        for s_ in ("weight", "bias", "running_mean", "running_var"):
            d[f"{pre}.{n}.bn.{s_}"] = ((co,), "float32")
    if blk.gsf_fold:
        from tdeed_amd import state_layout
        conv_bn("conv1.net", blk.cout, blk.cin, 1)
        state_layout._gate_shift(d, pre + ".conv1.gs", blk.gsf_fold, "gsf")
    else:
        conv_bn("conv1", blk.cout, blk.cin, 1)
    conv_bn("conv2", blk.cout, blk.gw, 3)
    conv_bn("conv3", blk.cout, blk.cout, 1)
    if blk.has_downsample:
        conv_bn("downsample", blk.cout, blk.cin, 1)
    d[pre + ".se.fc1.weight"] = ((blk.se_rd, blk.cout, 1, 1), "float32")
    d[pre + ".se.fc1.bias"] = ((blk.se_rd,), "float32")
    d[pre + ".se.fc2.weight"] = ((blk.cout, blk.se_rd, 1, 1), "float32")
    d[pre + ".se.fc2.bias"] = ((blk.cout,), "float32")
    return {k: t(v) for k, v in synth.make_state(d, seed).items()}


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("geom", [dict(cin=32, cout=24, stride=2, gw=8, N=3, h=20, w=24),      # s1.b1-like
                                  dict(cin=24, cout=56, stride=2, gw=8, N=2, h=14, w=14),      # s2.b1-like
                                  dict(cin=152, cout=152, stride=1, gw=8, N=4, h=7, w=7),      # identity block
                                  dict(cin=64, cout=128, stride=1, gw=16, N=2, h=6, w=10),     # 1x1 shortcut, stride 1
                                  dict(cin=152, cout=152, stride=1, gw=8, N=8, h=7, w=7, fold=40, T=4),     # s3.b2-like, GSF
                                  dict(cin=56, cout=152, stride=2, gw=8, N=6, h=10, w=12, fold=16, T=3)])   # s3.b1-like, GSF
def test_bottleneck_train_fwd_bwd_matches_autograd(dtype, geom):
    """A whole RegNetY bottleneck in training mode (batch-stat BN, SE, shortcut): output, running statistics, every
    parameter gradient and d x against autograd on the CPU oracle's regnet_block(training=True)."""
    from tdeed_amd.trunk_train import BottleneckTrain
    from tdeed_amd.regnet_spec import BlockSpec
    g = geom
    blk = BlockSpec(name="blk", stage=1, index=1, cin=g["cin"], cout=g["cout"], stride=g["stride"], groups=g["cout"] // g["gw"],
                    gw=g["gw"], se_rd=int(round(g["cin"] * 0.25)), has_downsample=(g["cin"] != g["cout"] or g["stride"] != 1),
                    gsf_fold=g.get("fold", 0), hin=g["h"])
    T = g.get("T", 1)
    sd = _block_state(blk)
    x = rnd(271, "x", (g["N"], g["cin"], g["h"], g["w"])).abs().to(dtype)
    sdr = {k: (v.clone().requires_grad_(True) if ("running" not in k and v.dtype == torch.float32) else v.clone())
           for k, v in sd.items()}
    xr = x.float().requires_grad_(True)
    ref = O.regnet_block(xr, sdr, "blk", blk, T, "gsf", training=True)
    dy = rnd(272, "dy", tuple(ref.shape)).to(dtype)
    ref.backward(dy.float())
    sdd = {k: v.clone().to(DEV) for k, v in sd.items()}
    bt = BottleneckTrain(sdd, "blk", blk, act_dtype=dtype, clip_len=T)
    out = bt.forward(x.permute(0, 2, 3, 1).contiguous().to(DEV))
    assert rel_err(out.float().permute(0, 3, 1, 2), ref) < (2e-4 if dtype == torch.float32 else 4e-2)
    grads = {}
    dx = bt.backward(dy.permute(0, 2, 3, 1).contiguous().to(DEV), grads)
    want = {k for k, v in sd.items() if "running" not in k and v.dtype == torch.float32}
    assert set(grads) == want, set(grads) ^ want
    if dtype == torch.float32:
        assert rel_err(dx.float().permute(0, 3, 1, 2), xr.grad) < 2e-3
        worst = max((rel_err(grads[k], sdr[k].grad), k) for k in want)
        assert worst[0] < 2e-3, worst
    else:
        l2 = lambda a, b: float((a.detach().cpu().double() - b.double()).norm() / b.double().norm().clamp_min(1e-12))  # noqa: E731
        assert l2(dx.float().permute(0, 3, 1, 2), xr.grad) < 0.1
        ga = torch.cat([grads[k].detach().cpu().double().reshape(-1) for k in sorted(want)])
        gr = torch.cat([sdr[k].grad.double().reshape(-1) for k in sorted(want)])
        assert float((ga - gr).norm() / gr.norm()) < 0.1      # batch-stat BN over a few hundred bf16 samples per channel


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("geom", [dict(C=152, F=40, B=2, T=6, h=5, w=7), dict(C=56, F=16, B=1, T=5, h=9, w=8),
                                  dict(C=368, F=92, B=2, T=4, h=7, w=7), dict(C=152, F=40, B=2, T=6, h=5, w=7, mode="gsm")])
def test_gate_shift_train_fwd_bwd_matches_autograd(dtype, geom):
    """GatedShift + _GSF in training mode (BatchNorm3d batch statistics): module output, parameter gradients and d x
    against autograd on the CPU oracle's gate_shift(training=True)."""
    from tdeed_amd.trunk_train import GateShiftTrain
    g = geom
    C, F, B, T, h, w = g["C"], g["F"], g["B"], g["T"], g["h"], g["w"]
    Fp = (F + 7) // 8 * 8
    N = B * T
    mode = g.get("mode", "gsf")                  # "gsm": the plain gate-shift module (impl/gsm.py), no fusion conv
    sd = {k: t(v) for k, v in module_state("gate_shift", "gs", 61, F=F, mode=mode).items()}
    x = rnd(281, "x", (N, C, h, w)).to(dtype)
    sdr = {k: (v.clone().requires_grad_(True) if v.dtype == torch.float32 and "running" not in k else v.clone()) for k, v in sd.items()}
    xr = x.float().requires_grad_(True)
    ref = O.gate_shift(xr[:, :F], sdr, "gs", T, mode, training=True)           # (N, F, h, w)
    dy = rnd(282, "dy", (N, F, h, w)).to(dtype)
    ref.backward(dy.float())
    sdd = {k: v.clone().to(DEV) for k, v in sd.items()}
    gs = GateShiftTrain(sdd, "gs", F, T, dtype)
    xd = x.permute(0, 2, 3, 1).contiguous().to(DEV)
    G = gs.forward(xd)
    assert rel_err(G[:, :F].float().view(N, h, w, F).permute(0, 3, 1, 2), ref) < (2e-4 if dtype == torch.float32 else 4e-2)
    dA = torch.zeros((N * h * w, Fp), dtype=dtype)
    dA[:, :F] = dy.permute(0, 2, 3, 1).reshape(-1, F)
    grads = {}
    d_xs, dz, _ = gs.backward(dA.to(DEV), grads)
    dx = (d_xs.float() + dz.float())[:, :F].view(N, h, w, F).permute(0, 3, 1, 2)
    want = {k for k, v in sd.items() if v.dtype == torch.float32 and "running" not in k}
    assert set(grads) == want, set(grads) ^ want
    if dtype == torch.float32:
        assert rel_err(dx, xr.grad[:, :F]) < 2e-3
        worst = max((rel_err(grads[k], sdr[k].grad), k) for k in want)
        assert worst[0] < 2e-3, worst
    else:
        l2 = lambda a, b: float((a.detach().cpu().double() - b.double()).norm() / b.double().norm().clamp_min(1e-12))  # noqa: E731
        assert l2(dx, xr.grad[:, :F]) < 0.1
        ga = torch.cat([grads[k].detach().cpu().double().reshape(-1) for k in sorted(want)])
        gr = torch.cat([sdr[k].grad.double().reshape(-1) for k in sorted(want)])
        assert float((ga - gr).norm() / gr.norm()) < 0.1
        # the BatchNorm3d backward left to the kernel that adds the module's input gradient into d x (statistics from the
        # conv3d input-gradient launch, apply on load): the same d x columns and BatchNorm parameter gradients
        from tdeed_amd import ops_bwd as B2
        if B2.gsf_bwd_bn_parts(B, T, h, w, C, Fp) > 0:
            grads2 = {}
            d_xs2, d_bn2, bn = gs.backward(dA.to(DEV), grads2, fused_bn=True)
            assert bn is not None
            M = N * h * w
            dxa = torch.zeros((M, C), dtype=dtype, device=DEV)
            sink = B2.GradSink(torch.ones((M, C), dtype=dtype, device=DEV), torch.zeros((M, C), dtype=dtype, device=DEV),
                               torch.zeros(C, device=DEV))
            B2.gsf_add_cols_sink(d_xs2, d_bn2, dxa, Fp, sink, bn=bn)
            dxb = torch.zeros((M, C), dtype=dtype, device=DEV)
            B2.gsf_add_cols(d_xs, dz, dxb, Fp)
            torch.cuda.synchronize()
            scale = float(dxb.float().abs().max())
            assert float((dxa.float() - dxb.float()).abs().max()) <= 2e-2 * scale
            for k in ("gs.bn.weight", "gs.bn.bias"):
                a, b_ = B2.materialize(grads2[k]).float(), B2.materialize(grads[k]).float()
                assert float((a - b_).abs().max()) <= 5e-3 * max(1.0, float(b_.abs().max())), k


def _oracle_train_loss(frames, sd, cfg, spec, lab, labD, masks, crop, flip):
    """the training-branch forward of TDEEDModel on the CPU oracle (batch-stat BN, dropout masks, one shared crop)"""
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
    f = O.regnet_features(x.reshape(B * T, *x.shape[2:]), sd, spec, T, "gsf", training=True)
    f = f.reshape(B, T, -1) + sd["temp_enc"][None]
    enc = O.ed_sgp_mixer(f, sd, cfg["n_layers"], cfg["clip_len"])
    dm = None if masks is None else (masks[1], masks[0])
    cls, displ = O.heads(enc, sd, cfg["radi_displacement"], drop_mask=dm)
    return O.loss_fn(cls, lab, displ, labD)


def test_full_train_step_800mf_matches_autograd():
    """The RegNetY-800MF variant (group width 16, folds 32/80/192, n_layers 3) through the same check, fp32."""
    from tdeed_amd import synth, state_layout
    from tdeed_amd.regnet_spec import regnet_spec
    from tdeed_amd.trainer import TrainEngine
    cfg = dict(feature_arch="rny008_gsf", clip_len=4, crop_dim=None, n_layers=3, sgp_ks=7, sgp_r=4, num_classes=5,
               radi_displacement=0)
    B, T, H, W = 2, cfg["clip_len"], 64, 64
    sd0 = {k: t(v) for k, v in synth.make_state(state_layout.model_state_shapes(cfg), 9).items()}
    frames = t(synth.uint8_clip(361, (B, T, 3, H, W)))
    lab = t(synth.labels(362, B, T, cfg["num_classes"], 1, fg_frac=0.4)[0]).long()
    par = [k for k in sd0 if state_layout.is_parameter(k)]
    sdr = {k: (v.clone().requires_grad_(True) if k in par else v.clone()) for k, v in sd0.items()}
    ref_loss = _oracle_train_loss(frames, sdr, cfg, regnet_spec(cfg["feature_arch"]), lab, None, None, None, False)
    ref_loss.backward()
    eng = TrainEngine(cfg, {k: v.clone() for k, v in sd0.items()}, act_dtype=torch.float32, lr=1e-3)
    loss, grads = eng.loss_and_grads(frames.to(DEV), lab.to(DEV))
    assert set(grads) == set(par)
    assert abs(float(loss[0]) - float(ref_loss.detach())) < 2e-4 * max(1.0, abs(float(ref_loss.detach())))
    ga = torch.cat([grads[k].detach().cpu().double().reshape(-1) for k in par])
    gr = torch.cat([sdr[k].grad.double().reshape(-1) for k in par])
    assert float((ga - gr).norm() / gr.norm()) < 2e-3
    eng.step(frames.to(DEV), lab.to(DEV))          # and the optimiser launch over the 800MF flat buffer
    torch.cuda.synchronize()


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_full_train_step_matches_torch(dtype):
    """One whole optimisation step (uint8 clip -> loss -> every gradient -> AdamW) of a small RegNetY-200MF+GSF+SGP model
    against autograd + torch.optim.AdamW on the CPU oracle."""
    from tdeed_amd import synth, state_layout
    from tdeed_amd.regnet_spec import regnet_spec
    from tdeed_amd.trainer import TrainEngine
    cfg = dict(feature_arch="rny002_gsf", clip_len=6, crop_dim=None, n_layers=2, sgp_ks=5, sgp_r=2, num_classes=3,
               radi_displacement=2)
    B, T, H, W = 2, cfg["clip_len"], 72, 80
    crop, flip = (4, 8, 64, 64), False
    sd0 = {k: t(v) for k, v in synth.make_state(state_layout.model_state_shapes(cfg), 7).items()}
    frames = t(synth.uint8_clip(301, (B, T, 3, H, W)))
    lab_np, labD_np = synth.labels(302, B, T, cfg["num_classes"], cfg["radi_displacement"], fg_frac=0.4)
    lab, labD = t(lab_np).long(), t(labD_np).float()
    C = regnet_spec(cfg["feature_arch"]).feat_dim
    masks = [(rnd(303 + i, "m", (B, T, C)) > 0).float() * 2.0 for i in range(2)]
    # ---- reference
    par = [k for k in sd0 if state_layout.is_parameter(k)]
    sdr = {k: (v.clone().requires_grad_(True) if k in par else v.clone()) for k, v in sd0.items()}
    ref_loss = _oracle_train_loss(frames, sdr, cfg, regnet_spec(cfg["feature_arch"]), lab, labD, masks, crop, flip)
    ref_loss.backward()
    opt = torch.optim.AdamW([sdr[k] for k in par], lr=1e-3)
    ref_grads = {k: sdr[k].grad.clone() for k in par}
    opt.step()
    # ---- device
    eng = TrainEngine(cfg, {k: v.clone() for k, v in sd0.items()}, act_dtype=dtype, lr=1e-3)
    loss, grads = eng.loss_and_grads(frames.to(DEV), lab.to(DEV), labD.to(DEV), crop=crop, flip=flip,
                                     drop_masks=[m.to(dtype).to(DEV) for m in masks])
    assert set(grads) == set(par), set(grads) ^ set(par)
    lt = 2e-4 if dtype == torch.float32 else 5e-2
    assert abs(float(loss[0]) - float(ref_loss.detach())) < lt * max(1.0, abs(float(ref_loss.detach())))
    ga = torch.cat([grads[k].detach().cpu().double().reshape(-1) for k in par])
    gr = torch.cat([ref_grads[k].double().reshape(-1) for k in par])
    gerr = float((ga - gr).norm() / gr.norm())
    if dtype == torch.float32:
        assert gerr < 2e-3, gerr
        # per tensor: within 2 % of its own norm, plus a floor of 1e-4 of the whole gradient for near-zero tensors
        # (single-scalar biases that are sums with heavy cancellation)
        gn = float(gr.norm())
        for k in par:
            d = float((grads[k].detach().cpu().double() - ref_grads[k].double()).norm())
            assert d <= 2e-2 * float(ref_grads[k].double().norm()) + 1e-4 * gn, (k, d, float(ref_grads[k].norm()))
    else:
        # bf16 maps through 13 bottlenecks whose BatchNorms see only B*T*h*w = 48..3k samples at this toy size: the
        # direction of the gradient is what can be asserted
        cos = float((ga * gr).sum() / (ga.norm() * gr.norm()))
        assert gerr < 0.45 and cos > 0.9, (gerr, cos)
    if dtype == torch.float32:
        # the optimiser: same step on both sides (re-run the step through the engine's own path)
        eng2 = TrainEngine(cfg, {k: v.clone() for k, v in sd0.items()}, act_dtype=dtype, lr=1e-3)
        eng2.step(frames.to(DEV), lab.to(DEV), labD.to(DEV), crop=crop, flip=flip,
                  drop_masks=[m.to(dtype).to(DEV) for m in masks])
        # the first Adam step moves every entry by ~lr * sign(g): compare where the sign of g is not in the noise
        for k in par:
            gref = ref_grads[k]
            sel = gref.abs() > 5e-2 * gref.abs().max()
            got, want_ = eng2.state[k].detach().cpu()[sel], sdr[k].detach()[sel]
            assert float((got - want_).abs().max()) < 2e-5 + 1e-4 * float(want_.abs().max()), k
            assert float((eng2.state[k].detach().cpu() - sd0[k]).abs().max()) < 1.2e-3     # |update| <= lr (+ decay)
        nbt = [k for k in sd0 if k.endswith("num_batches_tracked")]
        assert all(int(eng2.state[k]) == int(sd0[k]) + 1 for k in nbt)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_mix_frames_and_float_frame_stem(bops, dtype):
    """mixup (model.py:246) feeds fp32 frames: the stem and its weight gradient read them like the uint8 ones."""
    from tdeed_amd import ops, synth
    B, T, H, W = 2, 3, 40, 48
    a, b = t(synth.uint8_clip(311, (B, T, 3, H, W))), t(synth.uint8_clip(312, (B, T, 3, H, W)))
    lam = torch.tensor([0.3, 0.85])
    mixed = bops.mix_frames(a.to(DEV), b.to(DEV), lam.to(DEV))
    ref_mix = lam.view(B, 1, 1, 1, 1) * a.float() + (1 - lam).view(B, 1, 1, 1, 1) * b.float()
    assert rel_err(mixed, ref_mix) < 1e-6
    wt = rnd(313, "w", (32, 3, 3, 3), 0.3)
    crop = (2, 4, 32, 40)
    x = O.preprocess(ref_mix, None)[..., 2:34, 4:44].reshape(B * T, 3, 32, 40)
    wr = wt.clone().requires_grad_(True)
    y = F.conv2d(x, wr, stride=2, padding=1)
    dz = rnd(314, "dz", tuple(y.shape)).to(dtype)
    y.backward(dz.float())
    one, zero = torch.ones(32, device=DEV), torch.zeros(32, device=DEV)
    fr = mixed.view(B * T, 3, H, W)
    z = ops.stem(fr, wt.to(DEV), one, zero, dtype, crop=crop, relu=False)
    assert rel_err(z.float().permute(0, 3, 1, 2), y) < tol(dtype)
    dw = bops.stem_wgrad(fr, dz.permute(0, 2, 3, 1).contiguous().to(DEV), crop=crop)
    assert rel_err(dw, wr.grad) < (2e-4 if dtype == torch.float32 else 2e-2)
    # and the uint8 path of the same kernels
    x8 = O.preprocess(a, None)[..., 2:34, 4:44].reshape(B * T, 3, 32, 40)
    wr2 = wt.clone().requires_grad_(True)
    y8 = F.conv2d(x8, wr2, stride=2, padding=1)
    y8.backward(dz.float())
    dw8 = bops.stem_wgrad(a.view(B * T, 3, H, W).to(DEV), dz.permute(0, 2, 3, 1).contiguous().to(DEV), crop=crop)
    assert rel_err(dw8, wr2.grad) < (2e-4 if dtype == torch.float32 else 2e-2)


def test_train_step_graph_replay_matches_eager():
    """The captured step (weight re-pack + forward + loss + backward + gradient write-out in one HIP graph, AdamW outside)
    walks the same trajectory as the eager step."""
    from tdeed_amd import synth, state_layout
    from tdeed_amd.regnet_spec import regnet_spec
    from tdeed_amd.trainer import TrainEngine
    cfg = dict(feature_arch="rny002_gsf", clip_len=6, crop_dim=None, n_layers=2, sgp_ks=5, sgp_r=2, num_classes=3,
               radi_displacement=2)
    B, T, H, W = 2, cfg["clip_len"], 64, 64
    sd0 = {k: t(v) for k, v in synth.make_state(state_layout.model_state_shapes(cfg), 7).items()}
    C = regnet_spec(cfg["feature_arch"]).feat_dim
    data = []
    for i in range(3):
        lab_np, labD_np = synth.labels(330 + i, B, T, cfg["num_classes"], cfg["radi_displacement"], fg_frac=0.4)
        data.append((t(synth.uint8_clip(320 + i, (B, T, 3, H, W))).to(DEV), t(lab_np).long().to(DEV), t(labD_np).float().to(DEV),
                     [((rnd(340 + i + 10 * j, "m", (B, T, C)) > 0).float() * 2.0).to(DEV) for j in range(2)]))
    e1 = TrainEngine(cfg, {k: v.clone() for k, v in sd0.items()}, act_dtype=torch.float32, lr=1e-3)
    e2 = TrainEngine(cfg, {k: v.clone() for k, v in sd0.items()}, act_dtype=torch.float32, lr=1e-3)
    h = e2.build_graph(B, H, W)
    for fr, lab, labD, masks in data:
        l1 = e1.step(fr, lab, labD, drop_masks=masks)
        l2 = e2.step_graph(h, fr, lab, labD, drop_masks=masks)
        assert abs(float(l1[0]) - float(l2[0])) < 1e-5 * max(1.0, abs(float(l1[0])))
    for k in sd0:
        a, b = e1.state[k], e2.state[k]
        assert torch.equal(a, b) or rel_err(b.float(), a.float()) < 1e-5, k


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_double_head_loss_and_grads_match_autograd(dtype):
    """Joint-dataset training (model.py:278-306): two class heads side by side, every clip scored on its dataset's head."""
    from tdeed_amd.temporal_train import TemporalStack
    from tdeed_amd import synth
    B, C, T, n, ks, r, k1a, k1b, radi = 3, 32, 25, 2, 7, 4, 4, 6, 2
    cfg = dict(n_layers=n, clip_len=T, num_classes=k1a - 1, radi_displacement=radi)
    sd = {k: t(v) for k, v in module_state("pyramid", "_temp_fine", 43, C=C, ks=ks, r=r, n=n).items()}
    for nm, k in (("_pred_fine._fc1._fc_out", k1a), ("_pred_fine._fc2._fc_out", k1b), ("_pred_displ._fc_out", 1)):
        sd[nm + ".weight"], sd[nm + ".bias"] = rnd(44, nm + "w", (k, C), 0.2), rnd(44, nm + "b", (k,), 0.1)
    feat = rnd(351, "feat", (B, T, C)).to(dtype)
    ds = torch.tensor([1, 2, 2])
    lab = torch.stack([t(synth.labels(352 + i, 1, T, (k1a if ds[i] == 1 else k1b) - 1, 1, fg_frac=0.3)[0][0]).long() for i in range(B)])
    labD = t(synth.labels(355, B, T, 3, radi)[1]).float()
    masks = [(rnd(356 + i, "m", (B, T, C)) > 0).float() * 2.0 for i in range(3)]
    # reference: the loop of model.py:284-306 on the oracle's heads
    sdr = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    fr = feat.float().requires_grad_(True)
    enc = O.ed_sgp_mixer(fr, sdr, n, T)
    fc = lambda nm, m: F.linear(enc * m, sdr[nm + ".weight"], sdr[nm + ".bias"])          # noqa: E731
    pred = torch.cat([fc("_pred_fine._fc1._fc_out", masks[0]), fc("_pred_fine._fc2._fc_out", masks[1])], dim=2)
    predD = fc("_pred_displ._fc_out", masks[2]).squeeze(-1)
    wgt = torch.tensor([1.0] + [5.0] * (max(k1a, k1b) - 1))
    ref = 0.0
    for i in range(B):
        if ds[i] == 1:
            ref = ref + F.cross_entropy(pred[i][:, :k1a], lab[i], weight=wgt[:k1a]) / B
        else:
            ref = ref + F.cross_entropy(pred[i][:, k1a:], lab[i], weight=wgt[:k1b]) / B
    ref = ref + F.mse_loss(predD, labD, reduction="none").mean()
    ref.backward()
    # device: labels of dataset 2 arrive shifted (update_labels_2heads)
    lab_shift = lab.clone()
    lab_shift[ds == 2] += k1a
    ts = TemporalStack({k: v.to(DEV) for k, v in sd.items()}, cfg, act_dtype=dtype)
    loss, grads, d_feat = ts.loss_and_grads(feat.to(DEV), lab_shift.reshape(-1).to(DEV), labelD=labD.reshape(-1).to(DEV),
                                            drop_masks=[m.to(dtype).to(DEV) for m in masks], dataset=ds.to(DEV))
    assert abs(float(loss[0]) - float(ref.detach())) < (1e-4 if dtype == torch.float32 else 3e-2) * max(1.0, abs(float(ref.detach())))
    assert set(grads) == set(sd)
    ga = torch.cat([grads[k].detach().cpu().double().reshape(-1) for k in sd])
    gr = torch.cat([sdr[k].grad.double().reshape(-1) for k in sd])
    assert float((ga - gr).norm() / gr.norm()) < (2e-3 if dtype == torch.float32 else 4e-2)
    if dtype == torch.float32:
        assert rel_err(d_feat.float(), fr.grad) < 2e-3
        for k in [k for k in sd if k.startswith("_pred")]:
            assert rel_err(grads[k], sdr[k].grad) < 1e-3, k


def test_gradient_write_out_folds_partials_and_reads_column_slices(bops):
    """tdeed_multi_fold: one launch writes contiguous tensors, column slices of wider matrices and still-unfolded
    weight-gradient partials (LazyFold) into the flat buffer, scaled, overwriting or accumulating."""
    g = torch.Generator().manual_seed(7)
    a = torch.randn(300, generator=g).to(DEV)
    wide = torch.randn(7, 50, generator=g).to(DEV)
    part = torch.randn(600, 96, generator=g).to(DEV)              # many partials of a narrow output: 8 columns per workgroup
    part2 = torch.randn(3, 5000, generator=g).to(DEV)             # few partials: 64 columns per workgroup
    part3 = torch.randn(100, 40, generator=g).to(DEV)
    srcs = [a, wide[:, 10:25].reshape(7, 1, 15), bops.LazyFold(part, 600, 96, (12, 8)), bops.LazyFold(part2, 3, 5000),
            wide[:, 3:4].reshape(7, 1, 1), bops.LazyFold(part3, 100, 40).reshape(5, 8)]
    sizes = [300, 105, 96, 5000, 7, 40]
    offs, cur = [], 4
    for n in sizes:
        offs.append(cur)
        cur += (n + 3) // 4 * 4
    flat = torch.full((cur + 8,), 0.5, device=DEV)
    ref = flat.clone()
    want = [a, wide[:, 10:25].reshape(-1), part.sum(0), part2.sum(0), wide[:, 3], part3.sum(0)]
    for o, w_ in zip(offs, want):
        ref[o:o + w_.numel()] = 2.0 * w_.reshape(-1)
    bops.multi_copy(srcs, offs, flat, scale=2.0, accumulate=False)
    assert rel_err(flat, ref) < 1e-5
    bops.multi_copy(srcs, offs, flat, scale=2.0, accumulate=True)
    ref2 = ref.clone()
    for o, w_ in zip(offs, want):
        ref2[o:o + w_.numel()] += 2.0 * w_.reshape(-1)
    assert rel_err(flat, ref2) < 1e-5
    assert torch.equal(srcs[2].materialize(), srcs[2].materialize()) and srcs[2].materialize().shape == (12, 8)
    assert rel_err(srcs[2].materialize().reshape(-1), part.sum(0)) < 1e-5


def test_lazy_weight_gradients_give_the_same_step(monkeypatch):
    """TDEED_LAZY_WGRAD=0 (every weight gradient folded by its own launch) against the default (folded by the bucket's
    write-out launch): the same sums in the same order -- bit-identical parameters after two optimizer steps."""
    from tdeed_amd.trainer import TrainEngine
    from tdeed_amd import synth, state_layout
    cfg = dict(feature_arch="rny002_gsf", clip_len=8, crop_dim=None, n_layers=2, sgp_ks=5, sgp_r=2, num_classes=3,
               radi_displacement=2)
    B, T, H, W = 2, 8, 64, 64
    sd0 = {k: t(v) for k, v in synth.make_state(state_layout.model_state_shapes(cfg), 47).items()}
    frames = t(synth.uint8_clip(831, (B, T, 3, H, W))).to(DEV)
    lab_np, labD_np = synth.labels(832, B, T, cfg["num_classes"], cfg["radi_displacement"], fg_frac=0.4)
    lab, labD = t(lab_np).long().to(DEV), t(labD_np).float().to(DEV)
    res = {}
    for flag in ("1", "0"):
        monkeypatch.setenv("TDEED_LAZY_WGRAD", flag)
        eng = TrainEngine(cfg, {k: v.clone() for k, v in sd0.items()}, act_dtype=torch.bfloat16, lr=1e-3)
        for _ in range(2):
            loss = eng.step(frames, lab, labD)
        torch.cuda.synchronize()
        res[flag] = (loss.clone(), eng.params.flat.clone(), eng.params.grad.clone())
    assert torch.equal(res["1"][0], res["0"][0])
    assert torch.equal(res["1"][2], res["0"][2]) and torch.equal(res["1"][1], res["0"][1])


def test_shortcut_gradient_joins_in_the_contraction_epilogue(monkeypatch):
    """TDEED_TRAIN_FUSE_RES=1 (default: the identity shortcut's gradient is the residual operand of conv1's input-gradient
    contraction, the gate-shift columns leave it as a compact tensor) against 0 (separate add / slice copy / zero fill): the
    same sums with one bf16 rounding instead of two -- losses equal, gradients within bf16 rounding of each other."""
    from tdeed_amd.trainer import TrainEngine
    from tdeed_amd import synth, state_layout
    import tdeed_amd.trunk_train as TT
    cfg = dict(feature_arch="rny002_gsf", clip_len=8, crop_dim=None, n_layers=2, sgp_ks=5, sgp_r=2, num_classes=3,
               radi_displacement=2)
    B, T, H, W = 2, 8, 96, 96
    sd0 = {k: t(v) for k, v in synth.make_state(state_layout.model_state_shapes(cfg), 61).items()}
    frames = t(synth.uint8_clip(871, (B, T, 3, H, W))).to(DEV)
    lab_np, labD_np = synth.labels(872, B, T, cfg["num_classes"], cfg["radi_displacement"], fg_frac=0.4)
    lab, labD = t(lab_np).long().to(DEV), t(labD_np).float().to(DEV)
    res = {}
    for flag in (True, False):
        monkeypatch.setattr(TT, "FUSE_RES", flag)
        for dt in (torch.float32, torch.bfloat16):
            eng = TrainEngine(cfg, {k: v.clone() for k, v in sd0.items()}, act_dtype=dt, lr=1e-3)
            loss = eng.step(frames, lab, labD)
            torch.cuda.synchronize()
            res[(flag, dt)] = (loss.clone(), eng.params.grad.clone())
    for dt, tol in ((torch.float32, 1e-5), (torch.bfloat16, 3e-2)):
        a, b = res[(True, dt)], res[(False, dt)]
        assert torch.equal(a[0], b[0])                          # the forward is untouched
        assert rel_err(a[1], b[1]) < tol, (dt, rel_err(a[1], b[1]))


def test_conv1_operand_splice_gives_the_same_step(monkeypatch):
    """s3 / s4 conv1 of a gate-shift block: contraction and weight gradient reading [G | x[:, Fp:]] as two sources
    (TDEED_TRAIN_SPLICE=1, default, for M >= 4096 rows) against the materialised operand: bit-identical step."""
    from tdeed_amd.trainer import TrainEngine
    from tdeed_amd import synth, state_layout
    cfg = dict(feature_arch="rny002_gsf", clip_len=24, crop_dim=None, n_layers=2, sgp_ks=5, sgp_r=2, num_classes=3,
               radi_displacement=2)
    B, T, H, W = 1, 24, 224, 224                                 # s3: 24 frames x 14 x 14 = 4704 rows
    sd0 = {k: t(v) for k, v in synth.make_state(state_layout.model_state_shapes(cfg), 53).items()}
    frames = t(synth.uint8_clip(841, (B, T, 3, H, W))).to(DEV)
    lab_np, labD_np = synth.labels(842, B, T, cfg["num_classes"], cfg["radi_displacement"], fg_frac=0.4)
    lab, labD = t(lab_np).long().to(DEV), t(labD_np).float().to(DEV)
    res = {}
    for flag in ("1", "0"):
        monkeypatch.setenv("TDEED_TRAIN_SPLICE", flag)
        eng = TrainEngine(cfg, {k: v.clone() for k, v in sd0.items()}, act_dtype=torch.bfloat16, lr=1e-3)
        loss = eng.step(frames, lab, labD)
        torch.cuda.synchronize()
        spliced = [b.ctx.G is not None for b in eng.blocks if b.gs is not None]
        assert any(spliced) == (flag == "1")
        res[flag] = (loss.clone(), eng.params.grad.clone(), eng.params.flat.clone())
    assert torch.equal(res["1"][0], res["0"][0])
    assert torch.equal(res["1"][1], res["0"][1]) and torch.equal(res["1"][2], res["0"][2])


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_gradient_accumulation_over_micro_batches(dtype):
    """`acc_grad_iter` of the reference's step() (model.py:321-332: loss / acc_grad_iter, optimizer step every acc_grad_iter
    batches): two accumulate() calls with scale 1/2 (overwrite, then add -- the bucket write-out folds the weight-gradient
    partials and accumulates in the same launch) leave the mean of the two micro-batch gradients in the flat buffer."""
    from tdeed_amd.trainer import TrainEngine
    from tdeed_amd import synth, state_layout
    cfg = dict(feature_arch="rny002_gsf", clip_len=8, crop_dim=None, n_layers=2, sgp_ks=5, sgp_r=2, num_classes=3,
               radi_displacement=2)
    B, T, H, W = 2, 8, 64, 64
    sd0 = {k: t(v) for k, v in synth.make_state(state_layout.model_state_shapes(cfg), 59).items()}
    batches = []
    for i in range(2):
        frames = t(synth.uint8_clip(851 + i, (B, T, 3, H, W))).to(DEV)
        lab_np, labD_np = synth.labels(861 + i, B, T, cfg["num_classes"], cfg["radi_displacement"], fg_frac=0.4)
        batches.append((frames, t(lab_np).long().to(DEV), t(labD_np).float().to(DEV)))
    eng = TrainEngine(cfg, {k: v.clone() for k, v in sd0.items()}, act_dtype=dtype, lr=1e-3)
    eng.accumulate(*batches[0], scale=0.5, first=True)
    eng.accumulate(*batches[1], scale=0.5, first=False)
    torch.cuda.synchronize()
    acc = eng.params.grad.clone()
    ref = TrainEngine(cfg, {k: v.clone() for k, v in sd0.items()}, act_dtype=dtype, lr=1e-3)
    want = torch.zeros_like(acc)
    for b in batches:
        _, grads = ref.loss_and_grads(*b)
        for k, (o, n) in ref.params.index.items():
            want[o:o + n] += 0.5 * grads[k].reshape(-1)
    torch.cuda.synchronize()
    assert rel_err(acc, want) < (1e-5 if dtype == torch.float32 else 1e-5)
