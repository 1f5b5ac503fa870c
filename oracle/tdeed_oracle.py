"""CPU oracle for the T-DEED hot path -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

A from-scratch fp32 restatement (torch CPU ops) of what the reference computes in
``TDEEDModel.Impl.forward`` and the loss in ``TDEEDModel.epoch``.  Only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import this
module, and only as the checker / the reported CPU baseline.  The product path
(``tdeed_amd``) never imports it and has no CPU fallback.

Pinning: functions that restate code living under /root/reference (pre-proc,
GatedShift, GSF, GSM, SGP pyramid, heads, loss, process_prediction) are checked
against the reference itself, imported in the build container by
``tools/make_goldens.py``, through the fixtures in ``tests/golden/`` (see
``tests/test_oracle_golden.py``).  The RegNetY trunk lives in the un-vendored
third-party package ``timm==1.0.3`` (reference ``requirements.txt:39``, call site
``model/model.py:38-45``) which is absent here: for it PARITY IS UNPINNED by any
reference test or vector; it is restated from the published RegNetY design and
cross-checked against the independent HuggingFace ``RegNetYLayer`` implementation
(``tools/make_goldens.py``, fixture ``regnet_hf_*.npz``) and by parameter counts.

The functions take a flat ``state_dict`` with the reference's key grammar
 so that real checkpoints drive oracle and product alike.
"""
import math
import torch
import torch.nn.functional as F

IMAGENET_MEAN = (0.485, 0.456, 0.406)
IMAGENET_STD = (0.229, 0.224, 0.225)
BN_EPS = 1e-5
BN_UPDATES = "__bn_updates__"     # optional dict in a state: train-mode forwards leave the new BatchNorm buffers there


# ----------------------------------------------------------------------------- helpers
def _t(v):
    return v if isinstance(v, torch.Tensor) else torch.as_tensor(v)


def as_torch_state(sd):
    return {k: _t(v) for k, v in sd.items()}


def _bn(x, sd, pre, training, dims_per_channel=None):
    """BatchNorm (2d/3d share the formula); eval: running stats, train: biased batch stats."""
    w, b = sd[pre + ".weight"], sd[pre + ".bias"]
    if training:
        upd = sd.get(BN_UPDATES)
        if upd is not None:
            # what nn.BatchNorm2d/3d leaves in its buffers after a train-mode forward (torch defaults, as constructed by
            # timm / model/impl/gsf.py:26): momentum 0.1, UNBIASED batch variance, num_batches_tracked += 1
            with torch.no_grad():
                red = [d for d in range(x.dim()) if d != 1]
                n = x.numel() // x.shape[1]
                m = x.detach().mean(dim=red)
                v = x.detach().var(dim=red, unbiased=False) * (n / max(n - 1, 1))
                upd[pre + ".running_mean"] = 0.9 * sd[pre + ".running_mean"] + 0.1 * m
                upd[pre + ".running_var"] = 0.9 * sd[pre + ".running_var"] + 0.1 * v
                upd[pre + ".num_batches_tracked"] = sd[pre + ".num_batches_tracked"] + 1
        return F.batch_norm(x, None, None, w, b, True, 0.0, BN_EPS)
    return F.batch_norm(x, sd[pre + ".running_mean"], sd[pre + ".running_var"], w, b, False, 0.0, BN_EPS)


# ----------------------------------------------------------------------------- pre-proc
def preprocess(frames, crop_dim=None, augment_inference=False):
    """model/model.py:107-129 (inference branch), 151-152, 159-167.

    frames: (B,T,3,H,W) holding 0..255 (uint8 or float).  /255, centre crop, optional
    horizontal flip (test-time view), ImageNet standardisation."""
    x = frames.to(torch.float32) / 255.0
    B, T, C, H, W = x.shape
    if crop_dim is not None and crop_dim > 0 and (crop_dim != H or crop_dim != W):
        top = int(round((H - crop_dim) / 2.0))
        left = int(round((W - crop_dim) / 2.0))
        x = x[..., top:top + crop_dim, left:left + crop_dim]
    if augment_inference:
        x = x.flip(-1)
    mean = torch.tensor(IMAGENET_MEAN, dtype=torch.float32).view(1, 1, 3, 1, 1)
    std = torch.tensor(IMAGENET_STD, dtype=torch.float32).view(1, 1, 3, 1, 1)
    return (x - mean) / std




# ----------------------------------------------------------------------------- train-time augmentation
# model/model.py:76-83, 154-157 apply torchvision transforms (requirements.txt:41, torchvision==0.18.1, NOT vendored and not
# installed here: PARITY UNPINNED for this section) per clip on the cropped 0..1 frames.  Restated from the published
# float-tensor functional ops (torchvision/transforms/_functional_tensor.py: _rgb2hsv, _hsv2rgb, _blend, adjust_*,
# gaussian_blur).
def _gray(img):
    r, g, b = img.unbind(dim=-3)
    return (0.2989 * r + 0.587 * g + 0.114 * b).unsqueeze(-3)


def _blend(a, b, ratio):
    return (ratio * a + (1.0 - ratio) * b).clamp(0, 1.0)


def adjust_hue(img, f):
    r, g, b = img.unbind(dim=-3)
    maxc, minc = img.max(dim=-3).values, img.min(dim=-3).values
    eqc = maxc == minc
    cr = maxc - minc
    ones = torch.ones_like(maxc)
    s_ = cr / torch.where(eqc, ones, maxc)
    crd = torch.where(eqc, ones, cr)
    rc, gc, bc = (maxc - r) / crd, (maxc - g) / crd, (maxc - b) / crd
    hr = (maxc == r) * (bc - gc)
    hg = ((maxc == g) & (maxc != r)) * (2.0 + rc - bc)
    hb = ((maxc != g) & (maxc != r)) * (4.0 + gc - rc)
    h = torch.fmod((hr + hg + hb) / 6.0 + 1.0, 1.0)
    h = (h + f) % 1.0
    v = maxc
    i = torch.floor(h * 6.0)
    fr = h * 6.0 - i
    i = i.to(torch.int32) % 6
    p_ = (v * (1.0 - s_)).clamp(0, 1)
    q_ = (v * (1.0 - s_ * fr)).clamp(0, 1)
    t_ = (v * (1.0 - s_ * (1.0 - fr))).clamp(0, 1)
    sel = lambda opts: sum((i == k) * o for k, o in enumerate(opts))    # noqa: E731
    return torch.stack((sel((v, q_, p_, p_, t_, v)), sel((t_, v, v, q_, p_, p_)), sel((p_, p_, t_, v, v, q_))), dim=-3)


def gaussian_blur5(img, sigma):
    half = 2.0
    x = torch.linspace(-half, half, 5)
    pdf = torch.exp(-0.5 * (x / sigma) ** 2)
    k1 = pdf / pdf.sum()
    k2 = torch.outer(k1, k1)
    C = img.shape[-3]
    pad = F.pad(img, (2, 2, 2, 2), mode="reflect")
    return F.conv2d(pad, k2[None, None].expand(C, 1, 5, 5).contiguous(), groups=C)


def augment_clip(x01, prm):
    """x01 (T,3,h,w) in 0..1; prm = (hue, sat, bri, con, sigma) with identity (0,1,1,1,0): the colour / blur stages of the
    reference's `self.augmentation` in Compose order (the flip is applied by the caller)."""
    hue, sat, bri, con, sigma = [float(v) for v in prm[:5]]
    x = x01
    if hue != 0.0:
        x = adjust_hue(x, hue)
    if sat != 1.0:
        x = _blend(x, _gray(x), sat)
    if bri != 1.0:
        x = _blend(x, torch.zeros_like(x), bri)
    if con != 1.0:
        x = _blend(x, _gray(x).mean(dim=(-3, -2, -1), keepdim=True), con)
    if sigma > 0.0:
        x = gaussian_blur5(x, sigma)
    return x
# ----------------------------------------------------------------------------- GSF / GSM
def _shift_left(y):   # y[t] <- y[t+1], last = 0        (gsf.py:28-31)
    return torch.cat([y[:, :, 1:], torch.zeros_like(y[:, :, :1])], dim=2)


def _shift_right(y):  # y[t] <- y[t-1], first = 0       (gsf.py:33-36)
    return torch.cat([torch.zeros_like(y[:, :, :1]), y[:, :, :-1]], dim=2)


def _interleave(y, Bn, Fh, T, h, w):
    # channel c = i*(Fh/2) + j  ->  2*j + i  inside one half   (gsf.py:83-86)
    return y.reshape(Bn, 2, Fh // 2, T, h, w).permute(0, 2, 1, 3, 4, 5).reshape(Bn, Fh, T, h, w)


def gate_shift(x, sd, pre, T, mode="gsf", training=False, taps=None):
    """``_GSF.forward`` (model/impl/gsf.py:38-93) / ``_GSM.forward`` (model/impl/gsm.py:89-116)
    on the first ``fold`` channels.  x: (B*T, F, h, w) -> same shape."""
    N, Fp, h, w = x.shape
    Bn = N // T
    Fh = Fp // 2
    v = x.reshape(Bn, T, Fp, h, w).permute(0, 2, 1, 3, 4)            # (B,F,T,h,w)
    a = torch.relu(_bn(v, sd, pre + ".bn", training))
    gate = torch.tanh(F.conv3d(a, sd[pre + ".conv3D.weight"], sd[pre + ".conv3D.bias"],
                               stride=1, padding=1, groups=2))        # (B,2,T,h,w)
    x1, x2 = v[:, :Fh], v[:, Fh:]
    y1, y2 = gate[:, 0:1] * x1, gate[:, 1:2] * x2
    r1, r2 = x1 - y1, x2 - y2
    y1, y2 = _shift_left(y1), _shift_right(y2)
    if mode == "gsm":
        o1, o2 = y1 + r1, y2 + r2
    else:
        def fuse(y, r, cpre):
            ym, rm = y.mean(dim=(3, 4)), r.mean(dim=(3, 4))          # (B,Fh,T)
            plane = torch.stack([ym, rm], dim=1)                     # (B,2,Fh,T)
            wgt = torch.sigmoid(F.conv2d(plane, sd[cpre + ".weight"], sd[cpre + ".bias"], padding=1))
            wgt = wgt[:, 0, :, :, None, None]                        # (B,Fh,T,1,1)
            if taps is not None:
                taps.setdefault(cpre, wgt[..., 0, 0].detach())
            return y * wgt + r * (1.0 - wgt)
        o1 = fuse(y1, r1, pre + ".channel_conv1")
        o2 = fuse(y2, r2, pre + ".channel_conv2")
    o = torch.cat([_interleave(o1, Bn, Fh, T, h, w), _interleave(o2, Bn, Fh, T, h, w)], dim=1)
    if taps is not None:
        taps.setdefault(pre + ".gate", gate.detach())
    return o.permute(0, 2, 1, 3, 4).reshape(N, Fp, h, w)


# ----------------------------------------------------------------------------- RegNetY
def _conv_bn(x, sd, pre, stride=1, groups=1, relu=True, training=False):
    wgt = sd[pre + ".conv.weight"]
    x = F.conv2d(x, wgt, None, stride=stride, padding=wgt.shape[-1] // 2, groups=groups)
    x = _bn(x, sd, pre + ".bn", training)
    return torch.relu(x) if relu else x


def _se(x, sd, pre):
    s = x.mean(dim=(2, 3), keepdim=True)
    s = torch.relu(F.conv2d(s, sd[pre + ".fc1.weight"], sd[pre + ".fc1.bias"]))
    s = torch.sigmoid(F.conv2d(s, sd[pre + ".fc2.weight"], sd[pre + ".fc2.bias"]))
    return x * s


def regnet_block(x, sd, pre, blk, T, shift_mode, training=False, taps=None):
    """One RegNetY bottleneck; ``blk`` is a tdeed_amd.regnet_spec.BlockSpec (plain attributes)."""
    short = x
    if blk.gsf_fold > 0:
        # GatedShift.forward (model/shift.py:89-93): only the first fold channels go through the gate
        Fd = blk.gsf_fold
        g = gate_shift(x[:, :Fd], sd, pre + ".conv1.gs", T, shift_mode, training, taps)
        y = torch.cat([g, x[:, Fd:]], dim=1)
        if taps is not None:
            taps[pre + ".gs_out"] = g.detach()
        y = _conv_bn(y, sd, pre + ".conv1.net", training=training)
    else:
        y = _conv_bn(x, sd, pre + ".conv1", training=training)
    y = _conv_bn(y, sd, pre + ".conv2", stride=blk.stride, groups=blk.groups, training=training)
    y = _se(y, sd, pre + ".se")
    y = _conv_bn(y, sd, pre + ".conv3", relu=False, training=training)
    if blk.has_downsample:
        short = _conv_bn(short, sd, pre + ".downsample", stride=blk.stride, relu=False, training=training)
    out = torch.relu(y + short)
    if taps is not None:
        taps[pre] = out.detach()
    return out


def regnet_features(x, sd, spec, T, shift_mode="gsf", pre="_features.", training=False, taps=None):
    """x: (N,3,H,W) standardised -> (N, C) pooled features (head.fc = Identity, model/model.py:45)."""
    x = _conv_bn(x, sd, pre + "stem", stride=2, training=training)
    if taps is not None:
        taps[pre + "stem"] = x.detach()
    for blk in spec.blocks:
        x = regnet_block(x, sd, pre + blk.name, blk, T, shift_mode, training, taps)
    return x.mean(dim=(2, 3))


# ----------------------------------------------------------------------------- SGP pyramid
def channel_layernorm(x, w, b, eps=1e-5):
    """model/modules.py:320-363 on (B,C,T): statistics over C, biased variance, eps inside sqrt."""
    mu = x.mean(dim=1, keepdim=True)
    r = x - mu
    var = (r * r).mean(dim=1, keepdim=True)
    return r / torch.sqrt(var + eps) * w + b


def _dw(x, sd, pre):
    wgt = sd[pre + ".weight"]
    return F.conv1d(x, wgt, sd[pre + ".bias"], padding=wgt.shape[-1] // 2, groups=wgt.shape[0])


def _mlp(x, sd, pre):
    h = F.gelu(F.conv1d(x, sd[pre + ".mlp.0.weight"], sd[pre + ".mlp.0.bias"]))
    return F.conv1d(h, sd[pre + ".mlp.2.weight"], sd[pre + ".mlp.2.bias"])


def sgp_block(x, sd, pre):
    """SGPBlock.forward, mode='normal' (model/modules.py:159-188).  x: (B,C,T)."""
    o = channel_layernorm(x, sd[pre + ".ln.weight"], sd[pre + ".ln.bias"])
    psi = _dw(o, sd, pre + ".psi")
    fc = _dw(o, sd, pre + ".fc")
    cw = _dw(o, sd, pre + ".convw")
    ckw = _dw(o, sd, pre + ".convkw")
    phi = torch.relu(_dw(o.mean(dim=-1, keepdim=True), sd, pre + ".global_fc"))
    y = x + (fc * phi + (cw + ckw) * psi + o)
    g = F.group_norm(y, 16, sd[pre + ".gn.weight"], sd[pre + ".gn.bias"], 1e-5)
    return y + _mlp(g, sd, pre)


def upsample_linear(x, t_out):
    """nn.Upsample(size, mode='linear', align_corners=True) (model/modules.py:236)."""
    return F.interpolate(x, size=t_out, mode="linear", align_corners=True)


def sgp_mixer(x, z, sd, pre, t_size):
    """SGPMixer.forward with concat=True (model/modules.py:283-318).  z: skip (B,C,T_hi), x: (B,C,T_lo)."""
    z = channel_layernorm(z, sd[pre + ".ln1.weight"], sd[pre + ".ln1.bias"])
    x = channel_layernorm(x, sd[pre + ".ln2.weight"], sd[pre + ".ln2.bias"])
    x = upsample_linear(x, t_size)
    out1 = (_dw(z, sd, pre + ".convw1") + _dw(z, sd, pre + ".convkw1")) * _dw(z, sd, pre + ".psi1")
    out2 = (_dw(x, sd, pre + ".convw2") + _dw(x, sd, pre + ".convkw2")) * _dw(x, sd, pre + ".psi2")
    out3 = _dw(z, sd, pre + ".fc1") * torch.relu(_dw(z.mean(-1, keepdim=True), sd, pre + ".global_fc1"))
    out4 = _dw(x, sd, pre + ".fc2") * torch.relu(_dw(x.mean(-1, keepdim=True), sd, pre + ".global_fc2"))
    cat = torch.cat([out1, out2, out3, out4, z, x], dim=1)
    o = F.gelu(F.conv1d(cat, sd[pre + ".concat_fc.weight"], sd[pre + ".concat_fc.bias"]))
    g = F.group_norm(o, 16, sd[pre + ".gn.weight"], sd[pre + ".gn.bias"], 1e-5)
    return o + _mlp(g, sd, pre)


def adaptive_max_pool(x, out_len):
    """nn.AdaptiveMaxPool1d: window i = [floor(i*L/O), ceil((i+1)*L/O))."""
    L = x.shape[-1]
    cols = []
    for i in range(out_len):
        lo = (i * L) // out_len
        hi = -((-(i + 1) * L) // out_len)
        cols.append(x[..., lo:hi].amax(dim=-1))
    return torch.stack(cols, dim=-1)


def ed_sgp_mixer(feat, sd, n_layers, clip_len, pre="_temp_fine.", taps=None):
    """EDSGPMIXERLayers.forward (model/modules.py:69-87).  feat: (B,T,C) -> (B,T,C)."""
    x = feat.permute(0, 2, 1)
    stash = []
    for i in range(n_layers):
        x = sgp_block(x, sd, f"{pre}_sgp.{i}")
        if taps is not None:
            taps[f"{pre}_sgp.{i}"] = x.detach().permute(0, 2, 1)
        stash.append(x)
        x = adaptive_max_pool(x, math.ceil(clip_len / 2 ** (i + 1)))
    x = sgp_block(x, sd, f"{pre}_sgp.{n_layers}")
    if taps is not None:
        taps[f"{pre}_sgp.{n_layers}"] = x.detach().permute(0, 2, 1)
    for i in range(n_layers):
        lvl = n_layers - 1 - i
        x = sgp_mixer(x, stash[lvl], sd, f"{pre}_sgpMixer.{lvl}", math.ceil(clip_len / 2 ** lvl))
        if taps is not None:
            taps[f"{pre}_sgpMixer.{lvl}"] = x.detach().permute(0, 2, 1)
        x = sgp_block(x, sd, f"{pre}_sgp.{n_layers + 1 + i}")
        if taps is not None:
            taps[f"{pre}_sgp.{n_layers + 1 + i}"] = x.detach().permute(0, 2, 1)
    return x.permute(0, 2, 1)


# ----------------------------------------------------------------------------- heads / full forward
def heads(x, sd, radi_displacement, double_head=False, drop_mask=None):
    """FCLayers / FC2Layers (model/modules.py:366-387) + the calls at model/model.py:141-146.
    drop_mask: optional (B,T,C) 0/1 keep-mask already scaled by 1/(1-p) per head (train parity)."""
    def fc(pre, m):
        xi = x if m is None else x * m
        return F.linear(xi, sd[pre + "._fc_out.weight"], sd[pre + "._fc_out.bias"])
    md = mc = None
    if drop_mask is not None:
        md, mc = drop_mask
    if double_head:
        cls = torch.cat([fc("_pred_fine._fc1", mc), fc("_pred_fine._fc2", mc)], dim=2)
    else:
        cls = fc("_pred_fine", mc)
    if radi_displacement > 0:
        return cls, fc("_pred_displ", md).squeeze(-1)
    return cls, None


def forward(frames, sd, cfg, spec, augment_inference=False, training=False, taps=None):
    """TDEEDModel.Impl.forward(x, inference=True) (model/model.py:105-149).

    cfg: object/dict with clip_len, crop_dim, n_layers, radi_displacement, feature_arch.
    Returns (logits (B,T,K+1), displ (B,T) | None, feat (B,T,C))."""
    g = (lambda k: cfg[k]) if isinstance(cfg, dict) else (lambda k: getattr(cfg, k))
    sd = as_torch_state(sd)
    x = preprocess(frames, g("crop_dim"), augment_inference)
    B, T = x.shape[:2]
    arch = g("feature_arch")
    mode = "gsm" if arch.endswith("_gsm") else "gsf"
    f = regnet_features(x.reshape(B * T, *x.shape[2:]), sd, spec, T, mode, training=training, taps=taps)
    f = f.reshape(B, T, -1) + sd["temp_enc"][None]
    if taps is not None:
        taps["feat"] = f.detach()
    s = ed_sgp_mixer(f, sd, g("n_layers"), g("clip_len"), taps=taps)
    cls, displ = heads(s, sd, g("radi_displacement"),
                       double_head=("_pred_fine._fc1._fc_out.weight" in sd))
    return cls, displ, f


# ----------------------------------------------------------------------------- loss / post-proc
def loss_fn(logits, label, displ=None, labelD=None, fg_weight=5.0):
    """model/model.py:208-211, 308-319: weighted CE (hard int labels or soft (N,K+1)) + MSE."""
    K1 = logits.shape[-1]
    wgt = torch.tensor([1.0] + [float(fg_weight)] * (K1 - 1))
    lg = logits.reshape(-1, K1).float()
    lab = label.reshape(-1) if label.dim() == 2 and label.dtype == torch.int64 else label.reshape(-1, K1)
    loss = F.cross_entropy(lg, lab, weight=wgt)
    if displ is not None and labelD is not None:
        loss = loss + F.mse_loss(displ.float(), labelD.float(), reduction="none").mean()
    return loss


def process_prediction(pred, predD):
    """model/modules.py:406-414: softmax, then scatter-max of frame t onto t - round(displ)."""
    p = torch.softmax(pred.float(), dim=2)
    B, T, _ = p.shape
    out = torch.zeros_like(p)
    tgt = (torch.arange(T)[None, :] - torch.round(predD.float()).to(torch.int64)).clamp(0, T - 1)
    for b in range(B):
        for t in range(T):
            j = int(tgt[b, t])
            out[b, j] = torch.maximum(out[b, j], p[b, t])
    return out
