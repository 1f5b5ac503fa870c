"""Train-mode forward + backward of the temporal stack: SGP encoder-decoder (`EDSGPMIXERLayers`,
/root/reference/model/modules.py:58-87, blocks 159-188, mixers 283-318), the prediction heads with dropout
(modules.py:366-387) and the loss of `TDEEDModel.epoch` (model/model.py:308-319).

Everything numeric runs in the HIP kernels behind the C ABI (ops.py / ops_bwd.py); this file only orders the launches,
keeps the activations the backward needs and maps the kernels' packed gradient layouts back onto the reference's
state_dict names.  Inputs: trunk features (B, T, C) *after* the positional encoding; outputs: the loss, the gradient of
every `_temp_fine.*`, `_pred_fine.*`, `_pred_displ.*` parameter and the gradient w.r.t. the features, which
`trainer.TrainEngine` hands on to the trunk backward (trunk_train.py)."""
import os
from types import SimpleNamespace

import torch

from . import ops, ops_bwd as B_, repack as R
from .regnet_spec import pyramid_lengths

_BR = ["psi", "convw", "convkw", "fc", "global_fc"]
SPLITK_TRAIN = True


def _flat(v):
    return v.reshape(v.shape[0], -1).contiguous()


class _Dense:
    """A Conv1d(k=1) weight in the activation dtype, plus its transpose for the input-gradient contraction."""

    def __init__(self, w_master, dt):
        w = _flat(w_master)
        self.w = w if dt == torch.float32 else R.cast_bf16(w)
        self.wt = R.transpose(self.w)
        self.N, self.K = w.shape


def _pack_branch(sd, pre, names, C):
    w = torch.cat([sd[f"{pre}.{n}.weight"].reshape(C, -1) for n in names], dim=1).contiguous()
    b = torch.stack([sd[f"{pre}.{n}.bias"].reshape(C) for n in names], dim=0).contiguous()
    return w, b


def _unpack_branch(grads, pre, names, ddw, ddb, sd):
    col = 0
    for i, n in enumerate(names):
        k = sd[f"{pre}.{n}.weight"].shape[-1]
        grads[f"{pre}.{n}.weight"] = ddw[:, col:col + k].reshape(sd[f"{pre}.{n}.weight"].shape)
        grads[f"{pre}.{n}.bias"] = ddb[i].reshape(sd[f"{pre}.{n}.bias"].shape)
        col += k


class TemporalStack:
    """sd: dict name -> fp32 master tensor on the device (reference state_dict names and shapes)."""

    def __init__(self, sd, cfg, act_dtype=torch.float32, prefix="_temp_fine."):
        self.sd, self.dt, self.pre = sd, act_dtype, prefix
        self.C = sd[prefix + "_sgp.0.ln.weight"].numel()
        self.n = cfg["n_layers"]
        self.T = cfg["clip_len"]
        self.K1 = cfg["num_classes"] + 1
        self.radi = cfg.get("radi_displacement", 0)
        self._cls_w = {}
        self.repack()

    # ------------------------------------------------------------------ per-step parameter views
    def repack(self):
        """Kernel-layout views of the master parameters; call after every optimizer step."""
        sd, C, dt = self.sd, self.C, self.dt
        self.blocks, self.mixers = [], []
        for i in range(2 * self.n + 1):
            pre = f"{self.pre}_sgp.{i}"
            o = SimpleNamespace(pre=pre)
            o.ln_w, o.ln_b = sd[pre + ".ln.weight"].reshape(C), sd[pre + ".ln.bias"].reshape(C)
            o.dw, o.db = _pack_branch(sd, pre, _BR, C)
            o.ks, o.up = sd[pre + ".psi.weight"].shape[-1], sd[pre + ".convkw.weight"].shape[-1]
            o.gn_w, o.gn_b = sd[pre + ".gn.weight"], sd[pre + ".gn.bias"]
            o.fc1, o.b1 = _Dense(sd[pre + ".mlp.0.weight"], dt), sd[pre + ".mlp.0.bias"]
            o.fc2, o.b2 = _Dense(sd[pre + ".mlp.2.weight"], dt), sd[pre + ".mlp.2.bias"]
            self.blocks.append(o)
        for i in range(self.n):
            pre = f"{self.pre}_sgpMixer.{i}"
            o = SimpleNamespace(pre=pre)
            for k in ("ln1", "ln2"):
                setattr(o, k + "_w", sd[f"{pre}.{k}.weight"].reshape(C))
                setattr(o, k + "_b", sd[f"{pre}.{k}.bias"].reshape(C))
            o.dw1, o.db1 = _pack_branch(sd, pre, [n + "1" for n in _BR], C)
            o.dw2, o.db2 = _pack_branch(sd, pre, [n + "2" for n in _BR], C)
            o.ks, o.up = sd[pre + ".psi1.weight"].shape[-1], sd[pre + ".convkw1.weight"].shape[-1]
            o.cat, o.bcat = _Dense(sd[pre + ".concat_fc.weight"], dt), sd[pre + ".concat_fc.bias"]
            o.gn_w, o.gn_b = sd[pre + ".gn.weight"], sd[pre + ".gn.bias"]
            o.fc1, o.b1 = _Dense(sd[pre + ".mlp.0.weight"], dt), sd[pre + ".mlp.0.bias"]
            o.fc2, o.b2 = _Dense(sd[pre + ".mlp.2.weight"], dt), sd[pre + ".mlp.2.bias"]
            self.mixers.append(o)

    def _long_k(self, A, W, bias, R, residual=None):
        """A (R, K) @ W (N, K)^T + bias (+ residual) for the long contractions over few rows (fc2, concat_fc, fc1's input
        gradient: K = 4C .. 6C over B*T = 400 .. 1600 rows).  The tiled kernel walks K in ~100 dependent slabs on a few dozen
        workgroups (50 us whatever the row count); the split-K form of the inference path spreads K over the chip."""
        if self.dt == torch.bfloat16 and SPLITK_TRAIN and R <= 4096 and W.shape[1] >= 1024:
            return ops.gemm_splitk(A, W, None, bias, ops.ACT_NONE, residual=residual, M=R)
        return ops.gemm(A, W, None, bias, ops.ACT_NONE, residual=residual, M=R)

    # ------------------------------------------------------------------ mlp (shared by blocks and mixers)
    def _mlp_fwd(self, y, o, ctx):
        Bn, T, C = y.shape
        R = Bn * T
        gn = ops.groupnorm(y, 16, o.gn_w, o.gn_b)
        hpre = ops.gemm(gn, o.fc1.w, None, o.b1, ops.ACT_NONE, M=R).view(Bn, T, 4 * C)
        hid = B_.eltwise(hpre, None, B_.GELU_FWD)
        out = self._long_k(hid, o.fc2.w, o.b2, R, residual=y).view(Bn, T, C)
        ctx.gn, ctx.hpre, ctx.hid = gn, hpre, hid
        return out

    def _mlp_bwd(self, dout, y, o, ctx, grads):
        """dout: gradient of y + mlp(GN(y)); returns the gradient w.r.t. y."""
        Bn, T, C = y.shape
        R = Bn * T
        d_hid = ops.gemm(dout, o.fc2.wt, None, None, ops.ACT_NONE, M=R).view(Bn, T, 4 * C)
        dW2, db2 = B_.wgrad(dout, ctx.hid, M=R)
        d_hpre = B_.eltwise(ctx.hpre, d_hid, B_.GELU_BWD)
        d_gn = self._long_k(d_hpre, o.fc1.wt, None, R).view(Bn, T, C)
        dW1, db1 = B_.wgrad(d_hpre, ctx.gn, M=R)
        d_y = dout.clone()
        _, dgw, dgb = B_.groupnorm_bwd(y, d_gn, 16, o.gn_w, dx=d_y, accumulate=True)
        sd, pre = self.sd, o.pre
        grads[pre + ".mlp.0.weight"] = dW1.reshape(sd[pre + ".mlp.0.weight"].shape)
        grads[pre + ".mlp.0.bias"] = db1
        grads[pre + ".mlp.2.weight"] = dW2.reshape(sd[pre + ".mlp.2.weight"].shape)
        grads[pre + ".mlp.2.bias"] = db2
        grads[pre + ".gn.weight"], grads[pre + ".gn.bias"] = dgw, dgb
        return d_y

    # ------------------------------------------------------------------ SGPBlock
    def block_fwd(self, x, o):
        ctx = SimpleNamespace(x=x)
        ctx.ln = ops.layernorm(x, o.ln_w, o.ln_b)
        ctx.y = ops.sgp_branch(ctx.ln, x, o.ks, o.up, o.dw, o.db)
        return self._mlp_fwd(ctx.y, o, ctx), ctx

    def block_bwd(self, dout, o, ctx, grads):
        d_y = self._mlp_bwd(dout, ctx.y, o, ctx, grads)                   # y = x + branch(LN(x))
        d_o, ddw, ddb = B_.sgp_branch_bwd(ctx.ln, d_y, o.ks, o.up, o.dw, o.db)
        _, dlw, dlb = B_.layernorm_bwd(ctx.x, d_o, o.ln_w, dx=d_y, accumulate=True)      # d_y becomes d_x
        sd, pre = self.sd, o.pre
        _unpack_branch(grads, pre, _BR, ddw, ddb, sd)
        grads[pre + ".ln.weight"] = dlw.reshape(sd[pre + ".ln.weight"].shape)
        grads[pre + ".ln.bias"] = dlb.reshape(sd[pre + ".ln.bias"].shape)
        return d_y

    # ------------------------------------------------------------------ SGPMixer
    def mixer_fwd(self, xlo, z, o):
        Bn, T_hi, C = z.shape
        T_lo = xlo.shape[1]
        R = Bn * T_hi
        ctx = SimpleNamespace(xlo=xlo, z=z)
        cat = torch.empty((Bn, T_hi, 6 * C), dtype=self.dt, device=z.device)
        ops.layernorm(z, o.ln1_w, o.ln1_b, out=cat.view(-1)[4 * C:], ldy=6 * C, rows=R, C=C)
        xn = ops.layernorm(xlo, o.ln2_w, o.ln2_b)
        ops.mixer_branch(xn, cat, T_hi, o.ks, o.up, o.dw1, o.db1, o.dw2, o.db2)
        cpre = self._long_k(cat, o.cat.w, o.bcat, R).view(Bn, T_hi, C)
        mo = B_.eltwise(cpre, None, B_.GELU_FWD)
        ctx.cat, ctx.cpre, ctx.mo, ctx.T_lo = cat, cpre, mo, T_lo
        return self._mlp_fwd(mo, o, ctx), ctx

    def mixer_bwd(self, dout, o, ctx, grads):
        Bn, T_hi, C = ctx.z.shape
        R = Bn * T_hi
        d_mo = self._mlp_bwd(dout, ctx.mo, o, ctx, grads)
        d_cpre = B_.eltwise(ctx.cpre, d_mo, B_.GELU_BWD)
        d_cat = ops.gemm(d_cpre, o.cat.wt, None, None, ops.ACT_NONE, M=R).view(Bn, T_hi, 6 * C)
        dWc, dbc = B_.wgrad(d_cpre, ctx.cat, M=R)
        flat_cat, flat_d = ctx.cat.view(-1), d_cat.view(-1)
        slab = lambda buf, i: buf[i * C:]                                   # noqa: E731  (row stride stays 6C)
        d_zn, ddw1, ddb1 = B_.sgp_branch_bwd(slab(flat_cat, 4), (slab(flat_d, 0), slab(flat_d, 2), slab(flat_d, 4)), o.ks,
                                             o.up, o.dw1, o.db1, ldo=6 * C, ldg=6 * C, B=Bn, T=T_hi, C=C)
        d_xu, ddw2, ddb2 = B_.sgp_branch_bwd(slab(flat_cat, 5), (slab(flat_d, 1), slab(flat_d, 3), slab(flat_d, 5)), o.ks,
                                             o.up, o.dw2, o.db2, ldo=6 * C, ldg=6 * C, B=Bn, T=T_hi, C=C)
        d_xn = B_.upsample_bwd(d_xu, ctx.T_lo)
        d_z, dl1w, dl1b = B_.layernorm_bwd(ctx.z, d_zn, o.ln1_w)
        d_xlo, dl2w, dl2b = B_.layernorm_bwd(ctx.xlo, d_xn, o.ln2_w)
        sd, pre = self.sd, o.pre
        _unpack_branch(grads, pre, [n + "1" for n in _BR], ddw1, ddb1, sd)
        _unpack_branch(grads, pre, [n + "2" for n in _BR], ddw2, ddb2, sd)
        grads[pre + ".concat_fc.weight"] = dWc.reshape(sd[pre + ".concat_fc.weight"].shape)
        grads[pre + ".concat_fc.bias"] = dbc
        for k, (w_, b_) in (("ln1", (dl1w, dl1b)), ("ln2", (dl2w, dl2b))):
            grads[f"{pre}.{k}.weight"] = w_.reshape(sd[f"{pre}.{k}.weight"].shape)
            grads[f"{pre}.{k}.bias"] = b_.reshape(sd[f"{pre}.{k}.bias"].shape)
        return d_xlo, d_z

    # ------------------------------------------------------------------ the pyramid (modules.py:69-87)
    def pyramid_fwd(self, feat):
        n, lens = self.n, pyramid_lengths(feat.shape[1], self.n)
        tape = []
        cur, stash = feat, []
        for i in range(n):
            cur, c = self.block_fwd(cur, self.blocks[i])
            tape.append(("block", i, c))
            stash.append(cur)
            pooled = ops.maxpool(cur, lens[i + 1])
            tape.append(("pool", i, SimpleNamespace(x=cur)))
            cur = pooled
        cur, c = self.block_fwd(cur, self.blocks[n])
        tape.append(("block", n, c))
        for i in range(n):
            lvl = n - 1 - i
            cur, c = self.mixer_fwd(cur, stash[lvl], self.mixers[lvl])
            tape.append(("mixer", lvl, c))
            cur, c = self.block_fwd(cur, self.blocks[n + 1 + i])
            tape.append(("block", n + 1 + i, c))
        return cur, tape

    def pyramid_bwd(self, dout, tape, grads):
        d_cur = dout
        d_stash = {}
        for kind, idx, c in reversed(tape):
            if kind == "block":
                if idx < self.n and idx in d_stash:                       # encoder block: its output also fed a mixer
                    d_cur = B_.eltwise(d_cur, d_stash.pop(idx), B_.ADD)
                d_cur = self.block_bwd(d_cur, self.blocks[idx], c, grads)
            elif kind == "mixer":
                d_cur, d_z = self.mixer_bwd(d_cur, self.mixers[idx], c, grads)
                d_stash[idx] = d_z
            else:
                d_cur = B_.maxpool_bwd(c.x, d_cur)
        return d_cur

    # ------------------------------------------------------------------ heads + loss
    def _head_names(self):
        """FCLayers of the model in output-column order (each has its own Dropout, modules.py:366-387)."""
        sd = self.sd
        names = (["_pred_fine._fc1._fc_out", "_pred_fine._fc2._fc_out"] if "_pred_fine._fc1._fc_out.weight" in sd
                 else ["_pred_fine._fc_out"])
        if self.radi > 0:
            names.append("_pred_displ._fc_out")
        return names

    def forward_heads(self, feat, drop_masks=None):
        """Train-mode forward of the temporal stack: SGP pyramid, dropout (the given keep-masks), heads.
        Returns (head_out (B*T, n_out) fp32, ctx for `backward_heads`)."""
        enc, tape = self.pyramid_fwd(feat)
        names = self._head_names()
        sd = self.sd
        ws, bs = [sd[n + ".weight"] for n in names], [sd[n + ".bias"] for n in names]
        xs = [enc if drop_masks is None else B_.eltwise(enc, drop_masks[i], B_.MUL) for i in range(len(ws))]
        outs = [ops.heads(xs[i], ws[i], bs[i]) for i in range(len(ws))]
        head_out = outs[0] if len(outs) == 1 else torch.cat(outs, dim=1).contiguous()
        ctx = SimpleNamespace(tape=tape, names=names, ws=ws, xs=xs, drop_masks=drop_masks, shape=feat.shape)
        return head_out, ctx

    def head_layout(self):
        """(n_cls, displ_col, double) of the head_out columns."""
        names = self._head_names()
        ws = [self.sd[n + ".weight"] for n in names]
        double = len(names) - (1 if self.radi > 0 else 0) == 2
        n_cls = sum(w_.shape[0] for w_ in ws) - (1 if self.radi > 0 else 0)
        return n_cls, (n_cls if self.radi > 0 else -1), double

    def loss_fwd_bwd(self, head_out, Bn, T, label, labelD=None, soft=None, fg_weight=5.0, grad_scale=1.0, dataset=None):
        """Loss of model.py:308-319 (double head: 278-306) and its gradient w.r.t. head_out."""
        n_cls, dcol, double = self.head_layout()
        ld = labelD if self.radi > 0 else None
        dev = head_out.device
        if double:
            names = self._head_names()
            K1a, K1b = self.sd[names[0] + ".weight"].shape[0], self.sd[names[1] + ".weight"].shape[0]
            key = ("2h", fg_weight, max(K1a, K1b))
            if key not in self._cls_w:
                self._cls_w[key] = torch.tensor([1.0] + [float(fg_weight)] * (max(K1a, K1b) - 1), dtype=torch.float32,
                                                device=dev)
            if soft is not None:
                # mixup with the joint-dataset head (model.py:278-306 with 3-D labels): soft labels over the clip's own
                # head slice; labels were built over the concatenated (K1a + K1b) columns
                return ops.loss2_soft(head_out, Bn, T, K1a, K1b, dataset, soft, self._cls_w[key], displ_col=dcol,
                                      labelD=ld, grad_scale=grad_scale)
            return ops.loss2(head_out, Bn, T, K1a, K1b, dataset, label, self._cls_w[key], displ_col=dcol, labelD=ld,
                             want_grad=True, grad_scale=grad_scale)
        K1 = self.K1
        if fg_weight not in self._cls_w:        # built once per weight: no host->device copy inside a captured step
            self._cls_w[fg_weight] = torch.tensor([1.0] + [float(fg_weight)] * (K1 - 1), dtype=torch.float32, device=dev)
        cls_w = self._cls_w[fg_weight]
        loss = ops.loss(head_out, K1, cls_w, hard=label, soft=soft, displ_col=dcol, labelD=ld)
        dhead = ops.loss_bwd(head_out, K1, cls_w, hard=label, soft=soft, displ_col=dcol, labelD=ld, grad_scale=grad_scale)
        return loss, dhead

    def backward_heads(self, ctx, dhead, grads):
        """Backward of forward_heads: fills grads for every head / SGP parameter, returns d feat."""
        Bn, T, C = ctx.shape
        d_enc = None
        col = 0
        for i, nm in enumerate(ctx.names):
            n_out = ctx.ws[i].shape[0]
            dpart = dhead[:, col:col + n_out].contiguous()
            col += n_out
            dx, dw, db = ops.heads_bwd(dpart, ctx.xs[i], ctx.ws[i])
            if ctx.drop_masks is not None:
                dx = B_.eltwise(dx, ctx.drop_masks[i], B_.MUL)
            d_enc = dx if d_enc is None else B_.eltwise(d_enc, dx, B_.ADD)
            grads[nm + ".weight"], grads[nm + ".bias"] = dw, db
        return self.pyramid_bwd(d_enc.view(Bn, T, C), ctx.tape, grads)

    def loss_and_grads(self, feat, label, labelD=None, soft=None, drop_masks=None, fg_weight=5.0, grad_scale=1.0,
                       dataset=None):
        """feat (B,T,C) in the activation dtype; label int64 (B*T,) or soft (B*T,K1) fp32; labelD fp32 (B*T,).
        drop_masks: optional [(B,T,C) keep-mask scaled by 1/(1-p)] per FCLayers (class head(s) first, displacement last),
        activation dtype (train-mode dropout).  dataset: int64 (B,) in {1,2} when the model has the joint-dataset double
        head (labels of dataset 2 already shifted, update_labels_2heads).
        Returns (loss scalar tensor [total, ce, mse], grads dict, d_feat)."""
        Bn, T, C = feat.shape
        head_out, ctx = self.forward_heads(feat, drop_masks)
        loss, dhead = self.loss_fwd_bwd(head_out, Bn, T, label, labelD, soft, fg_weight, grad_scale, dataset)
        grads = {}
        d_feat = self.backward_heads(ctx, dhead, grads)
        return loss, grads, d_feat
