"""Torch-tensor wrappers of the backward entry points (training path).  Same conventions as ops.py: tensors are
device memory handed over as raw pointers on the current stream; scratch (`part*`) is allocated here when the caller
does not pass it (a training engine passes preallocated buffers)."""
import os

import ctypes

import torch

from . import _lib
from ._lib import call, ptr, stream_ptr, dtype_code

GELU_FWD, GELU_BWD, ADD, MUL = 0, 1, 2, 3


def _f32(shape, dev):
    return torch.empty(shape, dtype=torch.float32, device=dev)


def eltwise(x, dy, mode, out=None):
    if out is None:
        out = torch.empty_like(x)
    call("tdeed_eltwise", ptr(x), ptr(dy), ptr(out), x.numel(), mode, dtype_code(x.dtype), stream_ptr())
    return out


def transpose(x, out=None):
    R, Cc = x.shape
    if out is None:
        out = torch.empty((Cc, R), dtype=x.dtype, device=x.device)
    call("tdeed_transpose", ptr(x), R, Cc, ptr(out), dtype_code(x.dtype), stream_ptr())
    return out


class LazyFold:
    """A parameter gradient still in per-workgroup partials [P][n] (fp32, contiguous): the gradient write-out launch
    (`multi_copy` -> tdeed_multi_fold) folds it on its way into the flat gradient buffer, instead of one fold launch per
    tensor (~280 per step).  Quacks like the tensor it stands for as far as the write-out needs (numel / dtype / reshape)."""

    def __init__(self, part, P, n, shape=None, pstride=None):
        """part: fp32 tensor whose storage from data_ptr() on holds P rows of n values, pstride (default n) floats apart"""
        self.part, self.P, self.n = part, int(P), int(n)
        self.pstride = self.n if pstride is None else int(pstride)
        self.shape = tuple(shape) if shape is not None else (self.n,)
        self.dtype, self.device = torch.float32, part.device

    def numel(self):
        return self.n

    def reshape(self, *shape):
        shape = tuple(shape[0]) if len(shape) == 1 and isinstance(shape[0], (tuple, list, torch.Size)) else tuple(shape)
        return LazyFold(self.part, self.P, self.n, shape, self.pstride)

    view = reshape

    def materialize(self):
        if self.pstride != self.n:
            rows = torch.as_strided(self.part, (self.P, self.n), (self.pstride, 1), self.part.storage_offset())
            return rows.sum(0).view(self.shape)
        out = _f32((self.n,), self.device)
        call("tdeed_reduce_partials", ptr(self.part), self.P, self.n, ptr(out), 0, stream_ptr())
        return out.view(self.shape)


LAZY_WGRAD = False          # set by TrainEngine.backward_and_write: wgrad() then returns LazyFold objects


def materialize(g):
    return g.materialize() if isinstance(g, LazyFold) else g


def wgrad_splice_ok(dtype, M):
    """whether wgrad(X0=..., k0=...) is served (bf16, long contractions: the transposing-read kernel)"""
    return dtype == torch.bfloat16 and M >= 4096


def wgrad(dY, X, with_bias=True, dW=None, db=None, accumulate=False, M=None, X0=None, k0=0):
    """dW[n][k] = sum_m dY[m][n] X[m][k]; db[n] = sum_m dY[m][n].  dY (M,N), X (M,K) in the activation dtype.
    X0 (M,k0): columns k < k0 of the X operand come from X0 (the gate-shift splice, never materialised).
    With LAZY_WGRAD set (and no caller-provided outputs) the results are LazyFold objects."""
    N, K = dY.shape[-1], X.shape[-1]
    x0a = (ptr(X0), (X0.shape[-1] if X0 is not None else 0), int(k0) if X0 is not None else 0)
    M = dY.numel() // N if M is None else M
    Z = _lib.load().tdeed_wgrad_slices(M, N, K)
    dev = dY.device
    if LAZY_WGRAD and dW is None and db is None and not accumulate:
        pw, pb = _f32((Z, N, K), dev), (_f32((Z, N), dev) if with_bias else None)
        call("tdeed_wgrad", ptr(dY), N, ptr(X), K, *x0a, M, N, K, ptr(pw), ptr(pb), None, None, -1, dtype_code(dY.dtype),
             stream_ptr())
        return LazyFold(pw, Z, N * K, (N, K)), (LazyFold(pb, Z, N, (N,)) if with_bias else None)
    dW = _f32((N, K), dev) if dW is None else dW
    if with_bias and db is None:
        db = _f32((N,), dev)
    pw, pb = _f32((Z, N, K), dev), (_f32((Z, N), dev) if with_bias else None)
    call("tdeed_wgrad", ptr(dY), N, ptr(X), K, *x0a, M, N, K, ptr(pw), ptr(pb), ptr(dW), ptr(db if with_bias else None),
         int(accumulate), dtype_code(dY.dtype), stream_ptr())
    return dW, db


def layernorm_bwd(x, dy, w, eps=1e-5, dx=None, accumulate=False, ldx=None, ldy=None, rows=None, C=None):
    C = x.shape[-1] if C is None else C
    rows = x.numel() // x.shape[-1] if rows is None else rows
    dev = x.device
    if dx is None:
        dx = torch.empty_like(x)
    nb = _lib.load().tdeed_layernorm_bwd_blocks(rows)
    part = _f32((nb, 2, C), dev)
    if LAZY_WGRAD:
        # the parameter gradients stay per-workgroup partials: the gradient write-out folds them (no reduce launches here)
        call("tdeed_layernorm_bwd", ptr(x), (C if ldx is None else ldx), ptr(dy), (C if ldy is None else ldy), rows, C, ptr(w),
             eps, ptr(dx), int(accumulate), ptr(part), None, None, dtype_code(x.dtype), stream_ptr())
        flat = part.view(-1)
        return dx, LazyFold(flat, nb, C, pstride=2 * C), LazyFold(flat[C:], nb, C, pstride=2 * C)
    dw, db = _f32((C,), dev), _f32((C,), dev)
    call("tdeed_layernorm_bwd", ptr(x), (C if ldx is None else ldx), ptr(dy), (C if ldy is None else ldy), rows, C, ptr(w),
         eps, ptr(dx), int(accumulate), ptr(part), ptr(dw), ptr(db), dtype_code(x.dtype), stream_ptr())
    return dx, dw, db


def groupnorm_bwd(x, dy, G, w, eps=1e-5, dx=None, accumulate=False):
    B, T, C = x.shape
    dev = x.device
    if dx is None:
        dx = torch.empty_like(x)
    part = _f32((B, 2, C), dev)
    if LAZY_WGRAD:
        call("tdeed_groupnorm_bwd", ptr(x), ptr(dy), B, T, C, G, ptr(w), eps, ptr(dx), int(accumulate), ptr(part), None, None,
             dtype_code(x.dtype), stream_ptr())
        flat = part.view(-1)
        return dx, LazyFold(flat, B, C, pstride=2 * C), LazyFold(flat[C:], B, C, pstride=2 * C)
    dw, db = _f32((C,), dev), _f32((C,), dev)
    call("tdeed_groupnorm_bwd", ptr(x), ptr(dy), B, T, C, G, ptr(w), eps, ptr(dx), int(accumulate), ptr(part), ptr(dw),
         ptr(db), dtype_code(x.dtype), stream_ptr())
    return dx, dw, db


def sgp_branch_bwd(o, g, ks, up, dw, db, d_o=None, ldo=None, ldg=None, ld_do=None, B=None, T=None, C=None):
    """-> d_o (B,T,C), d_dw (C, 2ks+up+2), d_db (5, C) in the packed layouts of ops.sgp_branch.
    g: one tensor (SGPBlock) or a tuple (g_conv, g_inst, g_id) of views with a common row stride ldg (SGPMixer)."""
    gs = g if isinstance(g, (tuple, list)) else (g, g, g)
    if B is None:
        B, T, C = o.shape
    dev = o.device
    wlen = 2 * ks + up + 2
    if d_o is None:
        d_o = torch.empty((B, T, C), dtype=o.dtype, device=dev)
    pw, pb, ddw, ddb = _f32((B, C, wlen), dev), _f32((B, 5, C), dev), _f32((C, wlen), dev), _f32((5, C), dev)
    call("tdeed_sgp_branch_bwd", ptr(o), (C if ldo is None else ldo), ptr(gs[0]), ptr(gs[1]), ptr(gs[2]),
         (C if ldg is None else ldg), B, T, C, ks, up, ptr(dw), ptr(db), ptr(d_o), (C if ld_do is None else ld_do), ptr(pw),
         ptr(pb), ptr(ddw), ptr(ddb), dtype_code(o.dtype), stream_ptr())
    return d_o, ddw, ddb


def upsample_bwd(d_xu, T_lo, ld=None, B=None, T_hi=None, C=None):
    if B is None:
        B, T_hi, C = d_xu.shape
    out = torch.empty((B, T_lo, C), dtype=d_xu.dtype, device=d_xu.device)
    call("tdeed_upsample_bwd", ptr(d_xu), (C if ld is None else ld), B, T_hi, T_lo, C, ptr(out), dtype_code(d_xu.dtype),
         stream_ptr())
    return out


def maxpool_bwd(x, dy):
    B, T_in, C = x.shape
    dx = torch.empty_like(x)
    call("tdeed_maxpool_bwd", ptr(x), ptr(dy), B, T_in, dy.shape[1], C, ptr(dx), dtype_code(x.dtype), stream_ptr())
    return dx


# ----------------------------------------------------------------------------- train-mode trunk pieces
def bn_train(z, w, b, eps=1e-5, momentum=0.1, run_mean=None, run_var=None, res=None, relu=True, out=None):
    """BatchNorm with batch statistics over the rows of z (.., C) + optional residual + ReLU.
    Returns (y, ctx) where ctx = (mean, rstd) for the backward."""
    C = z.shape[-1]
    M = z.numel() // C
    dev = z.device
    part = _f32((_lib.load().tdeed_bn_slabs(M), 2, C), dev)
    mean, rstd, a, bb = (_f32((C,), dev) for _ in range(4))
    call("tdeed_bn_train_stats", ptr(z), M, C, ptr(w), ptr(b), eps, momentum, ptr(part), ptr(mean), ptr(rstd), ptr(a),
         ptr(bb), ptr(run_mean), ptr(run_var), dtype_code(z.dtype), stream_ptr())
    if out is None:
        out = torch.empty_like(z)
    call("tdeed_bn_apply", ptr(z), M, C, ptr(a), ptr(bb), ptr(res), int(relu), ptr(out), dtype_code(z.dtype), stream_ptr())
    return out, (mean, rstd, a, bb)


def bn_apply(z, a, b, res=None, relu=True, out=None):
    """y = act(z * a[c] + b[c] + res): the apply pass of a BatchNorm whose affine is known (tdeed_bn_apply)."""
    C = z.shape[-1]
    M = z.numel() // C
    if out is None:
        out = torch.empty_like(z)
    call("tdeed_bn_apply", ptr(z), M, C, ptr(a), ptr(b), ptr(res), int(relu), ptr(out), dtype_code(z.dtype), stream_ptr())
    return out


def bn_finalize_apply(z, part_s, part_q, pstride, P, w, b, eps=1e-5, momentum=0.1, run_mean=None, run_var=None, res=None,
                      relu=True, out=None, apply=True, res_affine=None, slice_out=None):
    """BatchNorm(batch statistics) of z from per-channel partial sums a producer's epilogue wrote (gemm(colpart=...),
    gconv3x3(pooled_sq=...)) + the apply pass.  Returns (y, (mean, rstd, a, b))."""
    C = z.shape[-1]
    M = z.numel() // C
    dev = z.device
    mean, rstd, a, bb = (_f32((C,), dev) for _ in range(4))
    if P > 2048:                                 # one partial row per (frame, band): fold to 64 rows first, chip-wide
        tmp = _f32((64, 2, C), dev)
        if pstride == 2 * C and part_q.data_ptr() == part_s.data_ptr() + 4 * C:
            # a contraction's [P][2][C] partials: sums and sums of squares are one row of 2C floats, folded by one launch
            call("tdeed_fold_rows", ptr(part_s), pstride, P, 2 * C, 64, ptr(tmp), 2 * C, stream_ptr())
        else:
            call("tdeed_fold_rows", ptr(part_s), pstride, P, C, 64, ptr(tmp), 2 * C, stream_ptr())
            call("tdeed_fold_rows", ptr(part_q), pstride, P, C, 64, ptr(tmp.view(-1)[C:]), 2 * C, stream_ptr())
        part_s, part_q, pstride, P = tmp.view(-1), tmp.view(-1)[C:], 2 * C, 64
    call("tdeed_bn_finalize", ptr(part_s), ptr(part_q), pstride, P, M, C, ptr(w), ptr(b), eps, momentum, ptr(mean),
         ptr(rstd), ptr(a), ptr(bb), ptr(run_mean), ptr(run_var), stream_ptr())
    if not apply:                                # the consumers apply relu(a*z + b) in their own loads: no post-BN map
        return None, (mean, rstd, a, bb)
    if out is None:
        out = torch.empty_like(z)
    if slice_out is not None:                   # (F, xs): also the compact gate-shift slice of the next block
        F2, xs = slice_out
        ra, rb = res_affine if res_affine is not None else (None, None)
        call("tdeed_bn_apply_slice", ptr(z), M, C, ptr(a), ptr(bb), ptr(res), ptr(ra), ptr(rb), int(relu), ptr(out), ptr(xs), F2,
             xs.shape[-1], dtype_code(z.dtype), stream_ptr())
    elif res_affine is not None:                # res is a raw conv output under its own BatchNorm affine (shortcut conv)
        call("tdeed_bn_apply2", ptr(z), M, C, ptr(a), ptr(bb), ptr(res), ptr(res_affine[0]), ptr(res_affine[1]), int(relu),
             ptr(out), dtype_code(z.dtype), stream_ptr())
    else:
        call("tdeed_bn_apply", ptr(z), M, C, ptr(a), ptr(bb), ptr(res), int(relu), ptr(out), dtype_code(z.dtype), stream_ptr())
    return out, (mean, rstd, a, bb)


def bn_train_bwd(z, dy, y, ctx, w, relu=True, want_res=False):
    """-> dz, d_res (or None), dw, db.  ctx = (mean, rstd[, a, b]): with the forward affine a, b in ctx and y None the ReLU
    mask is recomputed from z (no residual in front of the ReLU)."""
    C = z.shape[-1]
    M = z.numel() // C
    dev = z.device
    mean, rstd = ctx[0], ctx[1]
    fa, fb = (ctx[2], ctx[3]) if len(ctx) >= 4 else (None, None)
    if relu and y is None and fa is None:
        raise ValueError("bn_train_bwd: the ReLU mask needs y or the forward affine in ctx")
    part, sums = _f32((_lib.load().tdeed_bn_slabs(M), 2, C), dev), _f32((2, C), dev)
    dz = torch.empty_like(z)
    d_res = torch.empty_like(z) if want_res else None
    call("tdeed_bn_train_bwd", ptr(z), ptr(dy), ptr(y if relu else None), int(relu), M, C, ptr(mean), ptr(rstd), ptr(w),
         ptr(fa), ptr(fb), ptr(part), ptr(sums), ptr(dz), ptr(d_res), None, None, dtype_code(z.dtype), stream_ptr())
    return dz, d_res, sums[1], sums[0]              # dw = sum g * xhat, db = sum g: views of the folded sums, no copies


def pool_rows(x, x2=None, affine=None, affine_on=1):
    """x (N,h,w,C): mean over pixels (x2 None) or sum over pixels of x*x2 -> (N,C) fp32.  affine = (a, b) fp32 [C]:
    operand `affine_on` (1: x, 2: x2) is a raw conv output and relu(a*. + b) is applied on load."""
    N, C = x.shape[0], x.shape[-1]
    hw = x.numel() // (N * C)
    p = _f32((N, C), x.device)
    call("tdeed_pool_rows", ptr(x), ptr(x2), N, hw, C, ptr(affine[0] if affine else None), ptr(affine[1] if affine else None),
         int(affine_on) if affine else 0, ptr(p), dtype_code(x.dtype), stream_ptr())
    return p


def se_train_fwd(p, w1t, b1, w2t, b2):
    N, C = p.shape
    R = w1t.shape[1]
    hid, gate = _f32((N, R), p.device), _f32((N, C), p.device)
    call("tdeed_se_train_fwd", ptr(p), N, C, R, ptr(w1t), ptr(b1), ptr(w2t), ptr(b2), ptr(hid), ptr(gate), stream_ptr())
    return hid, gate


def se_train_bwd(d_gate, gate, hid, w1, w2):
    N, C = gate.shape
    R = hid.shape[1]
    d_pre2, d_hid, d_p = _f32((N, C), gate.device), _f32((N, R), gate.device), _f32((N, C), gate.device)
    call("tdeed_se_train_bwd", ptr(d_gate), ptr(gate), ptr(hid), N, C, R, ptr(w1), ptr(w2), ptr(d_pre2), ptr(d_hid),
         ptr(d_p), stream_ptr())
    return d_pre2, d_hid, d_p


def scale_rows(x, s, add=None, add_scale=1.0, out=None, affine=None):
    """out = x' * s[n] + add[n] * add_scale; x' = x, or relu(a*x + b) with affine = (a, b) (x a raw conv output)"""
    N, C = x.shape[0], x.shape[-1]
    hw = x.numel() // (N * C)
    if out is None:
        out = torch.empty_like(x)
    call("tdeed_scale_rows", ptr(x), ptr(s), ptr(add), float(add_scale), N, hw, C, ptr(affine[0] if affine else None),
         ptr(affine[1] if affine else None), ptr(out), dtype_code(x.dtype), stream_ptr())
    return out


def se_bn_bwd(d, z, bn, w, gate, hid, se_w1, se_w2):
    """The passes between conv3's input gradient d = d(y2 * gate) (N,h,w,C) and conv2's BatchNorm input gradient without
    the d_y2 map (csrc/trunk_bwd2.hip): z = conv2's raw output, bn = (mean, rstd, a, b) of its BatchNorm (y2 = relu(a z + b)),
    w its weight, gate (N,C) / hid (N,R) the SE forward's values, se_w1 (R,C) / se_w2 (C,R) the SE weights.
    -> dz (like z), dw, db (BatchNorm weight / bias gradients), d_pre2 (N,C), d_hid (N,R) (operands of the SE weight gradients)."""
    N, C = z.shape[0], z.shape[-1]
    hw = z.numel() // (N * C)
    dev = z.device
    mean, rstd, fa, fb = bn
    sums = _f32((5, N, C), dev)
    call("tdeed_se_bn_bwd_sums", ptr(d), ptr(z), N, hw, C, ptr(fa), ptr(fb), ptr(mean), ptr(sums), dtype_code(z.dtype),
         stream_ptr())
    d_pre2, d_hid, d_p = se_train_bwd(sums[0], gate, hid, se_w1, se_w2)
    st = _f32((2, C), dev)
    call("tdeed_se_bn_bwd_finalize", ptr(sums), ptr(gate), ptr(d_p), N, hw, C, ptr(rstd), ptr(st), stream_ptr())
    dz = torch.empty_like(z)
    call("tdeed_se_bn_bwd_apply", ptr(d), ptr(z), ptr(gate), ptr(d_p), N, hw, C, ptr(fa), ptr(fb), ptr(mean), ptr(rstd),
         ptr(w), ptr(st), ptr(dz), dtype_code(z.dtype), stream_ptr())
    return dz, st[1], st[0], d_pre2, d_hid


def gconv3x3_dgrad_stats(dy, wfrag_t, one, zero, gw, z1, bn1):
    """Stride-1 input gradient of conv2 (MFMA kernel, flipped / transposed weights) that also leaves the column sums of
    conv1's BatchNorm backward (tdeed_gconv3x3_dgrad_stats).  z1: conv1's raw output, bn1 = (mean, rstd, a, b).
    -> (d_y1, (part_s, part_q, row stride, rows))"""
    from . import ops
    N, Hi, Wi, C = dy.shape
    parts = ops.gconv3x3_parts(Hi, Wi, C, 1, dy.dtype)
    dx = torch.empty_like(dy)
    ps, pq = _f32((N, parts, C), dy.device), _f32((N, parts, C), dy.device)
    call("tdeed_gconv3x3_dgrad_stats", ptr(dy), N, Hi, Wi, C, gw, ptr(wfrag_t), ptr(one), ptr(zero), ptr(dx), ptr(z1), ptr(bn1[2]),
         ptr(bn1[3]), ptr(bn1[0]), ptr(ps), ptr(pq), stream_ptr())
    return dx, (ps, pq, C, N * parts)


def gconv3x3_bwd_stats_fits(N, Hi, Wi, C, gw):
    return _lib.load().tdeed_gconv3x3_bwd_stats_fits(N, Hi, Wi, C, gw) != 0


def gconv3x3_bwd_stats(x, dy, w_packed, gw, z1, bn1, in_affine=None):
    """gconv3x3_bwd (stride 2, bf16) whose input-gradient launch also leaves the column sums of conv1's BatchNorm backward
    (tdeed_gconv3x3_bwd_stats).  -> dx, dw, (part_s, part_q, row stride, rows)"""
    N, Hi, Wi, C = x.shape
    Ho, Wo = dy.shape[1], dy.shape[2]
    G = C // gw
    part = _f32((_lib.load().tdeed_gconv_wgrad_slabs(N * Ho * Wo), G * 9 * gw * gw), x.device)
    dx, dw = torch.empty_like(x), _f32((G, 9, gw, gw), x.device)
    nb = _lib.load().tdeed_gconv3x3_bwd_stats_bands(Hi)
    ps, pq = _f32((N * nb, C), x.device), _f32((N * nb, C), x.device)
    call("tdeed_gconv3x3_bwd_stats", ptr(x), ptr(dy), N, Hi, Wi, C, gw, ptr(w_packed), ptr(in_affine[0] if in_affine else None),
         ptr(in_affine[1] if in_affine else None), ptr(dx), ptr(part), ptr(dw), ptr(z1), ptr(bn1[2]), ptr(bn1[3]), ptr(bn1[0]),
         ptr(ps), ptr(pq), stream_ptr())
    return dx, dw, (ps, pq, C, N * nb)


def bn_bwd_masked_from_parts(z, dy, ctx, w, part):
    """BatchNorm + ReLU backward of z whose masked column sums the producer of dy left in `part` = (part_s, part_q, row
    stride, rows).  -> dz, dw, db"""
    C = z.shape[-1]
    M = z.numel() // C
    sums = _f32((2, C), z.device)
    dz = torch.empty_like(z)
    ps, pq, stride, P = part
    call("tdeed_bn_bwd_masked_from_parts", ptr(z), ptr(dy), M, C, ptr(ctx[0]), ptr(ctx[1]), ptr(w), ptr(ctx[2]), ptr(ctx[3]),
         ptr(ps), ptr(pq), stride, P, ptr(sums), ptr(dz), dtype_code(z.dtype), stream_ptr())
    return dz, sums[1], sums[0]


def narrow_conv1_bwd_fits(Co, Ci, dtype):
    return dtype == torch.bfloat16 and _lib.load().tdeed_narrow_conv1_bwd_fits(Co, Ci) != 0


NARROW_RECOMPUTE = True


def narrow_conv1_bwd(d_y1, z1, bn1, w, part, x, wt, sink=None, residual=None, r_hw=None, sums=None, recompute=False):
    """conv1 backward of a narrow bottleneck in one launch (tdeed_narrow_conv1_bwd): d_y1 / z1 (.., Co), bn1 = (mean, rstd, a, b),
    w its weight, part = the masked column-sum partials conv2's input-gradient launch left, x (.., Ci) conv1's input (= the
    sink's mask), wt (Ci, Co) the transposed weight, residual: shortcut gradient (r_hw = (hi, wi): on the even pixels only).
    recompute: z1 IS x @ wt (the raw conv output of exactly these operands): the launch recomputes it from the x tile it holds
    instead of reading the map (not at 128 <- 128).
    -> dx (M, Ci) (masked / summed for `sink`), dW (Co, Ci) as a LazyFold, BatchNorm dw, db."""
    Co, Ci = z1.shape[-1], x.shape[-1]
    rc = recompute and NARROW_RECOMPUTE and not (Co == 128 and Ci == 128)
    M = z1.numel() // Co
    dev = z1.device
    if sums is None:
        ps, pq, stride, P = part
        sums = _f32((2, Co), dev)
        call("tdeed_bn_sums_from_parts", ptr(ps), ptr(pq), stride, P, Co, ptr(bn1[1]), ptr(sums), stream_ptr())
    grid = _lib.load().tdeed_narrow_conv1_bwd_grid(M, Co, Ci)
    dx = torch.empty((M, Ci), dtype=z1.dtype, device=dev)
    wpart = _f32((grid, Co, Ci), dev)
    bpart = None
    if sink is not None:
        bpart = _f32((grid, 3, Ci), dev)
        sink.partA = bpart
    rh, rw = r_hw if r_hw is not None else (0, 0)
    call("tdeed_narrow_conv1_bwd", ptr(d_y1), ptr(None if rc else z1), M, Co, Ci, ptr(bn1[2]), ptr(bn1[3]), ptr(bn1[0]), ptr(bn1[1]), ptr(w),
         ptr(sums), ptr(x), ptr(wt), ptr(residual), (residual.shape[-1] if residual is not None else 0), rh, rw, ptr(dx),
         int(sink is not None), ptr(sink.z if sink else None), ptr(sink.mean if sink else None), ptr(sink.zd if sink else None),
         ptr(sink.mean_d if sink else None), ptr(bpart), ptr(wpart), stream_ptr())
    dW = LazyFold(wpart, grid, Co * Ci, (Co, Ci)) if LAZY_WGRAD else LazyFold(wpart, grid, Co * Ci, (Co, Ci)).materialize()
    return dx, dW, sums[1], sums[0]


class GradSink:
    """The ReLU backward at a block's output and the statistics of the BatchNorm backward behind it, delegated to whichever
    kernels produce the gradient arriving there (csrc/trunk_bwd2.hip "gradient sink"): mask = the block's output (ReLU
    backward: out > 0), z / mean = raw output and batch mean of its conv3 BatchNorm, zd / mean_d the same for its shortcut
    BatchNorm (blocks with a downsample conv).  The producers leave their partial column sums in partA (tdeed_gemm_dgrad:
    one row per 128-row tile) and partB (the gate-shift module's input gradient joining columns [0, nB))."""

    def __init__(self, mask, z, mean, zd=None, mean_d=None):
        C = z.shape[-1]
        self.mask, self.z, self.mean, self.zd, self.mean_d, self.C = mask.view(-1, C), z.view(-1, C), mean, zd, mean_d, C
        if zd is not None:
            self.zd = zd.view(-1, C)
        self.partA = self.partB = None
        self.nB = 0


DGRAD_RS = True
DGRAD_RS_MIN_ROWS = 60000


def gemm_dgrad(dz, wt, sink=None, residual=None, r_hw=None, out2=None, wt_ws=None):
    """dx (M, N) = ((dz (M,K) @ wt (N,K)^T) + residual) masked / summed for `sink` (tdeed_gemm_dgrad).  r_hw = (hi, wi): the
    residual has rows for the even pixels of every hi x wi frame only; out2 (M, n2): columns [0, n2) before the residual.
    wt_ws: the same matrix in the register-stationary kernel's fragment order -- K = N = 320 over many rows with a plain
    residual and a one-map sink then run on tdeed_gemm_dgrad_rs."""
    N, K = wt.shape
    M = dz.numel() // K
    dev = dz.device
    dx = torch.empty((M, N), dtype=dz.dtype, device=dev)
    if (DGRAD_RS and wt_ws is not None and sink is not None and sink.zd is None and residual is not None and r_hw is None
            and dz.dtype == torch.bfloat16 and M >= DGRAD_RS_MIN_ROWS and _lib.load().tdeed_gemm_rs_fits(M, K, N)):
        bpart = _f32((_lib.load().tdeed_gemm_rs_grid(M), 3, N), dev)
        sink.partA = bpart
        n2 = out2.shape[-1] if out2 is not None else 0
        call("tdeed_gemm_dgrad_rs", ptr(dz), K, M, K, N, ptr(wt_ws), ptr(residual), residual.shape[-1], ptr(dx), N, ptr(out2),
             n2, n2, ptr(sink.mask), N, ptr(sink.z), N, ptr(sink.mean), ptr(bpart), stream_ptr())
        return dx
    bpart = None
    if sink is not None:
        bpart = _f32(((M + 127) // 128, 3, N), dev)
        sink.partA = bpart
    rh, rw = r_hw if r_hw is not None else (0, 0)
    call("tdeed_gemm_dgrad", ptr(dz), K, M, K, N, ptr(wt), K, ptr(residual), (residual.shape[-1] if residual is not None else 0),
         rh, rw, ptr(dx), N, ptr(out2), (out2.shape[-1] if out2 is not None else 0), (out2.shape[-1] if out2 is not None else 0),
         ptr(sink.mask if sink else None), N, ptr(sink.z if sink else None), N, ptr(sink.mean if sink else None),
         ptr(sink.zd if sink else None), N, ptr(sink.mean_d if sink else None), ptr(bpart), dtype_code(dz.dtype), stream_ptr())
    return dx


def gsf_add_cols_sink(a, b, dx, Fp, sink, bn=None):
    """dx[:, :Fp] += (a + b) masked / summed for `sink` (tdeed_gsf_add_cols_sink).  bn = (xs, sums, mean, rstd, w): b is the
    gradient at the output of the module's BatchNorm3d, whose backward is applied on load (tdeed_gsf_add_cols_sink_bn)."""
    C = dx.shape[-1]
    M = dx.numel() // C
    P = _lib.load().tdeed_gsf_add_cols_sink_parts(M, Fp, dtype_code(dx.dtype))
    part = _f32((P, 3, Fp), dx.device)
    sink.partB, sink.nB = part, Fp
    if bn is not None:
        xs, sums, mean, rstd, w = bn
        call("tdeed_gsf_add_cols_sink_bn", ptr(a), ptr(b), M, C, Fp, ptr(dx), ptr(sink.mask), C, ptr(sink.z), C, ptr(sink.mean),
             ptr(sink.zd), C, ptr(sink.mean_d), ptr(part), ptr(xs), ptr(sums), ptr(mean), ptr(rstd), ptr(w),
             dtype_code(dx.dtype), stream_ptr())
        return dx
    call("tdeed_gsf_add_cols_sink", ptr(a), ptr(b), M, C, Fp, ptr(dx), ptr(sink.mask), C, ptr(sink.z), C, ptr(sink.mean),
         ptr(sink.zd), C, ptr(sink.mean_d), ptr(part), dtype_code(dx.dtype), stream_ptr())
    return dx


def bn_bwd_from_parts(z, g, ctx, w, sink, q=1, want_dz=True, want_sums=False):
    """BatchNorm backward of z for the already masked gradient g whose statistics the producers left in `sink`
    (q = 1: against sink.z, q = 2: against sink.zd).  -> dz (or None), dw, db; with want_sums -> dz, sums fp32 [2][C]
    (row 0 = db, row 1 = dw: the buffer stem_wgrad_bn / narrow_conv1_bwd read as a whole)."""
    C = z.shape[-1]
    M = z.numel() // C
    dev = z.device
    tmp, sums = _f32((2 * 64 * 3 * C,), dev), _f32((2, C), dev)
    dz = torch.empty_like(z) if want_dz else None
    pb = sink.partB
    call("tdeed_bn_bwd_from_parts", ptr(z), ptr(g), M, C, ptr(ctx[0]), ptr(ctx[1]), ptr(w), ptr(sink.partA), sink.partA.shape[0],
         ptr(pb), (pb.shape[0] if pb is not None else 0), sink.nB, q, ptr(tmp), ptr(sums), ptr(dz), dtype_code(z.dtype),
         stream_ptr())
    return (dz, sums[1], sums[0]) if not want_sums else (dz, sums)


def bn_sums_from_sink(z, g, ctx, w, sink, q=1):
    """(sums fp32 [2][C]: row 0 = d bias, row 1 = d weight) of the BatchNorm backward whose column-sum partials lie in
    `sink` (no apply pass)"""
    _, sums = bn_bwd_from_parts(z, g, ctx, w, sink, q=q, want_dz=False, want_sums=True)
    return sums


def gconv3x3_bwd(x, dy, w_packed, gw, stride, want_dx=True, in_affine=None):
    """x (N,Hi,Wi,C), dy (N,Ho,Wo,C) -> dx like x (None if not want_dx), dw fp32 [G][9][gw][gw].
    in_affine = (a, b) (bf16): x is a raw conv output, relu(a*x + b) is applied on load for the weight gradient."""
    N, Hi, Wi, C = x.shape
    Ho, Wo = dy.shape[1], dy.shape[2]
    G = C // gw
    part = _f32((_lib.load().tdeed_gconv_wgrad_slabs(N * Ho * Wo), G * 9 * gw * gw), x.device)
    dx, dw = (torch.empty_like(x) if want_dx else None), _f32((G, 9, gw, gw), x.device)
    call("tdeed_gconv3x3_bwd", ptr(x), ptr(dy), N, Hi, Wi, C, gw, stride, ptr(w_packed),
         ptr(in_affine[0] if in_affine else None), ptr(in_affine[1] if in_affine else None), ptr(dx), ptr(part), ptr(dw),
         dtype_code(x.dtype), stream_ptr())
    return dx, dw


def stride2_gather(x):
    F_, hi, wi, C = x.shape
    out = torch.empty((F_, (hi - 1) // 2 + 1, (wi - 1) // 2 + 1, C), dtype=x.dtype, device=x.device)
    call("tdeed_stride2_rows", ptr(x), ptr(out), F_, hi, wi, C, 0, dtype_code(x.dtype), stream_ptr())
    return out


def stride2_scatter_add(d_small, d_big):
    F_, hi, wi, C = d_big.shape
    call("tdeed_stride2_rows", ptr(d_small), ptr(d_big), F_, hi, wi, C, 1, dtype_code(d_big.dtype), stream_ptr())
    return d_big


def bn_stats(z, w, b, eps=1e-5, momentum=0.1, run_mean=None, run_var=None):
    """Batch statistics of z (.., C) only: -> (mean, rstd, a, b) with y = z*a + b the normalisation."""
    C = z.shape[-1]
    M = z.numel() // C
    dev = z.device
    part = _f32((_lib.load().tdeed_bn_slabs(M), 2, C), dev)
    mean, rstd, a, bb = (_f32((C,), dev) for _ in range(4))
    call("tdeed_bn_train_stats", ptr(z), M, C, ptr(w), ptr(b), eps, momentum, ptr(part), ptr(mean), ptr(rstd), ptr(a),
         ptr(bb), ptr(run_mean), ptr(run_var), dtype_code(z.dtype), stream_ptr())
    return mean, rstd, a, bb


# ----------------------------------------------------------------------------- gate-shift-fuse
def gsf_slice(x, F, Fp):
    C = x.shape[-1]
    M = x.numel() // C
    xs = torch.empty((M, Fp), dtype=x.dtype, device=x.device)
    call("tdeed_gsf_slice", ptr(x), M, C, F, Fp, ptr(xs), dtype_code(x.dtype), stream_ptr())
    return xs


def gsf_bwd_bn_parts(B, T, h, w, C, Fp):
    return _lib.load().tdeed_gsf_bwd_bn_parts(B, T, h, w, C, Fp)


def gsf_bwd(x, gate, fw, ysum, xsum, dA, B, T, F, Fp, w3, sa, sb, cw1, cw2, bn_mean=None):
    """-> d_xs (M,Fp), d_bn (M,Fp), d_w3 (F,27), d_b3 (2,), d_cw (2,18), d_cb (2,).  fw None: the plain gate-shift module
    (_GSM): no fusion conv, d_cw / d_cb come back None.  bn_mean (the BatchNorm3d's batch means): a seventh result, the
    statistics partials fp32 [P][3][Fp] of that BatchNorm's backward (tdeed_gsf_bwd_stats)."""
    N, h, w, C = x.shape
    dev = x.device
    entry, extra = "tdeed_gsf_bwd", ()
    bn_part = None
    if bn_mean is not None:
        bn_part = _f32((gsf_bwd_bn_parts(B, T, h, w, C, Fp), 3, Fp), dev)
        entry, extra = "tdeed_gsf_bwd_stats", (ptr(bn_mean), ptr(bn_part))
    ret = (lambda *r: r + (bn_part,)) if bn_mean is not None else (lambda *r: r)
    scratch = _f32((_lib.load().tdeed_gsf_bwd_scratch_floats(B, T, h * w, F),), dev)
    M = N * h * w
    d_xs = torch.empty((M, Fp), dtype=x.dtype, device=dev)
    d_bn = torch.empty((M, Fp), dtype=x.dtype, device=dev)
    fuse = fw is not None
    if LAZY_WGRAD:
        # parameter gradients stay the launch's per-frame / per-clip partials inside `scratch`; the gradient write-out folds
        # them (six reduce launches per module otherwise).  d_cw / d_cb come back as pairs (channel_conv1, channel_conv2)
        call(entry, ptr(x), ptr(gate), ptr(fw), ptr(ysum), ptr(xsum), ptr(dA), B, T, h, w, C, F, Fp, ptr(w3), ptr(sa),
             ptr(sb), ptr(cw1), ptr(cw2), ptr(scratch), ptr(d_xs), ptr(d_bn), None, None, None, None, *extra,
             dtype_code(x.dtype), stream_ptr())
        # the partials' place inside `scratch` comes from the library (the layout is the .hip file's to change)
        lay = (ctypes.c_long * 6)()
        call("tdeed_gsf_bwd_part_layout", B, T, h * w, F, lay)
        off_cw, PZ, cw_stride, off_w3, n_w3, row = (int(v) for v in lay)
        assert n_w3 == N and row == F * 27 + 2 and cw_stride == 38 and off_w3 + n_w3 * row == scratch.numel(), tuple(lay)
        d_w3 = LazyFold(scratch[off_w3:], N, F * 27, (F, 27), pstride=row)
        d_b3 = LazyFold(scratch[off_w3 + F * 27:], N, 2, pstride=row)
        if not fuse:
            return ret(d_xs, d_bn, d_w3, d_b3, None, None)
        cw = lambda c0, n: LazyFold(scratch[off_cw + c0:], PZ, n, pstride=cw_stride)       # noqa: E731
        return ret(d_xs, d_bn, d_w3, d_b3, (cw(0, 18), cw(19, 18)), (cw(18, 1), cw(37, 1)))
    d_w3, d_b3 = _f32((F, 27), dev), _f32((2,), dev)
    d_cw, d_cb = (_f32((2, 18), dev), _f32((2,), dev)) if fuse else (None, None)
    call(entry, ptr(x), ptr(gate), ptr(fw), ptr(ysum), ptr(xsum), ptr(dA), B, T, h, w, C, F, Fp, ptr(w3), ptr(sa),
         ptr(sb), ptr(cw1), ptr(cw2), ptr(scratch), ptr(d_xs), ptr(d_bn), ptr(d_w3), ptr(d_b3), ptr(d_cw), ptr(d_cb), *extra,
         dtype_code(x.dtype), stream_ptr())
    return ret(d_xs, d_bn, d_w3, d_b3, d_cw, d_cb)


def gsf_add_cols(a, b, dx, Fp):
    C = dx.shape[-1]
    M = dx.numel() // C
    call("tdeed_gsf_add_cols", ptr(a), ptr(b), M, C, Fp, ptr(dx), dtype_code(dx.dtype), stream_ptr())
    return dx


def avgpool_posenc_bwd(d_feat, hw):
    """d_feat (B,T,C) -> dx (B*T, hw, C) in d_feat's dtype, d_temp_enc (T,C) fp32"""
    Bn, T, C = d_feat.shape
    dx = torch.empty((Bn * T, hw, C), dtype=d_feat.dtype, device=d_feat.device)
    d_enc = _f32((T, C), d_feat.device)
    call("tdeed_avgpool_posenc_bwd", ptr(d_feat), Bn, T, hw, C, ptr(dx), ptr(d_enc), dtype_code(d_feat.dtype), stream_ptr())
    return dx, d_enc


def stem_wgrad_bn_fits(frames, crop, dtype):
    H, W = frames.shape[-2], frames.shape[-1]
    top, left, ch, cw = crop if crop is not None else (0, 0, H, W)
    return dtype == torch.bfloat16 and _lib.load().tdeed_stem_wgrad_bn_fits(H, W, ch, cw) != 0


def stem_wgrad(frames_u8, dz, crop=None, flip=False, bn=None):
    """frames (N,3,H,W) uint8, dz (N,Ho,Wo,32) -> dw (32,3,3,3) fp32.  bn = (z, sums, mean, rstd, w): `dz` is the masked gradient
    at the OUTPUT of the stem's BatchNorm, whose backward is applied while the rows are staged (tdeed_stem_wgrad_bn)."""
    N, _, H, W = frames_u8.shape
    top, left, ch, cw = crop if crop is not None else (0, 0, H, W)
    part, dw = _f32((N * ((dz.shape[1] + 15) // 16), 864), dz.device), _f32((32, 3, 3, 3), dz.device)
    from .ops import _flip_args
    fl, fmask = _flip_args(flip, N)
    if bn is not None:
        z, sums, mean, rstd, w = bn
        call("tdeed_stem_wgrad_bn", ptr(frames_u8), int(frames_u8.dtype == torch.float32), N, H, W, top, left, ch, cw, fl,
             ptr(fmask), ptr(dz), ptr(z), ptr(sums), ptr(mean), ptr(rstd), ptr(w), ptr(part), ptr(dw), stream_ptr())
        return dw
    call("tdeed_stem_wgrad", ptr(frames_u8), int(frames_u8.dtype == torch.float32), N, H, W, top, left, ch, cw, fl,
         ptr(fmask), ptr(dz), ptr(part), ptr(dw),
         dtype_code(dz.dtype), stream_ptr())
    return dw


def mix_frames(a_u8, b_u8, lam):
    """mixup of two uint8 clip batches (B,T,3,H,W) with per-clip weights lam (B,) fp32 -> fp32 frames"""
    Bn = a_u8.shape[0]
    out = torch.empty(a_u8.shape, dtype=torch.float32, device=a_u8.device)
    call("tdeed_mix_frames", ptr(a_u8), ptr(b_u8), ptr(lam), Bn, a_u8.numel() // Bn, ptr(out), stream_ptr())
    return out


class PinnedTables:
    """Pinned host buffers for the record tables of multi_copy, one small ring per call site (`role`).  A buffer is
    rewritten only after the copy that last read it has finished (event), so eager steps may run ahead of the GPU;
    under stream capture nothing may be allocated or waited for: the warm-up pass has created the buffers, and a
    replayed graph re-reads the same table from the same buffer."""

    def __init__(self, max_entries=2048, depth=4):
        self.max_entries, self.depth = max_entries, depth
        self.rings = {}

    def get(self, role):
        ring = self.rings.setdefault(role, dict(bufs=[], evs=[], i=0))
        capturing = torch.cuda.is_current_stream_capturing()
        if len(ring["bufs"]) < self.depth and not capturing:
            ring["bufs"].append(torch.empty((self.max_entries, 8), dtype=torch.int64).pin_memory())
            ring["evs"].append(None)
            j = len(ring["bufs"]) - 1
        else:
            if not ring["bufs"]:
                raise RuntimeError("multi_copy under stream capture needs a warm-up call outside the capture first")
            j = 0 if capturing else ring["i"] % len(ring["bufs"])
        ring["i"] = j + 1
        if ring["evs"][j] is not None and not capturing:
            ring["evs"][j].synchronize()
        return ring, j


def multi_copy(srcs, offsets, dst_flat, scale=1.0, accumulate=False, tables=None, role=0):
    """dst_flat[off_i : off_i + n_i] (= or +=) scale * srcs[i] for every i, one launch (tdeed_multi_fold).  srcs: fp32
    device tensors -- contiguous, or 2-D with unit column stride (a column slice of a wider matrix) -- or LazyFold partials,
    which are folded on the way; offsets: element offsets into dst_flat.  tables: a PinnedTables (reused across steps;
    required for calls inside a stream capture)."""
    nt = len(srcs)
    tables = tables if tables is not None else PinnedTables(max(nt, 16), depth=1)
    if nt > tables.max_entries:
        raise ValueError(f"multi_copy: {nt} tensors exceed the table size {tables.max_entries}")
    ring, j = tables.get(role)
    host = ring["bufs"][j]
    rows = []
    keep = []
    wg = 0
    cwf = _lib.load().tdeed_multi_fold_cw
    for t_, off in zip(srcs, offsets):
        if isinstance(t_, LazyFold):
            cw = cwf(t_.P, t_.n)
            if cw == 1024 and (t_.pstride % 4 or t_.part.data_ptr() % 16):
                cw = 256                                          # the 4-wide sequential fold wants 16-byte aligned rows
            rows.append((t_.part.data_ptr(), off, t_.n, wg, t_.P | (cw << 32), t_.pstride, t_.n, t_.n))
            wg += (t_.n + cw - 1) // cw if t_.P > 1 else (t_.n + 4095) // 4096
            keep.append(t_.part)
            continue
        if t_.dtype != torch.float32:
            raise TypeError("multi_copy: fp32 sources expected")
        n = t_.numel()
        sq = t_ if t_.is_contiguous() else t_.squeeze()
        if t_.is_contiguous():
            cols, ld = n, n
        elif sq.dim() == 2 and sq.stride(1) == 1 and sq.stride(0) >= sq.shape[1]:      # column slice of a wider matrix
            cols, ld = sq.shape[1], sq.stride(0)
        elif sq.dim() == 1 and sq.stride(0) >= 1:                                      # one column of it
            cols, ld = 1, sq.stride(0)
        else:
            t_ = t_.contiguous()
            cols, ld = n, n
        rows.append((t_.data_ptr(), off, n, wg, 1 | (4096 << 32), n, cols, ld))
        wg += (n + 4095) // 4096
        keep.append(t_)
    host[:nt] = torch.tensor(rows, dtype=torch.int64)
    dev = host[:nt].to(dst_flat.device, non_blocking=True)
    if not torch.cuda.is_current_stream_capturing():
        ev = torch.cuda.Event()
        ev.record()
        ring["evs"][j] = ev
    ring["live"] = (dev, keep)                       # keep the device table and the sources alive until the next call
    call("tdeed_multi_fold", ptr(dev), nt, wg, ptr(dst_flat), float(scale), int(accumulate), stream_ptr())
    return dev
