"""Train-mode forward + backward of the RegNetY trunk's bottleneck (timm `Bottleneck` with BatchNorm in batch-statistics
mode, as `TDEEDModel.epoch(optimizer=...)` runs it: /root/reference/model/model.py:38-45,133-135, 236-263).

Like temporal_train.py this file only orders launches of the HIP kernels (ops.py / ops_bwd.py), keeps the activations the
backward needs and maps the packed gradient layouts back to the reference's state_dict names.  The gate-shift wrapper of
the s3/s4 blocks (`GatedShift` + `_GSF`, shift.py:64-93, impl/gsf.py:38-93) is `GateShiftTrain`: its forward reuses the
inference kernels with the BatchNorm3d affine taken from batch statistics, its backward is gsf_bwd.hip."""
from types import SimpleNamespace

import torch

from . import ops, ops_bwd as B_, repack as R

BN_EPS = 1e-5
# ReLU masks of the BatchNorm backward recomputed from z (fa * z + fb > 0) instead of read from the stored activation
import os as _os

FUSE_RES = _os.environ.get("TDEED_TRAIN_FUSE_RES", "1") == "1"
# (round 6: the round-2 .. round-4 training forms below are module constants -- bit-identity tests flip them by attribute --
#  no longer environment switches; what is still read from the environment is listed in DESIGN section 8)
ZMASK = True
# SE gate gradient, SE scale backward and conv2's BatchNorm backward from five per-frame sums (trunk_bwd2.hip): 5 passes over
# the block's output-resolution maps instead of 9 (False: pool_rows / scale_rows / bn_train_bwd, what fp32 mode without ZMASK ran)
SE_BN_FUSED = ZMASK
# output-ReLU backward and BatchNorm-backward statistics applied by the producers of each block-input gradient
# (ops_bwd.GradSink; False: the masked statistics pass + d_res map per block)
SINK = True
# the gate-shift module's BatchNorm3d backward: statistics out of the conv3d input-gradient launch, apply inside the kernel that
# adds the module's input gradient into d x (no column-statistics pass, no dz map)
GSF_BN_FUSED = True
GSF_DENSE_IN = True
# K = N = 320 contractions over >= RS_MIN_ROWS rows on the register-stationary kernel (forward with the statistics epilogue,
# conv3's input gradient; False: the tiled kernel everywhere)
RS_TRAIN = True
# statistics of conv1's BatchNorm backward from the epilogue of conv2's (stride-1) input-gradient launch
DGRAD_STATS = True
# conv1's whole backward (BatchNorm apply + input gradient + weight gradient) of the narrow 800MF s1 / s2 layers in one launch
NARROW_BWD = True
# the next block's gate-shift slice written by this block's last BatchNorm-apply pass
SLICE_OUT = True
RS_MIN_ROWS = 60000


def _dense(w, dt):
    w = w.reshape(w.shape[0], -1).contiguous()
    w = w if dt == torch.float32 else R.cast_bf16(w)
    return SimpleNamespace(w=w, wt=R.transpose(w))


class GateShiftTrain:
    """`GatedShift` + `_GSF` on the first F channels of a block input, training mode."""

    def __init__(self, sd, pre, F, T, act_dtype=torch.float32):
        self.sd, self.pre, self.F, self.T, self.dt = sd, pre, F, T, act_dtype
        self.fuse = (pre + ".channel_conv1.weight") in sd          # _GSF; the plain _GSM has no fusion conv
        self.Fp = (F + 7) // 8 * 8
        # BatchNorm3d running statistics live in Fp-wide buffers (pad columns: mean 0, var 1) that the state dict entries
        # alias, so the statistics kernel updates them in place without a copy in / copy out per step
        dev = sd[pre + ".bn.running_mean"].device
        self.rm_pad = torch.zeros(self.Fp, dtype=torch.float32, device=dev)
        self.rv_pad = torch.ones(self.Fp, dtype=torch.float32, device=dev)
        self.rm_pad[:F] = sd[pre + ".bn.running_mean"]
        self.rv_pad[:F] = sd[pre + ".bn.running_var"]
        sd[pre + ".bn.running_mean"], sd[pre + ".bn.running_var"] = self.rm_pad[:F], self.rv_pad[:F]
        self.repack()

    def repack(self):
        sd, pre, F = self.sd, self.pre, self.F
        w3 = sd[pre + ".conv3D.weight"].reshape(F, 27).contiguous()
        self.w3 = w3                                   # [F][27]: backward
        self.wq = w3.t().contiguous()                  # [27][F]: forward (VALU tap kernel)
        self.wqf = None                                # bf16: the tap convolution runs on the MFMA kernel of the inference path
        if self.dt == torch.bfloat16:
            from .engine import gsf_q_frags_on_device
            self.wqf = gsf_q_frags_on_device(sd[pre + ".conv3D.weight"])
        self.b3 = sd[pre + ".conv3D.bias"]
        self.w_pad = R.pad1d(sd[pre + ".bn.weight"], self.Fp)      # BatchNorm3d affine, zero in the pad columns
        self.b_pad = R.pad1d(sd[pre + ".bn.bias"], self.Fp)
        if self.fuse:
            self.cw1, self.cb1 = sd[pre + ".channel_conv1.weight"].reshape(18), sd[pre + ".channel_conv1.bias"]
            self.cw2, self.cb2 = sd[pre + ".channel_conv2.weight"].reshape(18), sd[pre + ".channel_conv2.bias"]
        else:
            self.cw1 = self.cb1 = self.cw2 = self.cb2 = None

    def forward(self, x, xs=None):
        """x (N,h,w,C) -> G (N*h*w, Fp): the module output in conv1's operand layout (pad columns = copies of x).
        xs: the dense slice x[..., :F] (zeros up to Fp) when the producer of x wrote it along (tdeed_bn_apply_slice)."""
        sd, pre, F, Fp, T = self.sd, self.pre, self.F, self.Fp, self.T
        N = x.shape[0]
        c = SimpleNamespace(x=x, B=N // T)
        c.xs = xs if xs is not None else B_.gsf_slice(x, F, Fp)
        c.w_pad = self.w_pad
        c.mean, c.rstd, c.sa, c.sb = B_.bn_stats(c.xs, self.w_pad, self.b_pad, BN_EPS, 0.1, self.rm_pad, self.rv_pad)
        bufs = {}
        dev = x.device
        h, w = x.shape[1], x.shape[2]
        bufs["gate"] = torch.empty((N, h, w, 2), dtype=torch.float32, device=dev)
        bufs["ysum"] = torch.empty((N, F), dtype=torch.float32, device=dev)
        bufs["xsum"] = torch.empty((N, F), dtype=torch.float32, device=dev)
        if self.fuse:
            bufs["fw"] = torch.empty((c.B, F, T), dtype=torch.float32, device=dev)
        # the module's kernels read the DENSE slice where it holds everything they need (F == Fp: no pass-through pad columns;
        # the backward always -- it reads channels < F only): rows of 2 Fp bytes instead of 2 Fp bytes out of every 2 C
        c.xin = c.xs.view(N, h, w, Fp) if (F == Fp and GSF_DENSE_IN) else x
        G = ops.gate_shift(c.xin, c.B, T, F, Fp, c.sa[:F].contiguous(), c.sb[:F].contiguous(), self.wq, self.b3, self.cw1,
                           self.cb1, self.cw2, self.cb2, bufs=bufs, wqf=self.wqf, separate_weight=True)
        c.bufs = bufs
        self.ctx = c
        return G

    def backward(self, dA, grads, fused_bn=False):
        """dA (N*h*w, Fp): gradient of forward()'s output.  Returns (d_xs, dz_bn, None): the two dense (M,Fp) parts of the
        gradient w.r.t. x[..., :Fp] (to be added into the block-input gradient).  fused_bn (and the geometry served): the
        BatchNorm3d backward is left to the kernel that adds the parts into d x -- returns (d_xs, d_bn, bn) with d_bn the
        gradient at the BatchNorm's OUTPUT and bn = (xs, sums, mean, rstd, w) for ops_bwd.gsf_add_cols_sink(bn=...): its
        statistics come from the conv3d input-gradient launch (no pass over (d_bn, xs)), its apply pass does not exist."""
        sd, pre, F, Fp, T, c = self.sd, self.pre, self.F, self.Fp, self.T, self.ctx
        b = c.bufs
        N, h, w, C = c.x.shape
        fused_bn = (fused_bn and GSF_BN_FUSED and dA.dtype == torch.bfloat16
                    and B_.gsf_bwd_bn_parts(c.B, T, h, w, (Fp if GSF_DENSE_IN else C), Fp) > 0)
        xb = c.xs.view(N, h, w, Fp) if GSF_DENSE_IN else c.x
        r = B_.gsf_bwd(xb, b["gate"], b.get("fw"), b["ysum"], b["xsum"], dA, c.B, T, F, Fp, self.w3, c.sa[:F].contiguous(),
                       c.sb[:F].contiguous(), self.cw1, self.cw2, bn_mean=(c.mean if fused_bn else None))
        d_xs, d_bn, d_w3, d_b3, d_cw, d_cb = r[:6]
        bn = None
        if fused_bn:
            sink = SimpleNamespace(partA=r[6], partB=None, nB=0)
            sums = B_.bn_sums_from_sink(c.xs, d_bn, (c.mean, c.rstd), c.w_pad, sink, q=1)       # [2][Fp]: d bias | d weight
            dw, db = sums[1], sums[0]
            bn = (c.xs, sums, c.mean, c.rstd, c.w_pad)
            dz = d_bn
        else:
            dz, _, dw, db = B_.bn_train_bwd(c.xs, d_bn, None, (c.mean, c.rstd), c.w_pad, relu=False)
        grads[pre + ".conv3D.weight"] = d_w3.reshape(sd[pre + ".conv3D.weight"].shape)
        grads[pre + ".conv3D.bias"] = d_b3
        grads[pre + ".bn.weight"], grads[pre + ".bn.bias"] = dw[:F].contiguous(), db[:F].contiguous()
        if self.fuse:
            grads[pre + ".channel_conv1.weight"] = d_cw[0].reshape(sd[pre + ".channel_conv1.weight"].shape)
            grads[pre + ".channel_conv2.weight"] = d_cw[1].reshape(sd[pre + ".channel_conv2.weight"].shape)
            lazy = isinstance(d_cb, tuple)                                  # (partials folded by the gradient write-out)
            grads[pre + ".channel_conv1.bias"] = d_cb[0] if lazy else d_cb[0:1].contiguous()
            grads[pre + ".channel_conv2.bias"] = d_cb[1] if lazy else d_cb[1:2].contiguous()
        return d_xs, dz, bn


class StemTrain:
    """Packed copy of the stem conv weight for the MFMA training stem (bf16 mode; refreshed with the other kernel-layout
    copies after every optimizer step)."""

    def __init__(self, sd, dt):
        self.sd, self.dt = sd, dt
        self.repack()

    def repack(self):
        self.wf = None
        if self.dt == torch.bfloat16:
            from .engine import stem_frags_on_device
            self.wf = stem_frags_on_device(self.sd["_features.stem.conv.weight"])


class BottleneckTrain:
    """One bottleneck in training mode.  sd: name -> fp32 master tensor on the device (reference names); BatchNorm
    running statistics in sd are updated in place by forward()."""

    def __init__(self, sd, pre, blk, act_dtype=torch.float32, clip_len=None):
        self.sd, self.pre, self.blk, self.dt = sd, pre, blk, act_dtype
        self.pack_gen = 0                     # bumped by whoever refreshes the packed weights (TrainEngine.repack)
        self.c1 = pre + (".conv1.net" if blk.gsf_fold else ".conv1")          # GatedShift keeps the conv as .net
        self.gs = GateShiftTrain(sd, pre + ".conv1.gs", blk.gsf_fold, clip_len, act_dtype) if blk.gsf_fold else None
        # BatchNorm statistics out of the producing conv's epilogue (bf16 MFMA kernels); epi_stats = False (fp32 mode) is the
        # separate column-statistics pass
        import os
        self.epi_stats = act_dtype == torch.bfloat16 and True
        dev = sd[self.c1 + ".conv.weight"].device
        self.one, self.zero = torch.ones(blk.cout, device=dev), torch.zeros(blk.cout, device=dev)
        # the post-BN maps behind conv1 and conv2 are not materialised: their consumers (grouped conv forward and weight
        # gradient; SE squeeze / scale and the gate gradient) apply relu(a*z + b) in their own loads.  Needs the epilogue
        # statistics, the z-recomputed ReLU masks and the bf16 MFMA kernels; TDEED_TRAIN_ONLOAD=0 restores the maps.
        self.onload = (self.epi_stats and ZMASK and os.environ.get("TDEED_TRAIN_ONLOAD", "1") == "1")
        self.repack()

    def _check_recompute(self, c):
        """The narrow one-launch backward is called with recompute=True: it re-derives z = x @ W^T from the packed transposed
        weights instead of reading the saved map.  That equals the forward's z only while those are the weights the forward
        multiplied with (same packing, same k order and rounding: the forward's own kernel layout) -- so no repack / optimizer
        step may lie between a forward and its backward."""
        if c.pack_gen != self.pack_gen:
            raise RuntimeError(f"{self.pre}: the packed weights changed (repack generation {c.pack_gen} -> {self.pack_gen}) "
                               "between forward and backward; the narrow backward recomputes z from them")

    def repack(self):
        self.pack_gen = getattr(self, "pack_gen", 0) + 1
        sd, pre, blk, dt = self.sd, self.pre, self.blk, self.dt
        dev = sd[self.c1 + ".conv.weight"].device
        if self.gs is not None:
            self.gs.repack()
        self.w1 = _dense(sd[self.c1 + ".conv.weight"], dt)
        self.w3 = _dense(sd[pre + ".conv3.conv.weight"], dt)
        # 320-wide layers over many rows (the s3 blocks of RegNetY-800MF): fragment-ordered copies for the register-stationary
        # contraction (W in registers, activations cross the chip once: 106 vs 171 us per call at M = 313 600)
        self.w1.ws = self.w1.wst_ws = self.w3.ws = self.w3.wst_ws = None
        if dt == torch.bfloat16 and RS_TRAIN and blk.cin == blk.cout and ops.gemm_rs_fits(1 << 20, blk.cin, blk.cout):
            W1 = sd[self.c1 + ".conv.weight"].reshape(blk.cout, blk.cin)
            W3 = sd[pre + ".conv3.conv.weight"].reshape(blk.cout, blk.cout)
            self.w1.ws = R.to_bf16(R.pack_ws(W1))
            self.w1.wst_ws = R.to_bf16(R.pack_ws(W1.t()))
            self.w3.ws = R.to_bf16(R.pack_ws(W3))
            self.w3.wst_ws = R.to_bf16(R.pack_ws(W3.t()))
        self.wd = _dense(sd[pre + ".downsample.conv.weight"], dt) if blk.has_downsample else None
        G, gw = blk.groups, blk.gw
        self.w2p = (sd[pre + ".conv2.conv.weight"].reshape(G, gw, gw, 3, 3).permute(0, 3, 4, 2, 1)
                    .reshape(G, 9, gw, gw).contiguous())
        # bf16: conv2 forward and (stride 1) its input gradient run on the MFMA grouped-conv kernel; the input gradient is
        # a grouped conv of dy with the weights flipped in space and transposed inside each group
        self.w2frag = self.w2frag_t = None
        if dt == torch.bfloat16:
            from .engine import gconv_frags_on_device
            w2 = sd[pre + ".conv2.conv.weight"]
            self.w2frag = gconv_frags_on_device(w2, gw)
            if blk.stride == 1:
                wt = w2.reshape(G, gw, gw, 3, 3).transpose(1, 2).flip(3, 4).reshape(blk.cout, gw, 3, 3)
                self.w2frag_t = gconv_frags_on_device(wt, gw)
        Rd, C = blk.se_rd, blk.cout
        self.se_w1 = sd[pre + ".se.fc1.weight"].reshape(Rd, C).contiguous()
        self.se_w2 = sd[pre + ".se.fc2.weight"].reshape(C, Rd).contiguous()
        self.se_w1t, self.se_w2t = self.se_w1.t().contiguous(), self.se_w2.t().contiguous()

    def _bn(self, z, name, res=None, relu=True, part=None, apply=True, res_affine=None, slice_out=None):
        """BatchNorm(batch statistics) + residual + ReLU of a raw conv output.  part = (sums, sums of squares, row stride,
        rows): the per-channel partial sums the conv's own epilogue wrote (no second pass over z for the statistics).
        apply=False (with part): statistics and affine only, the map is applied by its consumers."""
        sd, p = self.sd, (self.c1 if name == "conv1" else f"{self.pre}.{name}") + ".bn"
        if part is not None:
            ps, pq, stride, P = part
            return B_.bn_finalize_apply(z, ps, pq, stride, P, sd[p + ".weight"], sd[p + ".bias"], BN_EPS, 0.1,
                                        sd[p + ".running_mean"], sd[p + ".running_var"], res=res, relu=relu, apply=apply,
                                        res_affine=res_affine, slice_out=slice_out)
        return B_.bn_train(z, sd[p + ".weight"], sd[p + ".bias"], BN_EPS, 0.1, sd[p + ".running_mean"],
                           sd[p + ".running_var"], res=res, relu=relu)

    def _conv1x1(self, a, w, M, N, A0=None, k0=0, ws=None):
        """raw 1x1 conv; with the column statistics of its output from the epilogue -> (z, part | None)"""
        if ws is not None and self.epi_stats and M >= RS_MIN_ROWS:
            return ops.gemm_rs_stats(a, ws, a.shape[-1], N, M=M, A0=A0, k0=k0)
        if not self.epi_stats:
            return ops.gemm(a, w, None, None, ops.ACT_NONE, M=M, A0=A0, k0=k0), None
        P = ops.gemm_colpart_rows(M)
        cp = torch.empty((P, 2, N), dtype=torch.float32, device=a.device)
        z = ops.gemm(a, w, None, None, ops.ACT_NONE, M=M, colpart=cp, A0=A0, k0=k0)
        flat = cp.view(-1)
        return z, (flat, flat[N:], 2 * N, P)

    def forward(self, x, xs=None, next_fold=0):
        """x (N,h,w,Cin) activation dtype -> (N,h2,w2,Cout); ctx kept on self.  xs: the compact gate-shift slice of x when its
        producer wrote one; next_fold: fold F of the NEXT block's gate-shift module -- this block's last pass then writes that
        block's slice along (ctx.out_slice) instead of a tdeed_gsf_slice pass over the map."""
        blk, sd, pre = self.blk, self.sd, self.pre
        N, h, w, Cin = x.shape
        C = blk.cout
        c = SimpleNamespace(x=x)
        c.G = None
        c.out_slice = None
        c.pack_gen = self.pack_gen           # which packing of the weights this forward multiplied with (checked in backward)
        if self.gs is not None:
            Fp = self.gs.Fp
            G = self.gs.forward(x, xs=xs)
            if B_.wgrad_splice_ok(self.dt, N * h * w) and _os.environ.get("TDEED_TRAIN_SPLICE", "1") == "1":
                # conv1 operand [G | x[..., Fp:]] is never built: the contraction and its weight gradient read the first Fp
                # columns from G and the rest from x (shift.py:89-93 as an operand splice, like the inference path)
                c.a1, c.G = x, G
            else:
                c.a1 = x.clone()
                c.a1.view(-1, Cin)[:, :Fp] = G
        else:
            c.a1 = x
        z1, part = self._conv1x1(c.a1, self.w1.w, N * h * w, C, A0=c.G, k0=(self.gs.Fp if c.G is not None else 0),
                                 ws=self.w1.ws)
        c.z1 = z1.view(N, h, w, C)
        onload = (self.onload and part is not None and self.w2frag is not None
                  and ops.gconv3x3_mfma_fits(h, w, C, blk.stride))
        c.onload = onload
        c.y1, c.bn1 = self._bn(c.z1, "conv1", part=part, apply=not onload)
        aff1 = (c.bn1[2], c.bn1[3]) if onload else None
        if self.epi_stats and self.w2frag is not None:
            parts = ops.gconv3x3_parts(h, w, C, blk.stride, self.dt)
            psq = torch.empty((N, parts, C), dtype=torch.float32, device=x.device)
            c.z2, pooled = ops.gconv3x3(c.z1 if onload else c.y1, self.w2p, self.one, self.zero, blk.gw, blk.stride,
                                        wfrag=self.w2frag, relu=False, pooled_sq=psq, in_affine=aff1)
            part2 = (pooled.view(-1), psq.view(-1), C, N * parts)
        else:
            c.z2, _ = ops.gconv3x3(c.y1, self.w2p, self.one, self.zero, blk.gw, blk.stride, wfrag=self.w2frag, relu=False)
            part2 = None
        c.y2, c.bn2 = self._bn(c.z2, "conv2", part=part2, apply=not onload)
        aff2 = (c.bn2[2], c.bn2[3]) if onload else None
        h2, w2 = c.z2.shape[1], c.z2.shape[2]
        c.p = B_.pool_rows(c.z2 if onload else c.y2, affine=aff2)
        c.hid, c.gate = B_.se_train_fwd(c.p, self.se_w1t, sd[pre + ".se.fc1.bias"], self.se_w2t, sd[pre + ".se.fc2.bias"])
        c.y2s = B_.scale_rows(c.z2 if onload else c.y2, c.gate, affine=aff2)
        z3, part3 = self._conv1x1(c.y2s, self.w3.w, N * h2 * w2, C, ws=self.w3.ws)
        c.z3 = z3.view(N, h2, w2, C)
        if blk.has_downsample:
            c.xs = B_.stride2_gather(x) if blk.stride == 2 else x
            zd, partd = self._conv1x1(c.xs, self.wd.w, N * h2 * w2, C)
            c.zd = zd.view(N, h2, w2, C)
            # with both statistics from the contractions' epilogues the shortcut's BatchNorm is applied inside conv3's apply pass
            # (tdeed_bn_apply2): the normalised shortcut map is not written and read back
            fuse_sc = partd is not None and part3 is not None and True
            c.sc, c.bnd = self._bn(c.zd, "downsample", relu=False, part=partd, apply=not fuse_sc)
        else:
            c.sc = x
            fuse_sc = False
        so = None
        if next_fold and part3 is not None and SLICE_OUT:
            Fp2 = (next_fold + 7) // 8 * 8
            c.out_slice = torch.empty((N * h2 * w2, Fp2), dtype=c.z3.dtype, device=c.z3.device)
            so = (next_fold, c.out_slice)
        if fuse_sc:
            c.out, c.bn3 = self._bn(c.z3, "conv3", res=c.zd, relu=True, part=part3, res_affine=(c.bnd[2], c.bnd[3]), slice_out=so)
        else:
            c.out, c.bn3 = self._bn(c.z3, "conv3", res=c.sc, relu=True, part=part3, slice_out=so)
        self.ctx = c
        return c.out

    def make_sink(self):
        """The sink the producers of this block's OUTPUT gradient fill (the next block's backward): output ReLU mask and
        the statistics of the conv3 / shortcut BatchNorm backward (ops_bwd.GradSink)."""
        c, ds = self.ctx, self.blk.has_downsample
        return B_.GradSink(c.out, c.z3, c.bn3[0], zd=(c.zd if ds else None), mean_d=(c.bnd[0] if ds else None))

    def backward(self, dout, grads, sink_in=None, sink_out=None):
        """dout: gradient of forward()'s output; fills grads[name] for this block's parameters, returns d x.
        sink_in: dout is ALREADY masked by this block's output ReLU and its BatchNorm column sums lie in sink_in (whoever
        produced dout filled the sink make_sink() returned); sink_out: the sink of the block in front, applied by the kernels
        that write d x here (then d x comes back masked, too)."""
        blk, sd, pre, c = self.blk, self.sd, self.pre, self.ctx
        N, h2, w2, C = c.out.shape
        hw2 = h2 * w2

        def bn_names(name, dw, db):
            p = (self.c1 if name == "conv1" else f"{pre}.{name}") + ".bn"
            grads[p + ".weight"], grads[p + ".bias"] = dw, db

        if sink_in is not None and NARROW_BWD and B_.narrow_conv1_bwd_fits(C, C, dout.dtype):
            # narrow layers: conv3's BatchNorm apply pass, its input gradient and its weight gradient in one launch (the conv1
            # kernel with no ReLU between BatchNorm and conv: fa = 0, fb = 1; dz3 lives in LDS only)
            self._check_recompute(c)
            sums3 = B_.bn_sums_from_sink(c.z3, dout, c.bn3, sd[pre + ".conv3.bn.weight"], sink_in, q=1)
            d_y2s, dW3, dw, db = B_.narrow_conv1_bwd(dout, c.z3, (c.bn3[0], c.bn3[1], self.zero, self.one),
                                                     sd[pre + ".conv3.bn.weight"], None, c.y2s, self.w3.wt, sums=sums3,
                                                     recompute=True)
            d_y2s = d_y2s.view(N, h2, w2, C)
            d_sc = dout
            bn_names("conv3", dw, db)
            grads[pre + ".conv3.conv.weight"] = dW3.reshape(sd[pre + ".conv3.conv.weight"].shape)
        else:
            if sink_in is not None:
                # only the apply pass is left of this BatchNorm backward, and dout itself is the shortcut's gradient
                dz3, dw, db = B_.bn_bwd_from_parts(c.z3, dout, c.bn3, sd[pre + ".conv3.bn.weight"], sink_in, q=1)
                d_sc = dout
            else:
                dz3, d_sc, dw, db = B_.bn_train_bwd(c.z3, dout, c.out, c.bn3, sd[pre + ".conv3.bn.weight"], relu=True, want_res=True)
            bn_names("conv3", dw, db)
            if self.w3.wst_ws is not None and N * hw2 >= RS_MIN_ROWS:
                d_y2s = ops.gemm_rs(dz3, self.w3.wst_ws, C, C, M=N * hw2).view(N, h2, w2, C)
            else:
                d_y2s = ops.gemm(dz3, self.w3.wt, None, None, ops.ACT_NONE).view(N, h2, w2, C)
            grads[pre + ".conv3.conv.weight"] = B_.wgrad(dz3, c.y2s, with_bias=False, M=N * hw2)[0].reshape(
                sd[pre + ".conv3.conv.weight"].shape)
        # SE + conv2's BatchNorm
        if SE_BN_FUSED and len(c.bn2) >= 4:
            # one pass over (d_y2s, z2) for every per-frame sum the SE gate gradient and the BatchNorm statistics need, one
            # pass that writes dz2: d_y2 = d_y2s * gate + d_pool / hw is never materialised (trunk_bwd2.hip)
            dz2, dw_bn2, db_bn2, d_pre2, d_hid = B_.se_bn_bwd(d_y2s, c.z2, c.bn2, sd[pre + ".conv2.bn.weight"], c.gate, c.hid,
                                                             self.se_w1, self.se_w2)
        else:
            if c.onload:
                d_gate = B_.pool_rows(d_y2s, c.z2, affine=(c.bn2[2], c.bn2[3]), affine_on=2)
            else:
                d_gate = B_.pool_rows(d_y2s, c.y2)
            d_pre2, d_hid, d_p = B_.se_train_bwd(d_gate, c.gate, c.hid, self.se_w1, self.se_w2)
            d_y2 = B_.scale_rows(d_y2s, c.gate, add=d_p, add_scale=1.0 / hw2)
            dz2, _, dw_bn2, db_bn2 = B_.bn_train_bwd(c.z2, d_y2, None if ZMASK else c.y2, c.bn2, sd[pre + ".conv2.bn.weight"],
                                                     relu=True)
        dW2, db2 = B_.wgrad(d_pre2, c.hid)
        dW1, db1 = B_.wgrad(d_hid, c.p)
        grads[pre + ".se.fc1.weight"] = dW1.reshape(sd[pre + ".se.fc1.weight"].shape)
        grads[pre + ".se.fc1.bias"] = db1
        grads[pre + ".se.fc2.weight"] = dW2.reshape(sd[pre + ".se.fc2.weight"].shape)
        grads[pre + ".se.fc2.bias"] = db2
        # conv2
        bn_names("conv2", dw_bn2, db_bn2)
        xin, aff1 = (c.z1, (c.bn1[2], c.bn1[3])) if c.onload else (c.y1, None)
        part1 = None
        if self.w2frag_t is not None:
            if (DGRAD_STATS and ZMASK and len(c.bn1) >= 4 and dz2.dtype == torch.bfloat16
                    and ops.gconv3x3_mfma_fits(h2, w2, C, 1)):
                # ... and the statistics of conv1's BatchNorm backward out of the same launch: no pass over (d_y1, z1) for them
                d_y1, part1 = B_.gconv3x3_dgrad_stats(dz2, self.w2frag_t, self.one, self.zero, blk.gw, c.z1, c.bn1)
            else:
                d_y1, _ = ops.gconv3x3(dz2, self.w2p, self.one, self.zero, blk.gw, 1, wfrag=self.w2frag_t, relu=False)
            _, dw2p = B_.gconv3x3_bwd(xin, dz2, self.w2p, blk.gw, blk.stride, want_dx=False, in_affine=aff1)
        elif (DGRAD_STATS and ZMASK and len(c.bn1) >= 4 and blk.stride == 2 and dz2.dtype == torch.bfloat16
              and B_.gconv3x3_bwd_stats_fits(xin.shape[0], xin.shape[1], xin.shape[2], C, blk.gw)):
            d_y1, dw2p, part1 = B_.gconv3x3_bwd_stats(xin, dz2, self.w2p, blk.gw, c.z1, c.bn1, in_affine=aff1)
        else:
            d_y1, dw2p = B_.gconv3x3_bwd(xin, dz2, self.w2p, blk.gw, blk.stride, in_affine=aff1)
        G, gw = blk.groups, blk.gw
        grads[pre + ".conv2.conv.weight"] = (dw2p.reshape(G, 3, 3, gw, gw).permute(0, 4, 3, 1, 2)
                                             .reshape(sd[pre + ".conv2.conv.weight"].shape).contiguous())
        # conv1
        Nf, h, w, Cin = c.x.shape
        if (part1 is not None and NARROW_BWD and sink_out is not None and self.gs is None and c.G is None
                and B_.narrow_conv1_bwd_fits(C, Cin, d_y1.dtype) and sink_out.mask.data_ptr() == c.x.data_ptr()):
            # narrow layers (RegNetY-800MF s1 / s2): BatchNorm + ReLU backward, input gradient (+ shortcut gradient, the sink of
            # the block in front) and weight gradient in ONE launch; dz1 lives in LDS only (trunk_bwd3.hip)
            self._check_recompute(c)
            res, r_hw = d_sc.view(-1, Cin) if not blk.has_downsample else None, None
            if blk.has_downsample:
                wdn = sd[pre + ".downsample.bn.weight"]
                if sink_in is not None and B_.narrow_conv1_bwd_fits(C, Cin, dout.dtype):
                    # the shortcut conv's backward through the same one-launch kernel (no ReLU, no sink)
                    sums_d = B_.bn_sums_from_sink(c.zd, dout, c.bnd, wdn, sink_in, q=2)
                    res, dWd, dw, db = B_.narrow_conv1_bwd(dout, c.zd, (c.bnd[0], c.bnd[1], self.zero, self.one), wdn, None,
                                                           c.xs, self.wd.wt, sums=sums_d, recompute=True)
                    bn_names("downsample", dw, db)
                    grads[pre + ".downsample.conv.weight"] = dWd.reshape(sd[pre + ".downsample.conv.weight"].shape)
                else:
                    if sink_in is not None:
                        dzd, dw, db = B_.bn_bwd_from_parts(c.zd, dout, c.bnd, wdn, sink_in, q=2)
                    else:
                        dzd, _, dw, db = B_.bn_train_bwd(c.zd, d_sc, None, c.bnd, wdn, relu=False)
                    bn_names("downsample", dw, db)
                    res = ops.gemm(dzd, self.wd.wt, None, None, ops.ACT_NONE)
                    grads[pre + ".downsample.conv.weight"] = B_.wgrad(dzd, c.xs, with_bias=False, M=N * hw2)[0].reshape(
                        sd[pre + ".downsample.conv.weight"].shape)
                r_hw = (h, w) if blk.stride == 2 else None
            # (z3 / zd / z1 of these layers are the raw products of exactly the operands the launch holds: recomputed, not read)
            dx, dW1, dw, db = B_.narrow_conv1_bwd(d_y1, c.z1, c.bn1, sd[self.c1 + ".bn.weight"], part1, c.x, self.w1.wt,
                                                  sink=sink_out, residual=res, r_hw=r_hw, recompute=True)
            bn_names("conv1", dw, db)
            grads[self.c1 + ".conv.weight"] = dW1.reshape(sd[self.c1 + ".conv.weight"].shape)
            return dx.view(Nf, h, w, Cin)
        if part1 is not None:
            dz1, dw, db = B_.bn_bwd_masked_from_parts(c.z1, d_y1, c.bn1, sd[self.c1 + ".bn.weight"], part1)
        else:
            dz1, _, dw, db = B_.bn_train_bwd(c.z1, d_y1, None if ZMASK else c.y1, c.bn1, sd[self.c1 + ".bn.weight"], relu=True)
        bn_names("conv1", dw, db)
        Nf, h, w, Cin = c.x.shape
        # identity shortcut: its gradient joins dx in the contraction's epilogue (no separate add pass); behind a gate-shift the
        # first Fp columns of the contraction belong to the module's backward alone: they leave as a compact tensor and dx keeps
        # only the shortcut gradient there (no slice copy / zero fill)
        if sink_out is not None:
            # the contraction's epilogue takes every other contribution to d x as its residual (identity shortcut: d_sc; shortcut
            # conv: its input gradient, on the even pixels of a stride-2 block -- no scatter-add pass), applies the ReLU mask of
            # the block in front and leaves that block's BatchNorm column sums (tdeed_gemm_dgrad)
            res, r_hw = d_sc.view(-1, Cin) if not blk.has_downsample else None, None
            if blk.has_downsample:
                wdn = sd[pre + ".downsample.bn.weight"]
                if sink_in is not None:
                    dzd, dw, db = B_.bn_bwd_from_parts(c.zd, dout, c.bnd, wdn, sink_in, q=2)
                else:
                    dzd, _, dw, db = B_.bn_train_bwd(c.zd, d_sc, None, c.bnd, wdn, relu=False)
                bn_names("downsample", dw, db)
                res = ops.gemm(dzd, self.wd.wt, None, None, ops.ACT_NONE)
                grads[pre + ".downsample.conv.weight"] = B_.wgrad(dzd, c.xs, with_bias=False, M=N * hw2)[0].reshape(
                    sd[pre + ".downsample.conv.weight"].shape)
                r_hw = (h, w) if blk.stride == 2 else None
            dA = (torch.empty((Nf * h * w, self.gs.Fp), dtype=dz1.dtype, device=dz1.device) if self.gs is not None else None)
            dx = B_.gemm_dgrad(dz1, self.w1.wt, sink=sink_out, residual=res, r_hw=r_hw, out2=dA,
                               wt_ws=self.w1.wst_ws).view(Nf, h, w, Cin)
            grads[self.c1 + ".conv.weight"] = B_.wgrad(dz1, c.a1, with_bias=False, M=Nf * h * w, X0=c.G,
                                                       k0=(self.gs.Fp if c.G is not None else 0))[0].reshape(
                sd[self.c1 + ".conv.weight"].shape)
            if self.gs is not None:
                d_xs, dz_bn, bn = self.gs.backward(dA, grads, fused_bn=True)
                B_.gsf_add_cols_sink(d_xs, dz_bn, dx, self.gs.Fp, sink_out, bn=bn)
            return dx
        fuse = FUSE_RES and not blk.has_downsample
        dA = (torch.empty((Nf * h * w, self.gs.Fp), dtype=dz1.dtype, device=dz1.device)
              if (self.gs is not None and FUSE_RES) else None)
        dx = ops.gemm(dz1, self.w1.wt, None, None, ops.ACT_NONE, residual=(d_sc.view(-1, Cin) if fuse else None),
                      out2=dA, out2_pre=dA is not None).view(Nf, h, w, Cin)
        grads[self.c1 + ".conv.weight"] = B_.wgrad(dz1, c.a1, with_bias=False, M=Nf * h * w, X0=c.G,
                                                   k0=(self.gs.Fp if c.G is not None else 0))[0].reshape(
            sd[self.c1 + ".conv.weight"].shape)
        if self.gs is not None:
            Fp = self.gs.Fp
            if dA is None:
                dA = dx.view(-1, Cin)[:, :Fp].contiguous()                      # gradient of the gate-shift output
                dx.view(-1, Cin)[:, :Fp] = 0
            d_xs, dz_bn, _ = self.gs.backward(dA, grads)
            B_.gsf_add_cols(d_xs, dz_bn, dx, Fp)
        # shortcut
        if blk.has_downsample:
            dzd, _, dw, db = B_.bn_train_bwd(c.zd, d_sc, None, c.bnd, sd[pre + ".downsample.bn.weight"], relu=False)
            bn_names("downsample", dw, db)
            d_xs = ops.gemm(dzd, self.wd.wt, None, None, ops.ACT_NONE).view(c.xs.shape)
            grads[pre + ".downsample.conv.weight"] = B_.wgrad(dzd, c.xs, with_bias=False, M=N * hw2)[0].reshape(
                sd[pre + ".downsample.conv.weight"].shape)
            if blk.stride == 2:
                B_.stride2_scatter_add(d_xs, dx)
            else:
                dx = B_.eltwise(dx, d_xs, B_.ADD)
        elif not fuse:
            dx = B_.eltwise(dx, d_sc, B_.ADD)
        return dx
