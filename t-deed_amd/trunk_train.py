"""Train-mode forward + backward of the RegNetY trunk's bottleneck (timm `Bottleneck` with BatchNorm in batch-statistics
mode, as `TDEEDModel.epoch(optimizer=...)` runs it: /root/reference/model/model.py:38-45,133-135, 236-263).

Like temporal_train.py this file only orders launches of the HIP kernels (ops.py / ops_bwd.py), keeps the activations the
backward needs and maps the packed gradient layouts back to the reference's state_dict names.  The gate-shift wrapper of
the s3/s4 blocks (shift.py:64-93) is not differentiated yet: blocks are handled with `gsf_fold == 0`."""
from types import SimpleNamespace

import torch

from . import ops, ops_bwd as B_

BN_EPS = 1e-5


def _dense(w, dt):
    w = w.reshape(w.shape[0], -1).contiguous()
    w = w if dt == torch.float32 else ops.cast_bf16(w)
    return SimpleNamespace(w=w, wt=B_.transpose(w))


class BottleneckTrain:
    """One bottleneck in training mode.  sd: name -> fp32 master tensor on the device (reference names); BatchNorm
    running statistics in sd are updated in place by forward()."""

    def __init__(self, sd, pre, blk, act_dtype=torch.float32):
        assert blk.gsf_fold == 0, "gate-shift blocks: backward not built yet"
        self.sd, self.pre, self.blk, self.dt = sd, pre, blk, act_dtype
        self.repack()

    def repack(self):
        sd, pre, blk, dt = self.sd, self.pre, self.blk, self.dt
        dev = sd[pre + ".conv1.conv.weight"].device
        self.w1 = _dense(sd[pre + ".conv1.conv.weight"], dt)
        self.w3 = _dense(sd[pre + ".conv3.conv.weight"], dt)
        self.wd = _dense(sd[pre + ".downsample.conv.weight"], dt) if blk.has_downsample else None
        G, gw = blk.groups, blk.gw
        self.w2p = (sd[pre + ".conv2.conv.weight"].reshape(G, gw, gw, 3, 3).permute(0, 3, 4, 2, 1)
                    .reshape(G, 9, gw, gw).contiguous())
        self.one, self.zero = torch.ones(blk.cout, device=dev), torch.zeros(blk.cout, device=dev)
        R, C = blk.se_rd, blk.cout
        self.se_w1 = sd[pre + ".se.fc1.weight"].reshape(R, C).contiguous()
        self.se_w2 = sd[pre + ".se.fc2.weight"].reshape(C, R).contiguous()
        self.se_w1t, self.se_w2t = self.se_w1.t().contiguous(), self.se_w2.t().contiguous()

    def _bn(self, z, name, res=None, relu=True):
        sd, p = self.sd, f"{self.pre}.{name}.bn"
        return B_.bn_train(z, sd[p + ".weight"], sd[p + ".bias"], BN_EPS, 0.1, sd[p + ".running_mean"],
                           sd[p + ".running_var"], res=res, relu=relu)

    def forward(self, x):
        """x (N,h,w,Cin) activation dtype -> (N,h2,w2,Cout); ctx kept on self."""
        blk, sd, pre = self.blk, self.sd, self.pre
        N, h, w, Cin = x.shape
        C = blk.cout
        c = SimpleNamespace(x=x)
        c.z1 = ops.gemm(x, self.w1.w, None, None, ops.ACT_NONE).view(N, h, w, C)
        c.y1, c.bn1 = self._bn(c.z1, "conv1")
        c.z2, _ = ops.gconv3x3(c.y1, self.w2p, self.one, self.zero, blk.gw, blk.stride, relu=False)
        c.y2, c.bn2 = self._bn(c.z2, "conv2")
        h2, w2 = c.z2.shape[1], c.z2.shape[2]
        c.p = B_.pool_rows(c.y2)
        c.hid, c.gate = B_.se_train_fwd(c.p, self.se_w1t, sd[pre + ".se.fc1.bias"], self.se_w2t, sd[pre + ".se.fc2.bias"])
        c.y2s = B_.scale_rows(c.y2, c.gate)
        c.z3 = ops.gemm(c.y2s, self.w3.w, None, None, ops.ACT_NONE).view(N, h2, w2, C)
        if blk.has_downsample:
            c.xs = B_.stride2_gather(x) if blk.stride == 2 else x
            c.zd = ops.gemm(c.xs, self.wd.w, None, None, ops.ACT_NONE).view(N, h2, w2, C)
            c.sc, c.bnd = self._bn(c.zd, "downsample", relu=False)
        else:
            c.sc = x
        c.out, c.bn3 = self._bn(c.z3, "conv3", res=c.sc, relu=True)
        self.ctx = c
        return c.out

    def backward(self, dout, grads):
        """dout: gradient of forward()'s output; fills grads[name] for this block's parameters, returns d x."""
        blk, sd, pre, c = self.blk, self.sd, self.pre, self.ctx
        N, h2, w2, C = c.out.shape
        hw2 = h2 * w2

        def bn_names(name, dw, db):
            grads[f"{pre}.{name}.bn.weight"], grads[f"{pre}.{name}.bn.bias"] = dw, db

        dz3, d_sc, dw, db = B_.bn_train_bwd(c.z3, dout, c.out, c.bn3, sd[pre + ".conv3.bn.weight"], relu=True, want_res=True)
        bn_names("conv3", dw, db)
        d_y2s = ops.gemm(dz3, self.w3.wt, None, None, ops.ACT_NONE).view(N, h2, w2, C)
        grads[pre + ".conv3.conv.weight"] = B_.wgrad(dz3, c.y2s, with_bias=False, M=N * hw2)[0].reshape(
            sd[pre + ".conv3.conv.weight"].shape)
        # SE
        d_gate = B_.pool_rows(d_y2s, c.y2)
        d_pre2, d_hid, d_p = B_.se_train_bwd(d_gate, c.gate, c.hid, self.se_w1, self.se_w2)
        d_y2 = B_.scale_rows(d_y2s, c.gate, add=d_p, add_scale=1.0 / hw2)
        dW2, db2 = B_.wgrad(d_pre2, c.hid)
        dW1, db1 = B_.wgrad(d_hid, c.p)
        grads[pre + ".se.fc1.weight"] = dW1.reshape(sd[pre + ".se.fc1.weight"].shape)
        grads[pre + ".se.fc1.bias"] = db1
        grads[pre + ".se.fc2.weight"] = dW2.reshape(sd[pre + ".se.fc2.weight"].shape)
        grads[pre + ".se.fc2.bias"] = db2
        # conv2
        dz2, _, dw, db = B_.bn_train_bwd(c.z2, d_y2, c.y2, c.bn2, sd[pre + ".conv2.bn.weight"], relu=True)
        bn_names("conv2", dw, db)
        d_y1, dw2p = B_.gconv3x3_bwd(c.y1, dz2, self.w2p, blk.gw, blk.stride)
        G, gw = blk.groups, blk.gw
        grads[pre + ".conv2.conv.weight"] = (dw2p.reshape(G, 3, 3, gw, gw).permute(0, 4, 3, 1, 2)
                                             .reshape(sd[pre + ".conv2.conv.weight"].shape).contiguous())
        # conv1
        dz1, _, dw, db = B_.bn_train_bwd(c.z1, d_y1, c.y1, c.bn1, sd[pre + ".conv1.bn.weight"], relu=True)
        bn_names("conv1", dw, db)
        Nf, h, w, Cin = c.x.shape
        dx = ops.gemm(dz1, self.w1.wt, None, None, ops.ACT_NONE).view(Nf, h, w, Cin)
        grads[pre + ".conv1.conv.weight"] = B_.wgrad(dz1, c.x, with_bias=False, M=Nf * h * w)[0].reshape(
            sd[pre + ".conv1.conv.weight"].shape)
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
        else:
            dx = B_.eltwise(dx, d_sc, B_.ADD)
        return dx
