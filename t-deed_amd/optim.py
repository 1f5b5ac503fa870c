"""Optimizer-side pieces of the training path.

``FlatParams`` keeps every trainable tensor of the model in ONE flat fp32 device buffer (the tensors of the
state_dict become views into it), so the fused AdamW kernel updates all 431 tensors in a single launch and the
same buffer is what a data-parallel job all-reduces (``dist.all_reduce_mean_``).  ``FusedAdamW`` mirrors
``BaseRGBModel.get_optimizer`` (/root/reference/model/modules.py:37-39: torch.optim.AdamW with default betas / eps /
weight decay 0.01 on *all* parameters), ``warmup_cosine_lr`` mirrors ``get_lr_scheduler``
(/root/reference/train_tdeed.py:79-87: LinearLR(0.01 -> 1) chained with CosineAnnealingLR, both stepping from 0).
`trainer.TrainEngine` fills the gradient buffer (train-mode forward + backward kernels) and drives these classes.
"""
import math

import torch

from . import ops, state_layout


class FlatParams:
    def __init__(self, state, device=None):
        keys = [k for k in state if state_layout.is_parameter(k)]
        device = device if device is not None else state[keys[0]].device
        sizes = [int(state[k].numel()) for k in keys]
        # every tensor starts on a 16-byte boundary so that per-tensor kernels can use vector accesses
        offs, cur = [], 0
        for n in sizes:
            offs.append(cur)
            cur += (n + 3) // 4 * 4
        self.numel = cur
        self.flat = torch.zeros(cur, dtype=torch.float32, device=device)
        self.grad = torch.zeros(cur, dtype=torch.float32, device=device)
        self.index, self.shapes = {}, {}
        for k, o, n in zip(keys, offs, sizes):
            view = self.flat[o:o + n].view(state[k].shape)
            view.copy_(state[k].to(device=device, dtype=torch.float32))
            state[k] = view                      # the state_dict entry now aliases the flat buffer
            self.index[k] = (o, n)
            self.shapes[k] = tuple(view.shape)
        # BatchNorm step counters (nn.BatchNorm.num_batches_tracked): one int64 vector, bumped by one launch per step
        nbt = [k for k in state if k.endswith("num_batches_tracked")]
        self.nbt = torch.zeros(max(len(nbt), 1), dtype=torch.int64, device=device)
        for i, k in enumerate(nbt):
            self.nbt[i] = state[k].to(device=device, dtype=torch.int64).reshape(())
            state[k] = self.nbt[i:i + 1].view(())

    def grad_view(self, key):
        o, n = self.index[key]
        return self.grad[o:o + n]


class FusedAdamW:
    def __init__(self, params: FlatParams, lr, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.01):
        self.p, self.lr, self.betas, self.eps, self.wd = params, lr, betas, eps, weight_decay
        self.exp_avg = torch.zeros_like(params.flat)
        self.exp_avg_sq = torch.zeros_like(params.flat)
        self.t = 0

    def zero_grad(self):
        self.p.grad.zero_()

    def step(self, lr_factor=1.0, grad_scale=1.0):
        self.t += 1
        ops.adamw_step(self.p.flat, self.p.grad, self.exp_avg, self.exp_avg_sq, self.t, self.lr * lr_factor,
                       self.betas, self.eps, self.wd, grad_scale)


def warmup_cosine_lr(step, warmup_steps, cosine_steps, start_factor=0.01):
    """LR multiplier after `step` scheduler steps of ChainedScheduler([LinearLR, CosineAnnealingLR(T_max)])."""
    lin = 1.0 if warmup_steps <= 0 else start_factor + (1.0 - start_factor) * min(step, warmup_steps) / warmup_steps
    cos = 0.5 * (1.0 + math.cos(math.pi * step / cosine_steps)) if cosine_steps > 0 else 1.0
    return lin * cos
