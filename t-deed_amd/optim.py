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


BUCKET_ALIGN = 64      # elements: bucket boundaries of the flat buffer are multiples of this (see FlatParams)


def is_late_bucket_key(key):
    """Tensors whose gradients the backward completes first (temporal stack + heads): bucket 0 of the data-parallel
    reduction (trainer.TrainEngine.grad_buckets)."""
    return key.startswith(("_temp_fine.", "_pred_"))


class FlatParams:
    @staticmethod
    def _layout(keys, sizes):
        """-> (offsets, numel).  Every tensor starts on a 16-byte boundary so that per-tensor kernels can use vector
        accesses; the boundary between the trunk and the temporal stack (= the boundary between the two gradient buckets of
        a data-parallel job) and the end of the buffer sit on multiples of BUCKET_ALIGN elements, so that each bucket
        divides evenly over 2, 4 and 8 ranks with 16-byte-aligned shards: the reduce-scatter + all-gather form of the
        all-reduce wants that (with 4-element padding only, both buckets were 4 mod 8 and fell back to
        a plain all-reduce at 8 ranks)."""
        offs, cur = [], 0
        prev_late = None
        for k, n in zip(keys, sizes):
            late = is_late_bucket_key(k)
            if prev_late is not None and late != prev_late:
                cur = (cur + BUCKET_ALIGN - 1) // BUCKET_ALIGN * BUCKET_ALIGN
            prev_late = late
            offs.append(cur)
            cur += (n + 3) // 4 * 4
        cur = (cur + BUCKET_ALIGN - 1) // BUCKET_ALIGN * BUCKET_ALIGN
        return offs, cur

    @classmethod
    def layout_only(cls, state):
        """The index / numel this class would give `state` (name -> anything with .numel() / .shape), nothing allocated."""
        from types import SimpleNamespace
        keys = [k for k in state if state_layout.is_parameter(k)]
        sizes = [int(state[k].numel()) for k in keys]
        offs, numel = cls._layout(keys, sizes)
        return SimpleNamespace(numel=numel, index={k: (o, n) for k, o, n in zip(keys, offs, sizes)})

    def __init__(self, state, device=None):
        keys = [k for k in state if state_layout.is_parameter(k)]
        device = device if device is not None else state[keys[0]].device
        sizes = [int(state[k].numel()) for k in keys]
        offs, cur = self._layout(keys, sizes)
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
