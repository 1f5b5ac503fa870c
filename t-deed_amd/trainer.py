"""One optimisation step of T-DEED on the HIP kernels: train-mode forward, loss, backward and fused AdamW, the
counterpart of the training branch of `TDEEDModel.epoch` (/root/reference/model/model.py:193-263) and
`BaseRGBModel.step` / `get_optimizer` (model/modules.py:37-39, 390-404) with the LR schedule of
train_tdeed.py:79-87.

    uint8 clip (B,T,3,H,W) --crop/flip/standardise + stem conv--> BN(batch stats)+ReLU --> 13/14 bottlenecks
    (trunk_train.py; gate-shift on s3/s4) --> avg-pool + temp_enc --> SGP encoder-decoder + heads + loss
    (temporal_train.py) --> the same chain backwards --> gradients in ONE flat fp32 buffer --> AdamW in one launch.

Every numeric step is a HIP kernel behind the C ABI and is parity-tested against torch autograd on the CPU oracle
(tests/test_gpu_bwd.py).  `build_graph/step_graph` capture weight re-packing + forward + loss + backward + gradient
write-out into one HIP graph (AdamW stays outside: lr and the step count change every step).  The step is correct but only
lightly tuned: DESIGN.md section 8 has the per-kernel breakdown.  Both gate-shift variants train: `_gsf` (every shipped
config) and the optional `_gsm` (impl/gsm.py; the same backward kernels without the fusion-conv path)."""
import os

import torch

from . import ops, ops_bwd as B_, _lib
from .streams import new_stream
from .optim import FlatParams, FusedAdamW, warmup_cosine_lr, is_late_bucket_key
from .regnet_spec import regnet_spec
from .temporal_train import TemporalStack
from .trunk_train import BottleneckTrain, StemTrain, BN_EPS

STEM_MFMA = True
STEM_BN_FUSED = True


class TrainEngine:
    """state: name -> tensor (reference state_dict names; fp32 parameters, BN buffers).  The parameters are moved into one
    flat buffer (`FlatParams`), `state` afterwards holds views into it."""

    def __init__(self, cfg, state, act_dtype=torch.bfloat16, device="cuda", lr=1e-3, weight_decay=0.01):
        self.cfg, self.dt, self.device = dict(cfg), act_dtype, device
        for k in list(state):                                    # in place: the caller's dict ends up holding the views
            v = state[k]
            state[k] = (v if isinstance(v, torch.Tensor) else torch.as_tensor(v)).to(device)
        self.state = state
        self.params = FlatParams(self.state, device)
        self.opt = FusedAdamW(self.params, lr, weight_decay=weight_decay)
        self.spec = regnet_spec(cfg["feature_arch"])
        self.T = cfg["clip_len"]
        sd = self.state
        self.blocks = [BottleneckTrain(sd, "_features." + b.name, b, act_dtype, clip_len=self.T) for b in self.spec.blocks]
        self.temporal = TemporalStack(sd, self.cfg, act_dtype)
        self.stem = StemTrain(sd, act_dtype)
        self.one32 = torch.ones(32, device=device)
        self.zero32 = torch.zeros(32, device=device)
        self.sched_step = 0
        self.reducer = None                                      # dist.GradReducer of a data-parallel job (set_reducer)
        # kernel-layout weight copies through recorded index tables (TDEED_REPACK_GATHER=0: every module re-packs itself
        # with its own casts / transposes / gathers, ~350 small launches per step)
        import os
        self.pack = None
        self.pack_gather = os.environ.get("TDEED_REPACK_GATHER", "1") == "1"
        if self.pack_gather:
            self.repack()

    # ------------------------------------------------------------------ data parallel
    def grad_buckets(self):
        """Element ranges of the flat gradient buffer in the order the backward completes them: [temporal stack + heads]
        (everything from the first `_temp_fine.` tensor to the end of the buffer), then [temp_enc + trunk]."""
        idx = self.params.index
        first = min(o for k, (o, n) in idx.items() if is_late_bucket_key(k))
        stray = [k for k, (o, n) in idx.items() if (o >= first) != is_late_bucket_key(k)]
        if stray:
            # the buckets are element RANGES reduced as soon as backward_heads / backward_trunk have written them: a state
            # dict whose trunk tensors sit behind the temporal stack would have them reduced before they are written
            raise RuntimeError(f"state_dict order: trunk and temporal-stack tensors interleave in the flat buffer ({stray[:3]} ...)")
        return [(first, self.params.numel), (0, first)]

    def _check_bucket_keys(self, keys, role):
        """Every gradient the write-out of bucket `role` carries must lie inside that bucket's element range (the range is
        all-reduced right after the write-out)."""
        bad = [k for k in keys if is_late_bucket_key(k) != (role == 0)]
        if bad:
            raise RuntimeError(f"gradient of {bad[0]} was produced by the {'temporal' if role == 0 else 'trunk'} backward "
                               f"but belongs to the other gradient bucket")

    def set_reducer(self, reducer="auto"):
        """Attach the gradient reducer of a data-parallel job (one process per GPU).  "auto": a dist.GradReducer over this
        engine's flat gradient buffer when torch.distributed is initialised with more than one rank, else none."""
        from . import dist as tdist
        import torch.distributed as td
        if reducer == "auto":
            reducer = None
            if td.is_available() and td.is_initialized() and td.get_world_size() > 1:
                reducer = tdist.GradReducer(self.params.grad, self.grad_buckets(), device=torch.device(self.device))
        self.reducer = reducer
        return reducer

    # ------------------------------------------------------------------ forward + backward
    def _frame_flip(self, flip, Bn, T):
        """bool (all frames) or per-clip flags (B,) -> what ops.stem takes: bool or per-frame uint8 flags (B*T,)."""
        if isinstance(flip, torch.Tensor):
            if flip.numel() != Bn:
                raise ValueError(f"flip: one flag per clip expected ({flip.numel()} vs {Bn})")
            return flip.to(self.device).to(torch.uint8).repeat_interleave(T).contiguous()
        return bool(flip)

    def forward_train(self, frames_u8, crop=None, flip=False, drop_masks=None):
        """Train-mode forward (`Impl.forward` under .train(): batch-statistics BatchNorm with running-stat updates, dropout
        through the given keep-masks, model.py:105-149).  frames (B,T,3,H,W) uint8 (or fp32 0..255) on the device;
        crop = (top, left, h, w) shared by all clips (model.py:115); flip: bool or per-clip flags (B,) (model.py:83).
        Returns (head_out (B*T, n_out) fp32, ctx for backward_train)."""
        sd, dt = self.state, self.dt
        _lib.SCOPE = "stem.fwd"
        Bn, T = frames_u8.shape[:2]
        fr = frames_u8.reshape(Bn * T, *frames_u8.shape[2:])
        fl = self._frame_flip(flip, Bn, T)
        H_, W_ = fr.shape[-2:]
        chw = (crop[2], crop[3]) if crop is not None else (H_, W_)
        if self.stem.wf is not None and STEM_MFMA and ops.stem_mfma_parts(*chw) > 0:
            # bf16: conv on the MFMA pipe, BatchNorm statistics from its epilogue (no second pass over the 112^2 map)
            z0, cp = ops.stem_mfma(fr, self.stem.wf, crop=crop, flip=fl)
            cpf = cp.view(-1)
            y0, bn0 = B_.bn_finalize_apply(z0, cpf, cpf[32:], 64, cp.shape[0], sd["_features.stem.bn.weight"],
                                           sd["_features.stem.bn.bias"], BN_EPS, 0.1, sd["_features.stem.bn.running_mean"],
                                           sd["_features.stem.bn.running_var"], relu=True)
        else:
            z0 = ops.stem(fr, sd["_features.stem.conv.weight"], self.one32, self.zero32, dt, crop=crop, flip=fl, relu=False)
            y0, bn0 = B_.bn_train(z0, sd["_features.stem.bn.weight"], sd["_features.stem.bn.bias"], BN_EPS, 0.1,
                                  sd["_features.stem.bn.running_mean"], sd["_features.stem.bn.running_var"], relu=True)
        x = y0
        xs = None
        for i, blk in enumerate(self.blocks):
            _lib.SCOPE = blk.blk.name + ".fwd"
            nxt = self.blocks[i + 1] if i + 1 < len(self.blocks) else None
            x = blk.forward(x, xs=xs, next_fold=(nxt.blk.gsf_fold if nxt is not None and nxt.gs is not None else 0))
            xs = blk.ctx.out_slice
        _lib.SCOPE = "temporal.fwd"
        hw = x.shape[1] * x.shape[2]
        feat = ops.avgpool_posenc(x, Bn, T, sd["temp_enc"])
        head_out, tctx = self.temporal.forward_heads(feat, drop_masks)
        self.params.nbt += 1                                     # BatchNorm step counters (nn.BatchNorm.num_batches_tracked)
        from types import SimpleNamespace
        return head_out, SimpleNamespace(fr=fr, crop=crop, flip=fl, z0=z0, y0=y0, bn0=bn0, x_shape=x.shape, hw=hw, tctx=tctx,
                                         B=Bn, T=T)

    def backward_train(self, ctx, dhead):
        """Backward of forward_train from d(loss)/d(head_out): returns grads dict name -> fp32 tensor."""
        grads = {}
        _lib.SCOPE = "temporal.bwd"
        d_feat = self.temporal.backward_heads(ctx.tctx, dhead, grads)
        grads.update(self.backward_trunk(ctx, d_feat))
        return {k: B_.materialize(g) for k, g in grads.items()}

    def backward_trunk(self, ctx, d_feat):
        """Backward of avg-pool + positional encoding, the bottlenecks and the stem from d(loss)/d(features)."""
        sd = self.state
        grads = {}
        dx, d_enc = B_.avgpool_posenc_bwd(d_feat, ctx.hw)
        grads["temp_enc"] = d_enc
        dx = dx.view(ctx.x_shape)
        from .trunk_train import ZMASK, SINK
        nb = len(self.blocks)
        sinks = [blk.make_sink() for blk in self.blocks[:-1]] if SINK else []
        stem_sink = B_.GradSink(ctx.y0, ctx.z0, ctx.bn0[0]) if SINK else None
        for i in reversed(range(nb)):
            blk = self.blocks[i]
            _lib.SCOPE = blk.blk.name + ".bwd"
            if SINK:
                # block i's input gradient is produced masked by the ReLU in front of it, with the column sums the BatchNorm
                # backward of block i-1 (or of the stem) needs
                dx = blk.backward(dx, grads, sink_in=(sinks[i] if i + 1 < nb else None),
                                  sink_out=(sinks[i - 1] if i > 0 else stem_sink))
            else:
                dx = blk.backward(dx, grads)
        _lib.SCOPE = "stem.bwd"
        wbn = sd["_features.stem.bn.weight"]
        fr4 = ctx.fr.view(-1, *ctx.fr.shape[-3:])
        if SINK and STEM_BN_FUSED and B_.stem_wgrad_bn_fits(fr4, ctx.crop, dx.dtype):
            # the stem BatchNorm's backward is applied inside the weight-gradient launch while it stages the gradient rows:
            # dz0 (the largest map of the step) is neither written nor read back
            sums = B_.bn_sums_from_sink(ctx.z0, dx, ctx.bn0, wbn, stem_sink, q=1)
            grads["_features.stem.bn.weight"], grads["_features.stem.bn.bias"] = sums[1], sums[0]
            grads["_features.stem.conv.weight"] = B_.stem_wgrad(ctx.fr, dx, crop=ctx.crop, flip=ctx.flip,
                                                                bn=(ctx.z0, sums, ctx.bn0[0], ctx.bn0[1], wbn))
            return grads
        if SINK:
            dz0, dw, db = B_.bn_bwd_from_parts(ctx.z0, dx, ctx.bn0, wbn, stem_sink, q=1)
        else:
            dz0, _, dw, db = B_.bn_train_bwd(ctx.z0, dx, None if ZMASK else ctx.y0, ctx.bn0, wbn, relu=True)
        grads["_features.stem.bn.weight"], grads["_features.stem.bn.bias"] = dw, db
        grads["_features.stem.conv.weight"] = B_.stem_wgrad(ctx.fr, dz0, crop=ctx.crop, flip=ctx.flip)
        return grads

    def loss_and_grads(self, frames_u8, label, labelD=None, soft=None, crop=None, flip=False, drop_masks=None,
                       fg_weight=5.0, dataset=None):
        """frames (B,T,3,H,W) uint8 on the device; label int64 (B,T) or None with soft (B,T,K+1); labelD float (B,T).
        crop = (top, left, h, w) (the one random crop the reference shares across B and T, model.py:115) or None.
        Returns (loss[3] = total, ce, mse ; grads dict name -> fp32 tensor)."""
        head_out, ctx = self.forward_train(frames_u8, crop, flip, drop_masks)
        loss, dhead = self.temporal.loss_fwd_bwd(
            head_out, ctx.B, ctx.T, None if label is None else label.reshape(-1).contiguous(),
            labelD=None if labelD is None else labelD.reshape(-1).float().contiguous(),
            soft=None if soft is None else soft.reshape(-1, soft.shape[-1]).contiguous(), fg_weight=fg_weight,
            dataset=dataset)
        return loss, self.backward_train(ctx, dhead)

    def repack(self):
        """Refresh the kernel-layout copies of the weights (after an optimizer step or a load_state_dict): one gather
        launch per dtype through the recorded index tables (repack.PackPlan)."""
        if not self.pack_gather:
            for blk in self.blocks:
                blk.repack()
            self.temporal.repack()
            self.stem.repack()
        elif self.pack is None:
            from .repack import PackPlan
            self.pack = PackPlan(self.params, self.device).build(list(self.blocks) + [self.temporal, self.stem])
        else:
            self.pack.run()
            for blk in self.blocks:          # the gather refreshed every block's packed copies in place (blk.repack() bumps
                blk.pack_gen = getattr(blk, "pack_gen", 0) + 1   # its own counter): a backward of an older forward must not recompute from them

    # ------------------------------------------------------------------ one optimiser step
    def write_grads(self, grads, scale=1.0, first=True, partial=False, role=0):
        """Gradient write-out into the flat buffer (times `scale`; overwriting when `first`, adding otherwise:
        `acc_grad_iter` of the reference's step()): multi-tensor copies, a handful of launches instead of one per tensor."""
        if not partial:
            missing = set(self.params.index) - set(grads)
            if missing:
                raise RuntimeError(f"no gradient produced for {sorted(missing)[:4]} ...")
        keys = list(grads)
        srcs = [grads[k] for k in keys]                          # tensors (contiguous or column slices) or LazyFold partials
        for k, g in zip(keys, srcs):
            if g.numel() != self.params.index[k][1] or g.dtype != torch.float32:
                raise RuntimeError(f"gradient of {k}: {tuple(g.shape)} {g.dtype} does not match the parameter")
        # one launch for all tensors (a per-tensor device copy is ~3 us and a step has 431 of them)
        if not hasattr(self, "_grad_tabs"):
            self._grad_tabs = B_.PinnedTables(max_entries=len(self.params.index) + 8)
        B_.multi_copy(srcs, [self.params.index[k][0] for k in keys], self.params.grad, scale=scale, accumulate=not first,
                      tables=self._grad_tabs, role=role)

    def accumulate(self, frames_u8, label, labelD=None, soft=None, crop=None, flip=False, drop_masks=None, scale=1.0,
                   first=True, dataset=None, fg_weight=5.0, reduce=False):
        """forward + backward of one (micro-)batch; its gradients (times `scale`) go into the flat gradient buffer.
        reduce (data parallel, last micro-batch of a step): each bucket's all-reduce is enqueued as soon as the backward has
        produced it -- the temporal stack + heads travel over xGMI while the trunk backward runs."""
        head_out, ctx = self.forward_train(frames_u8, crop, flip, drop_masks)
        loss, dhead = self.temporal.loss_fwd_bwd(
            head_out, ctx.B, ctx.T, None if label is None else label.reshape(-1).contiguous(),
            labelD=None if labelD is None else labelD.reshape(-1).float().contiguous(),
            soft=None if soft is None else soft.reshape(-1, soft.shape[-1]).contiguous(), fg_weight=fg_weight,
            dataset=dataset)
        self.backward_and_write(ctx, dhead, scale, first, reduce)
        return loss

    def backward_and_write(self, ctx, dhead, scale=1.0, first=True, reduce=False):
        """Backward of forward_train from d(loss)/d(head_out) with the gradients (times `scale`) written into the flat
        buffer bucket by bucket; with `reduce` each bucket's all-reduce starts as soon as it is complete."""
        red = self.reducer if reduce else None
        g_t = {}
        # weight gradients stay in their per-workgroup partials until the bucket's write-out launch folds them
        lazy = os.environ.get("TDEED_LAZY_WGRAD", "1") == "1"
        B_.LAZY_WGRAD = lazy
        _lib.SCOPE = "temporal.bwd"
        try:
            d_feat = self.temporal.backward_heads(ctx.tctx, dhead, g_t)
        finally:
            B_.LAZY_WGRAD = False
        self._check_bucket_keys(g_t, 0)
        _lib.SCOPE = "write_grads"
        self.write_grads(g_t, scale, first, partial=True, role=0)
        if red is not None:
            red.reduce_bucket(0)
        B_.LAZY_WGRAD = lazy
        try:
            g_b = self.backward_trunk(ctx, d_feat)
        finally:
            B_.LAZY_WGRAD = False
        missing = set(self.params.index) - set(g_t) - set(g_b)
        if missing:
            raise RuntimeError(f"no gradient produced for {sorted(missing)[:4]} ...")
        self._check_bucket_keys(g_b, 1)
        _lib.SCOPE = "write_grads"
        self.write_grads(g_b, scale, first, partial=True, role=1)
        if red is not None:
            red.reduce_bucket(1)
        self._reduced = red is not None                          # apply() checks it (scale only what was reduced)

    def apply(self, lr=None, lr_factor=1.0, all_reduce=None):
        """AdamW on the accumulated gradients (one launch), then refresh the kernels' views of the weights.  With a reducer
        attached the current stream first waits for the bucket all-reduces and the 1/world of the mean rides on AdamW's
        grad_scale; `all_reduce` is the older blocking form (callable(flat_grad))."""
        gs = 1.0
        _lib.SCOPE = "optim"
        if all_reduce is not None:
            all_reduce(self.params.grad)
        elif self.reducer is not None:
            if not getattr(self, "_reduced", False):
                # the step's gradients were accumulated without `reduce=True` (e.g. accumulate() called directly): sum them
                # over the ranks now, blocking on nothing but the device -- never scale un-reduced gradients by 1/world
                self.reducer.reduce_all()
            self.reducer.join()
            gs = self.reducer.scale
        self._reduced = False
        if lr is not None:
            self.opt.lr = lr
        self.opt.step(lr_factor=lr_factor, grad_scale=gs)
        self.repack()

    def step(self, frames_u8, label, labelD=None, soft=None, crop=None, flip=False, drop_masks=None, lr_factor=1.0,
             all_reduce=None):
        """forward + backward + (data parallel: bucketed gradient all-reduce overlapped with the backward) + AdamW.
        all_reduce: the older blocking form, a callable(flat_grad) (dist.all_reduce_mean_).  Returns [total, ce, mse]."""
        loss = self.accumulate(frames_u8, label, labelD, soft, crop, flip, drop_masks,
                               reduce=self.reducer is not None and all_reduce is None)
        self.apply(lr_factor=lr_factor, all_reduce=all_reduce)
        return loss

    # ------------------------------------------------------------------ the step as one HIP graph
    def build_graph(self, B, H, W, frames_dtype=torch.uint8, with_labelD=None, soft=False, fg_weight=5.0):
        """Capture weight re-packing + train-mode forward + loss + backward + gradient write-out for one batch geometry
        into a HIP graph (the eager step is host-bound: ~1700 launches).  Inputs live in static buffers
        (`.frames/.label/.labelD/.soft/.masks` of the returned handle); the AdamW launch stays outside the graph because lr
        and the step count change every step.  torch's graph-private memory pool keeps every temporary of the step alive
        between replays.
        Data parallel (default, `mode == "two"`): the step is captured as TWO graphs -- [re-pack, forward, loss, temporal
        backward, bucket 0 write-out] and [trunk backward, bucket 1 write-out] -- and bucket 0's all-reduce is launched
        between the two replays on the communicator's own stream, so it travels while the trunk backward runs.  With
        TDEED_DP_IN_GRAPH=1 and the C-ABI RCCL communicator the all-reduces are captured into ONE graph instead (its stream
        forks from / joins the captured stream through events, `mode == "one"`); if that capture fails the two-graph form is
        built."""
        import os
        import sys
        from types import SimpleNamespace
        dev, T = self.device, self.T
        K1 = self.cfg["num_classes"] + 1
        radi = self.cfg.get("radi_displacement", 0)
        if with_labelD is None:
            with_labelD = radi > 0
        C = self.spec.feat_dim
        h = SimpleNamespace(B=B)
        h.frames = torch.zeros((B, T, 3, H, W), dtype=frames_dtype, device=dev)
        h.label = None if soft else torch.zeros((B, T), dtype=torch.int64, device=dev)
        h.soft = torch.full((B, T, K1), 1.0 / K1, dtype=torch.float32, device=dev) if soft else None
        h.labelD = torch.zeros((B, T), dtype=torch.float32, device=dev) if with_labelD else None
        h.masks = [torch.ones((B, T, C), dtype=self.dt, device=dev) for _ in range(2 if radi > 0 else 1)]
        red = self.reducer
        # the gradient write-out launches read their record tables from pinned host memory at every replay: the graph gets
        # tables of its own (the eager steps' ring and other graphs of this engine must never rewrite them)
        eager_tabs = getattr(self, "_grad_tabs", None)
        self._grad_tabs = B_.PinnedTables(max_entries=len(self.params.index) + 8, depth=1)
        # default: two graphs with the collectives launched eagerly between them -- the same overlap, and nothing depends
        # on the collective library's behaviour under stream capture (which no single-GPU box can exercise);
        # TDEED_DP_IN_GRAPH=1 captures the RCCL calls into one graph instead
        want_in_graph = red is not None and red.capturable and os.environ.get("TDEED_DP_IN_GRAPH", "0") == "1"
        keep = {k: v.clone() for k, v in self.state.items() if k.endswith(("running_mean", "running_var", "num_batches_tracked"))}

        def restore():
            for k, v in keep.items():
                self.state[k].copy_(v)

        def part_a():
            head_out, ctx = self.forward_train(h.frames, None, False, h.masks)
            loss, dhead = self.temporal.loss_fwd_bwd(
                head_out, ctx.B, ctx.T, None if h.label is None else h.label.reshape(-1),
                labelD=None if h.labelD is None else h.labelD.reshape(-1),
                soft=None if h.soft is None else h.soft.reshape(-1, h.soft.shape[-1]), fg_weight=fg_weight)
            g_t = {}
            B_.LAZY_WGRAD = os.environ.get("TDEED_LAZY_WGRAD", "1") == "1"
            try:
                d_feat = self.temporal.backward_heads(ctx.tctx, dhead, g_t)
            finally:
                B_.LAZY_WGRAD = False
            self._check_bucket_keys(g_t, 0)
            self.write_grads(g_t, 1.0, True, partial=True, role=0)
            return loss, ctx, d_feat, set(g_t)

        def part_b(ctx, d_feat, done):
            B_.LAZY_WGRAD = os.environ.get("TDEED_LAZY_WGRAD", "1") == "1"
            try:
                g_b = self.backward_trunk(ctx, d_feat)
            finally:
                B_.LAZY_WGRAD = False
            missing = set(self.params.index) - done - set(g_b)
            if missing:
                raise RuntimeError(f"no gradient produced for {sorted(missing)[:4]} ...")
            self._check_bucket_keys(g_b, 1)
            self.write_grads(g_b, 1.0, True, partial=True, role=1)

        def capture_one(in_graph):
            run = lambda: self.accumulate(h.frames, h.label, h.labelD, soft=h.soft, drop_masks=h.masks,      # noqa: E731
                                          reduce=in_graph, fg_weight=fg_weight)
            # eager warm-up on a side stream (lazy kernel attributes / module loads must happen outside the capture); the
            # BatchNorm buffers it touches are restored afterwards
            side = new_stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                self.repack()
                run()
                if in_graph:
                    red.join()
            torch.cuda.current_stream().wait_stream(side)
            torch.cuda.synchronize()
            restore()
            g = torch.cuda.CUDAGraph()
            # thread-local capture: other threads (the RCCL watchdog of a data-parallel job) may keep calling the runtime
            with torch.cuda.graph(g, stream=side, capture_error_mode="thread_local"):
                self.repack()
                h.loss = run()
                if in_graph:
                    red.join()
            h.graph, h.graph_b, h.mode, h.reduce_in_graph = g, None, "one", in_graph

        def capture_two():
            side = new_stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                self.repack()
                _, ctx, d_feat, done = part_a()
                red.reduce_bucket(0)
                part_b(ctx, d_feat, done)
                red.reduce_bucket(1)
                red.join()
            torch.cuda.current_stream().wait_stream(side)
            torch.cuda.synchronize()
            restore()
            ga, gb = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
            with torch.cuda.graph(ga, stream=side, capture_error_mode="thread_local"):
                self.repack()
                h.loss, ctx, d_feat, done = part_a()
            with torch.cuda.graph(gb, pool=ga.pool(), stream=side, capture_error_mode="thread_local"):
                part_b(ctx, d_feat, done)
            h.keep_ctx = (ctx, d_feat)                           # activations of the first half that the second half reads
            h.graph, h.graph_b, h.mode, h.reduce_in_graph = ga, gb, "two", False

        if red is None:
            capture_one(False)
        elif want_in_graph:
            try:
                capture_one(True)
            except Exception as e:                               # noqa: BLE001  (a runtime that cannot capture the collectives)
                print(f"[tdeed_amd] capturing the RCCL collectives failed ({type(e).__name__}: {e}); "
                      f"falling back to two graphs with the all-reduces launched between them", file=sys.stderr, flush=True)
                torch.cuda.synchronize()
                capture_two()
        else:
            capture_two()
        h.keep = self._grad_tabs                                 # the gradient write-out tables the graph's copies read
        if eager_tabs is not None:
            self._grad_tabs = eager_tabs
        else:
            del self._grad_tabs
        self._reduced = False                                    # (the warm-up pass of an in-graph capture reduced)
        restore()                                                # capture does not execute, but stay explicit
        torch.cuda.synchronize()
        return h

    def step_graph(self, h, frames, label, labelD=None, soft=None, drop_masks=None, lr=None, lr_factor=1.0, all_reduce=None):
        """One optimisation step through a captured graph: refresh the static inputs, replay, all-reduce (DP), AdamW."""
        h.frames.copy_(frames, non_blocking=True)
        if h.label is not None:
            h.label.copy_(label, non_blocking=True)
        if h.soft is not None:
            h.soft.copy_(soft, non_blocking=True)
        if h.labelD is not None:
            h.labelD.copy_(labelD, non_blocking=True)
        if drop_masks is not None:
            for dst, src in zip(h.masks, drop_masks):
                dst.copy_(src, non_blocking=True)
        h.graph.replay()
        gs = 1.0
        if h.mode == "two":
            # bucket 0 (temporal stack + heads) is complete: its all-reduce travels while the trunk backward replays
            if all_reduce is None:
                self.reducer.reduce_bucket(0)
            h.graph_b.replay()
            if all_reduce is None:
                self.reducer.reduce_bucket(1)
                self.reducer.join()
                gs = self.reducer.scale
        elif all_reduce is None and self.reducer is not None:
            if not h.reduce_in_graph:
                # a graph captured before set_reducer(): its gradients are local, reduce them here
                self.reducer.reduce_all()
                self.reducer.join()
            gs = self.reducer.scale                             # (else: reduced and joined inside the graph)
        if all_reduce is not None:
            all_reduce(self.params.grad)
        if lr is not None:
            self.opt.lr = lr
        self.opt.step(lr_factor=lr_factor, grad_scale=gs)       # the next replay starts with repack(): no refresh needed here
        return h.loss

    def make_step(self, B, H, W, frames, label, labelD=None, drop_masks=None, use_graph=True, world=1, all_reduce=None):
        """A zero-argument callable running one optimisation step on the given (static) batch: through the captured HIP graph
        or eagerly; for world > 1 the flat gradient buffer is all-reduced (RCCL) before AdamW.  all_reduce: a callable
        (flat_grad) replacing the bucketed reduction (the bench's "no communication" arm passes a no-op)."""
        ar = all_reduce
        if world > 1 and self.reducer is None:
            self.set_reducer("auto")
        if not use_graph:
            return lambda: self.step(frames, label, labelD, drop_masks=drop_masks, all_reduce=ar)
        hnd = getattr(self, "last_graph", None)
        if hnd is None or getattr(hnd, "geom", None) != (B, H, W):
            hnd = self.build_graph(B, H, W)
            hnd.geom = (B, H, W)
        self.last_graph = hnd                                    # (tests / the bench line read .mode)
        return lambda: self.step_graph(hnd, frames, label, labelD, drop_masks=drop_masks, all_reduce=ar)

    def lr_factor(self, warmup_steps, cosine_steps):
        f = warmup_cosine_lr(self.sched_step, warmup_steps, cosine_steps)
        self.sched_step += 1
        return f


class HipAdamW(torch.optim.Optimizer):
    """What `TDEEDModel.get_optimizer` returns: a torch.optim.Optimizer whose single "parameter" is the flat fp32
    buffer and whose step() is the fused AdamW launch, so the reference's LR schedulers (LinearLR + CosineAnnealingLR
    chained, train_tdeed.py:79-87) drive it unchanged through param_groups[0]['lr']."""

    def __init__(self, engine: TrainEngine, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.01):
        self.engine = engine
        engine.opt.lr, engine.opt.betas, engine.opt.eps, engine.opt.wd = lr, betas, eps, weight_decay
        super().__init__([engine.params.flat], dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))

    def zero_grad(self, set_to_none=False):
        self.engine.opt.zero_grad()

    @torch.no_grad()
    def step(self, closure=None, all_reduce=None):
        g = self.param_groups[0]
        o = self.engine.opt
        o.betas, o.eps, o.wd = g["betas"], g["eps"], g["weight_decay"]
        self.engine.apply(lr=g["lr"], all_reduce=all_reduce)
