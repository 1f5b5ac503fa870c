"""Drop-in counterpart of the reference's ``model/model.py`` API, backed by the HIP engine.

Same constructor arguments, attributes and method signatures as
``TDEEDModel`` / ``TDEEDModel.Impl`` (/root/reference/model/model.py:21-376) and
``BaseRGBModel`` (/root/reference/model/modules.py:35-55), so ``train_tdeed.py:142-148``,
``util/eval.py:301-339`` and ``evaluate_tdeed_challenge.py`` can import this module
instead.  There is exactly one execution backend (the gfx950 kernels); without the
built library or without a GPU every compute call raises.
"""
from types import SimpleNamespace

import numpy as np
import torch

from . import ops, state_layout, augment
from . import init as ref_init
from .engine import ForwardEngine
from .regnet_spec import regnet_spec


def _cfg_from_args(args):
    crop = getattr(args, "crop_dim", None)
    if crop is not None and crop <= 0:      # train_tdeed.py:110-111
        crop = None
    return dict(feature_arch=args.feature_arch, clip_len=args.clip_len, crop_dim=crop,
                n_layers=args.n_layers, sgp_ks=args.sgp_ks, sgp_r=args.sgp_r,
                num_classes=args.num_classes, radi_displacement=args.radi_displacement)


class TDEEDModel:

    class Impl:
        """The network.  Holds the fp32 master state in the reference's key grammar."""

        def __init__(self, args=None, seed=None):
            self._modality = args.modality
            assert self._modality == "rgb", "Only RGB supported for now"
            self._temp_arch = args.temporal_arch
            assert self._temp_arch in ["ed_sgp_mixer"], "Only ed_sgp_mixer supported for now"
            self._radi_displacement = args.radi_displacement
            self._feature_arch = args.feature_arch
            assert "rny" in self._feature_arch, "Only rny supported for now"
            if not self._feature_arch.startswith(("rny002", "rny008")):
                raise NotImplementedError(self._feature_arch)
            self._double_head = False
            self._cfg = _cfg_from_args(args)
            self._spec = regnet_spec(self._feature_arch)
            self._d = self._feat_dim = self._spec.feat_dim
            self._require_clip_len = args.clip_len if self._feature_arch.endswith(("_gsm", "_gsf")) else -1
            self.croping = self._cfg["crop_dim"]
            self._head_classes = None
            # The reference's construction-time state (init.py: timm's RegNet init with zero_init_last, BatchNorm identity
            # statistics, temp_enc ~ N(0, 1/L), depthwise convs N(0, 0.1) with zero biases, torch defaults elsewhere;
            # model.py:38-70, modules.py:146-157, 255-275).  There is no network here, so the ImageNet weights that
            # `pretrained=True` (model.py:38-41) downloads arrive through load_timm_backbone() / load().
            # seed: None draws from torch's global CPU generator like the reference's constructors do.
            shapes = state_layout.model_state_shapes(self._cfg)
            gen = None if seed is None else torch.Generator().manual_seed(int(seed))
            self._state = ref_init.reference_init(shapes, self._cfg, gen)
            self._device = "cpu"
            self.training = False
            self._engines = {}
            self._train_engine = None               # trainer.TrainEngine over self._state (get_optimizer / first train-mode call)
            self._train_dtype = torch.bfloat16      # the reference trains under autocast; torch.float32 for parity runs
            self._train_ctx = None                  # activations of the last train-mode forward (for the backward)
            self.augment_fn = None                  # optional hook replacing the built-in train-time augmentation:
            #   augment_fn(frames (B,T,3,H,W) uint8|fp32 0..255 on the device, crop (top,left,h,w)|None) -> frames of the
            #   crop window (B,T,3,h,w), uint8 or fp32 0..255, on the device (flips included, if wanted)
            self.augment_generator = None           # torch.Generator for the augmentation draws (None: the global CPU one)
            self.dropout_mask_fn = None             # optional hook replaying a recorded dropout draw of the heads:
            #   dropout_mask_fn(B, T, C, n_heads) -> n_heads tensors (B,T,C) of 0 / 2 (class head(s) first, displacement last)

        # ---- nn.Module-like surface the reference's callers touch
        @staticmethod
        def _norm_device(device):
            d = torch.device(device)
            if d.type == "cuda" and d.index is None and torch.cuda.is_available():
                d = torch.device("cuda", torch.cuda.current_device())
            return d

        def to(self, device):
            if self._train_engine is not None and self._norm_device(device) != self._norm_device(self._device):
                raise RuntimeError("the model cannot change device once its training engine exists")
            self._device = str(device)
            for k in list(self._state):          # in place: trainer / optimizer may hold this dict
                self._state[k] = self._state[k].to(device)
            self._engines = {}
            return self

        def cuda(self):
            return self.to("cuda")

        def train(self, mode=True):
            self.training = mode
            return self

        def eval(self):
            return self.train(False)

        def state_dict(self):
            return dict(self._state)

        def load_state_dict(self, sd, strict=True):
            missing = [k for k in self._state if k not in sd]
            extra = [k for k in sd if k not in self._state]
            if strict and (missing or extra):
                raise RuntimeError(f"state_dict mismatch: missing {missing[:5]} unexpected {extra[:5]}")
            for k, v in sd.items():
                if k in self._state:
                    v = torch.as_tensor(v)
                    if tuple(v.shape) != tuple(self._state[k].shape):
                        raise RuntimeError(f"shape mismatch for {k}: {tuple(v.shape)} vs {tuple(self._state[k].shape)}")
                    # in place: after get_optimizer() the parameters are views into one flat buffer that the fused
                    # optimizer and the train engine hold on to
                    self._state[k].copy_(v.detach().to(self._state[k].dtype).to(self._device))
            self._engines = {}
            if self._train_engine is not None:
                # the train engine keeps packed copies (bf16 casts, transposes, MFMA fragments) of the master weights
                self._train_engine.repack()

        def load_timm_backbone(self, timm_state_dict):
            """Fill the trunk from a timm `regnety_002` / `regnety_008` state_dict (what `timm.create_model(...,
            pretrained=True)` holds at model.py:38-41): `stem.* / s{1..4}.b{n}.*` -> `_features.*`, with `conv1.*` ->
            `conv1.net.*` on the gate-shift stages s3 / s4 (shift.py:46-59); `head.fc.*` is dropped (model.py:45).  The
            gate-shift modules, temp_enc, the temporal stack and the heads keep their construction-time state.
            Returns the list of keys it filled."""
            mapped = ref_init.map_timm_backbone(timm_state_dict, self._cfg)
            self.load_state_dict(mapped, strict=False)
            return list(mapped)

        def parameters(self):
            return [v for k, v in self._state.items() if state_layout.is_parameter(k)]

        def update_pred_head(self, num_classes=[1, 1]):
            """model.py:169-172: replace the class head by two heads (joint-dataset training)."""
            C = self._feat_dim
            for k in [k for k in self._state if k.startswith("_pred_fine.")]:
                del self._state[k]
            shapes = {}
            for i, n in enumerate(num_classes, start=1):
                shapes[f"_pred_fine._fc{i}._fc_out.weight"] = ((n, C), "float32")
                shapes[f"_pred_fine._fc{i}._fc_out.bias"] = ((n,), "float32")
            new = {k: v.to(self._device) for k, v in ref_init.reference_init(shapes, self._cfg, None).items()}   # nn.Linear defaults
            # keep the reference's key order: heads sit before _pred_displ
            displ = {k: self._state.pop(k) for k in [k for k in self._state if k.startswith("_pred_displ.")]}
            self._state.update(new)
            self._state.update(displ)
            self._double_head = True
            self._head_classes = list(num_classes)
            self._engines = {}
            if self._train_engine is not None:
                raise RuntimeError("update_pred_head() must be called before get_optimizer() (train_tdeed.py:147-151 does)")

        def print_stats(self):
            def cnt(pfx):
                return sum(v.numel() for k, v in self._state.items()
                           if k.startswith(pfx) and state_layout.is_parameter(k))
            print("Model params:", cnt(""))
            print("  CNN features:", cnt("_features."))
            print("  Temporal:", cnt("_temp_fine."))
            print("  Head:", cnt("_pred_fine."))

        # ---- forward
        def engine(self, act_dtype):
            if act_dtype not in self._engines:
                if not str(self._device).startswith("cuda"):
                    raise RuntimeError("tdeed_amd runs on the GPU only: construct TDEEDModel(device='cuda')")
                self._engines[act_dtype] = ForwardEngine(self._cfg, self._state, act_dtype, self._device)
            return self._engines[act_dtype]

        def train_engine(self):
            """The training engine over this model's state (created by get_optimizer(), or lazily by the first
            train-mode forward): moves the parameters into one flat buffer, `self._state` then holds views into it."""
            if self._train_engine is None:
                if not str(self._device).startswith("cuda"):
                    raise RuntimeError("tdeed_amd runs on the GPU only: construct TDEEDModel(device='cuda')")
                from .trainer import TrainEngine
                self._train_engine = TrainEngine(self._cfg, self._state, act_dtype=self._train_dtype, device=self._device)
                self._engines = {}
            return self._train_engine

        def _pack_head(self, head, B, T, n_cls, displ_col, y):
            head = head.view(B, T, -1)
            im_feat = head[..., :n_cls]
            if self._radi_displacement > 0:
                return {"im_feat": im_feat, "displ_feat": head[..., displ_col], "_head_out": head}, y
            return im_feat, y

        def _forward_train(self, x, y, inference, augment_inference):
            """model.py:105-149 under .train(): batch-statistics BatchNorm (running stats updated), dropout in the heads;
            `inference` only selects the crop / augmentation branch (model.py:110-129) like in the reference."""
            import random
            eng = self.train_engine()
            if x.dtype not in (torch.uint8, torch.float32):
                x = x.float()
            x = x.to(self._device).contiguous()           # uint8, or fp32 0..255 (mixup batches / callers' .float())
            B, T, _, H, W = x.shape
            if self._require_clip_len > 0 and T != self._require_clip_len:
                raise ValueError(f"clip length {T} != clip_len {self._require_clip_len} (gate-shift needs exact clips)")
            cd = self.croping
            crop, flip = None, False
            if not inference:
                if cd and (cd != H or cd != W):
                    # torchvision RandomCrop.get_params: torch.randint for the row, then for the column; ONE window for
                    # the whole batch (model.py:115 crops the 5-D tensor)
                    g = self.augment_generator
                    top = int(torch.randint(0, H - cd + 1, size=(1,), generator=g).item())
                    left = int(torch.randint(0, W - cd + 1, size=(1,), generator=g).item())
                    crop = (top, left, cd, cd)
                if self.augment_fn is not None:
                    x = self.augment_fn(x, crop)
                    crop = None
                    if x.dtype not in (torch.uint8, torch.float32) or not x.is_cuda:
                        raise TypeError("augment_fn must return uint8 / float32 frames on the device")
                    x = x.contiguous()
                else:
                    prm, flip_c = augment.draw_params(B, self.augment_generator)
                    if not augment.is_identity(prm):
                        x = augment.apply(x, prm, crop)
                        crop = None
                    flip = flip_c.to(self._device) if bool(flip_c.any()) else False
            else:
                if cd and (cd != H or cd != W):
                    crop = (int(round((H - cd) / 2.0)), int(round((W - cd) / 2.0)), cd, cd)
                flip = bool(augment_inference)
            C = self._feat_dim
            n_heads = (2 if self._double_head else 1) + (1 if self._radi_displacement > 0 else 0)
            if self.dropout_mask_fn is not None:
                # replay of a recorded nn.Dropout draw (parity tests against reference fixtures): n_heads tensors (B,T,C)
                # holding 0 / 2, class head(s) first, displacement head last
                masks = [m.to(self._device).to(eng.dt).contiguous() for m in self.dropout_mask_fn(B, T, C, n_heads)]
            else:
                masks = [((torch.rand((B, T, C), device=self._device) >= 0.5).to(eng.dt) * 2.0) for _ in range(n_heads)]
            head, ctx = eng.forward_train(x, crop=crop, flip=flip, drop_masks=masks)
            self._train_ctx = ctx
            n_cls, dcol, _ = eng.temporal.head_layout()
            return self._pack_head(head, B, T, n_cls, dcol, y)

        def _forward_eval_augmented(self, x, y, act_dtype):
            """model.py:105-129 with `inference=False` on a module in eval() mode: the training branch's random crop (one
            window for the whole batch, model.py:115) and per-clip augmentation (model.py:76-83, 154-157) in front of
            running-statistics BatchNorm and no dropout.  No caller of the reference uses this pairing (epoch() pairs
            train() with inference=False and eval() with inference=True, model.py:196-203); it is here because the
            reference's nn.Module allows it.  Same RNG draw order as the train-mode forward."""
            if x.dtype not in (torch.uint8, torch.float32):
                x = x.float()
            x = x.to(self._device).contiguous()
            B, T, _, H, W = x.shape
            cd = self.croping
            crop = None
            if cd and (cd != H or cd != W):
                g = self.augment_generator
                top = int(torch.randint(0, H - cd + 1, size=(1,), generator=g).item())
                left = int(torch.randint(0, W - cd + 1, size=(1,), generator=g).item())
                crop = (top, left, cd, cd)
            flip_frames = None
            if self.augment_fn is not None:
                x = self.augment_fn(x, crop)
                if x.dtype not in (torch.uint8, torch.float32) or not x.is_cuda:
                    raise TypeError("augment_fn must return uint8 / float32 frames on the device")
            else:
                prm, flip_c = augment.draw_params(B, self.augment_generator)
                if not augment.is_identity(prm):
                    x = augment.apply(x, prm, crop)                 # fp32 0..255 frames of the crop window
                elif crop is not None:
                    x = x[..., crop[0]:crop[0] + cd, crop[1]:crop[1] + cd]
                if bool(flip_c.any()):
                    flip_frames = flip_c.to(self._device).to(torch.uint8).repeat_interleave(T).contiguous()
            eng = self.engine(act_dtype)
            head, _ = eng.forward_augmented(x.contiguous(), flip_frames)
            pw = eng.pw
            return self._pack_head(head, B, T, pw.n_cls, pw.displ_col, y)

        def forward(self, x, y=None, inference=False, augment_inference=False, act_dtype=torch.bfloat16, slot=0):
            """model.py:105-149.  x: (B,T,3,H,W) uint8, or float holding 0..255 values.  Like the reference's nn.Module,
            .train()/.eval() select BatchNorm statistics + dropout and `inference` selects the crop / augmentation branch.
            slot (eval only): which of the engine's independent buffer sets to use (two batches in flight on two streams)."""
            if self.training:
                return self._forward_train(x, y, inference, augment_inference)
            if not inference:
                return self._forward_eval_augmented(x, y, act_dtype)
            if x.dtype != torch.uint8:
                x = x.round().clamp_(0, 255).to(torch.uint8)
            x = x.to(self._device)
            B, T = x.shape[:2]
            eng = self.engine(act_dtype)
            head, _ = eng.forward(x.contiguous(), augment_inference, slot=slot)
            pw = eng.pw
            return self._pack_head(head, B, T, pw.n_cls, pw.displ_col, y)

        __call__ = forward

    # ----------------------------------------------------------------------------------------
    def __init__(self, device="cuda", args=None):
        self.device = device
        self._model = TDEEDModel.Impl(args=args)
        self._model.print_stats()
        self._args = args
        self._model.to(device)
        self._num_classes = args.num_classes + 1
        self._stream = None

    # BaseRGBModel (modules.py:35-55)
    def get_optimizer(self, opt_args):
        """modules.py:37-39: AdamW over all parameters (+ a GradScaler in the reference; bf16 needs none -> None).
        The returned optimizer is a torch.optim.Optimizer over the model's single flat parameter buffer whose step() is
        the fused AdamW kernel, so torch LR schedulers work on it unchanged."""
        from .trainer import HipAdamW
        eng = self._model.train_engine()
        if eng.reducer is None:
            eng.set_reducer("auto")          # data-parallel job (torch.distributed initialised, world > 1): bucketed all-reduce
        return HipAdamW(eng, **opt_args), None

    @property
    def _train_dtype(self):
        return self._model._train_dtype

    @_train_dtype.setter
    def _train_dtype(self, dt):
        if self._model._train_engine is not None:
            raise RuntimeError("set _train_dtype before get_optimizer() / the first train-mode forward")
        self._model._train_dtype = dt

    def _get_params(self):
        return list(self._model.parameters())

    def state_dict(self):
        return self._model.state_dict()

    def load(self, state_dict):
        self._model.load_state_dict(state_dict)

    def _ctx(self):
        if self._stream is None:
            from .streams import new_stream
            self._stream = new_stream()
        return torch.cuda.stream(self._stream)

    def predict(self, seq, use_amp=True, augment_inference=False):
        """model.py:334-369 -> (pred_cls (B,T) int64 numpy, scores (B,T,K+1) float32 numpy)."""
        if not isinstance(seq, torch.Tensor):
            seq = torch.as_tensor(np.asarray(seq))
        if seq.dim() == 4:
            seq = seq.unsqueeze(0)
        self._model.eval()
        dt = torch.bfloat16 if use_amp else torch.float32
        cur = torch.cuda.current_stream()
        with self._ctx():
            self._stream.wait_stream(cur)
            pred, _ = self._model(seq.to(self.device), inference=True, augment_inference=augment_inference,
                                  act_dtype=dt)
            B, T = seq.shape[:2]
            if isinstance(pred, dict):
                head = pred["_head_out"].reshape(B * T, -1)
                pw = self._model.engine(dt).pw
                k1 = (self._args.num_classes + 1) if self._model._double_head else pw.n_cls
                cls, scores = ops.process_prediction(head, B, T, k1, pw.displ_col)
            else:
                head = pred.reshape(B * T, -1).contiguous()
                cls, scores = ops.process_prediction(head, B, T, head.shape[-1], -1)
            self._stream.synchronize()
        return cls.cpu().numpy(), scores.cpu().numpy()

    def epoch(self, loader, optimizer=None, scaler=None, lr_scheduler=None, acc_grad_iter=1, fg_weight=5,
              valMAP=False):
        """model.py:193-332: validation pass (optimizer None) or one training epoch (optimizer from get_optimizer)."""
        if optimizer is not None:
            return self._train_epoch(loader, optimizer, lr_scheduler, acc_grad_iter, fg_weight)
        self._model.eval()
        K1 = self._num_classes
        w = torch.tensor([1.0] + [float(fg_weight)] * (K1 - 1), dtype=torch.float32, device=self.device)
        map_labels, map_preds = [], []
        n = 0
        # two batches in flight: consecutive batches alternate between two buffer sets / HIP graphs on two streams
        from .streams import new_stream
        if self._stream is None:
            self._stream = new_stream()
        if getattr(self, "_stream2", None) is None:
            self._stream2 = new_stream(avoid=[self._stream])
        streams = [self._stream, self._stream2]
        totals = [torch.zeros((), dtype=torch.float32, device=self.device) for _ in streams]
        torch.cuda.current_stream().synchronize()     # w / totals were filled on the current stream; the two are non-blocking
        from . import feeder
        for i, batch in enumerate(feeder.prefetch(loader, self.device, auto=False)):
            slot = i % 2
            with torch.cuda.stream(streams[slot]):
                feeder.wait(batch)                     # uint8 frames: pinned staging ring + copy stream, one batch ahead
                frame = batch["frame"].to(self.device)
                label = batch["label"].to(self.device)
                B, T = frame.shape[:2]
                pred, _ = self._model(frame, y=label, inference=True, slot=slot)
                labelD = batch["labelD"].to(self.device).float().reshape(-1).contiguous() if "labelD" in batch else None
                if isinstance(pred, dict):
                    head = pred["_head_out"].reshape(B * T, -1)
                    dcol = self._model.engine(torch.bfloat16).pw.displ_col if labelD is not None else -1
                else:
                    head, dcol = pred.reshape(B * T, -1).contiguous(), -1
                if self._model._double_head:
                    # joint-dataset validation (model.py:278-306): per-clip CE on the clip's own head
                    k1a, k1b = self._model._head_classes
                    ds = torch.as_tensor(batch["dataset"]).to(self.device).long()
                    lab2 = update_labels_2heads(label.clone(), ds, self._args.num_classes).reshape(-1).contiguous()
                    w2 = torch.tensor([1.0] + [float(fg_weight)] * (max(k1a, k1b) - 1), dtype=torch.float32, device=self.device)
                    out, _ = ops.loss2(head, B, T, k1a, k1b, ds, lab2, w2, displ_col=dcol, labelD=labelD)
                else:
                    out = ops.loss(head, K1, w, hard=label.reshape(-1).contiguous(), displ_col=dcol, labelD=labelD)
                totals[slot] += out[0]
                n += 1
                if valMAP:
                    cls, scores = ops.process_prediction(head, B, T, K1, dcol)
                    map_preds.append(scores.cpu())
                    from .modules import process_labels
                    map_labels.append(process_labels(label.cpu(), batch.get("labelD"), num_classes=K1))
                feeder.done(batch)
        for st in streams:
            st.synchronize()
        total = totals[0] + totals[1]
        avg = float(total.item()) / max(n, 1)       # one device sync per epoch, not per batch
        if valMAP:
            return avg, torch.cat(map_labels, 0), torch.cat(map_preds, 0)
        return avg


def _train_epoch_impl(self, loader, optimizer, lr_scheduler, acc_grad_iter, fg_weight):
    """Training branch of model.py:193-332 + BaseRGBModel.step (modules.py:390-404): per batch, mixup when the loader
    delivers 'frame2'/'label2'/'labelD2' (model.py:233-260: Beta(0.2,0.2) weights from `random`, soft labels), the
    train-mode `Impl.forward(frame, inference=False)` (one random crop per batch, per-clip augmentation, dropout), the
    loss (single or joint-dataset double head, hard or soft labels), the backward and, every `acc_grad_iter` batches,
    the optimizer + scheduler step."""
    import random
    eng = optimizer.engine
    if eng is not self._model.train_engine():
        raise RuntimeError("the optimizer was built for another model")
    self._model.train()
    optimizer.zero_grad()
    total = torch.zeros((), dtype=torch.float32, device=self.device)
    n = 0
    cur = torch.cuda.current_stream()
    with self._ctx():
        self._stream.wait_stream(cur)        # engine construction / load() / zero_grad were queued on the caller's stream
        from . import feeder
        for batch_idx, batch in enumerate(feeder.prefetch(loader, self.device, auto=False)):
            feeder.wait(batch)

            def u8(x):
                x = x.to(self.device)
                return (x if x.dtype == torch.uint8 else x.round().clamp_(0, 255).to(torch.uint8)).contiguous()
            frame = u8(batch["frame"])
            label = batch["label"].to(self.device)
            labelD = batch["labelD"].to(self.device).float() if "labelD" in batch else None
            B, T = frame.shape[:2]
            soft = None
            dataset = None
            if self._model._double_head:
                dataset = torch.as_tensor(batch["dataset"]).to(self.device).long()
                label = update_labels_2heads(label.clone(), dataset, self._args.num_classes)
            if "frame2" in batch:
                from . import ops_bwd
                K1 = self._num_classes                               # train_tdeed.py:148 widens it for the double head
                lam = torch.tensor([random.betavariate(0.2, 0.2) for _ in range(B)], dtype=torch.float32, device=self.device)
                frame = ops_bwd.mix_frames(frame, u8(batch["frame2"]), lam)          # fp32 0..255 frames
                label2 = batch["label2"].to(self.device)
                oh = torch.nn.functional.one_hot
                soft = (lam.view(B, 1, 1) * oh(label, K1).float() + (1 - lam).view(B, 1, 1) * oh(label2, K1).float())
                label = None
                if "labelD2" in batch:
                    labelD = lam.view(B, 1) * labelD + (1 - lam).view(B, 1) * batch["labelD2"].to(self.device).float()
            pred, _ = self._model(frame, y=label, inference=False)
            head = (pred["_head_out"] if isinstance(pred, dict) else pred).reshape(B * T, -1)
            scale = 1.0 / acc_grad_iter
            loss, dhead = eng.temporal.loss_fwd_bwd(
                head, B, T, None if label is None else label.reshape(-1).contiguous(),
                labelD=None if labelD is None else labelD.reshape(-1).contiguous(),
                soft=None if soft is None else soft.reshape(B * T, -1).contiguous(), fg_weight=fg_weight, dataset=dataset)
            last = (batch_idx + 1) % acc_grad_iter == 0
            # data parallel (one process per GPU under torchrun): the bucket all-reduces of the step's last micro-batch are
            # enqueued from inside the backward; optimizer.step() waits for them on the device
            eng.backward_and_write(self._model._train_ctx, dhead, scale=scale, first=batch_idx % acc_grad_iter == 0,
                                   reduce=last and eng.reducer is not None)
            self._model._train_ctx = None
            if last:
                optimizer.step()
                if lr_scheduler is not None:
                    lr_scheduler.step()
                optimizer.zero_grad()
            total += loss[0]
            n += 1
            feeder.done(batch)
        self._stream.synchronize()
    self._model._engines = {}                      # the inference engines hold packed copies of the old weights
    return float(total.item()) / max(n, 1)


TDEEDModel._train_epoch = _train_epoch_impl


def update_labels_2heads(labels, datasets, num_classes1=1):
    """model.py:371-376."""
    for i in range(len(datasets)):
        if datasets[i] == 2:
            labels[i] = labels[i] + num_classes1 + 1
    return labels
