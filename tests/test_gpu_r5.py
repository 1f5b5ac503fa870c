"""Round-5 parity tests on the MI355X (through the C ABI).  -m gpu only."""
import numpy as np
import pytest
import torch

from helpers import load_golden, model_state, t, max_abs
from tdeed_amd import synth, state_layout

pytestmark = pytest.mark.gpu
DEV = "cuda"


def test_backward_after_a_repack_is_refused_where_it_would_recompute_from_the_new_weights():
    """the narrow one-launch backward (trunk_bwd3.hip, recompute=True) re-derives z from the packed weights;
    a repack between forward and backward would silently change what it differentiates.  The engine now refuses."""
    from tdeed_amd.trainer import TrainEngine
    cfg = dict(feature_arch="rny008_gsf", clip_len=4, crop_dim=None, n_layers=2, sgp_ks=5, sgp_r=2, num_classes=3,
               radi_displacement=0)
    B, T, H, W = 2, cfg["clip_len"], 64, 64
    sd0 = {k: t(v) for k, v in synth.make_state(state_layout.model_state_shapes(cfg), 9).items()}
    frames = t(synth.uint8_clip(561, (B, T, 3, H, W))).to(DEV)
    lab = t(synth.labels(562, B, T, cfg["num_classes"], 1, fg_frac=0.4)[0]).long().to(DEV)
    eng = TrainEngine(cfg, {k: v.clone() for k, v in sd0.items()}, act_dtype=torch.bfloat16, lr=1e-3)
    head, ctx = eng.forward_train(frames)
    _, dhead = eng.temporal.loss_fwd_bwd(head, B, T, lab.reshape(-1).contiguous())
    eng.repack()
    with pytest.raises(RuntimeError, match="packed weights changed"):
        eng.backward_train(ctx, dhead)
    # and the ordinary order still works
    head, ctx = eng.forward_train(frames)
    _, dhead = eng.temporal.loss_fwd_bwd(head, B, T, lab.reshape(-1).contiguous())
    eng.backward_train(ctx, dhead)
    torch.cuda.synchronize()
