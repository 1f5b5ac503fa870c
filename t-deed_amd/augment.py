"""Train-time augmentation of `TDEEDModel.Impl.forward` (/root/reference/model/model.py:76-83, 154-157).

The reference runs, per clip i, `T.Compose([RandomApply([ColorJitter(hue=0.2)], p=0.25), RandomApply([ColorJitter(
saturation=(0.7, 1.2))], 0.25), RandomApply([ColorJitter(brightness=(0.7, 1.2))], 0.25), RandomApply([ColorJitter(
contrast=(0.7, 1.2))], 0.25), RandomApply([GaussianBlur(5)], 0.25), RandomHorizontalFlip()])` on the cropped 0..1 clip
(T,3,h,w): one parameter draw per clip, shared by its frames.  Here the draw happens on the host (`draw_params`) and the
arithmetic in csrc/augment.hip (`apply`) + the stem's per-frame flip flag.

RNG contract.  `draw_params` consumes torch's global CPU generator (or the generator passed in) in exactly the order
torchvision 0.18.1 does for that Compose, clip after clip:
    RandomApply.forward:            u = torch.rand(1);  skipped when p < u
    ColorJitter.forward/get_params: torch.randperm(4);  then float(torch.empty(1).uniform_(lo, hi)) for the one active factor
    GaussianBlur.forward/get_params: torch.empty(1).uniform_(0.1, 2.0)
    RandomHorizontalFlip.forward:   torch.rand(1) < 0.5
so a run seeded with torch.manual_seed(s) draws the same augmentation parameters as the reference seeded the same way
(torchvision is not installed in the build container: the order is restated from its source, not pinned by a fixture).
"""
import torch

from . import _lib
from ._lib import call, ptr, stream_ptr

P_APPLY = 0.25
HUE = (-0.2, 0.2)
SAT = BRI = CON = (0.7, 1.2)
SIGMA = (0.1, 2.0)
IDENTITY = (0.0, 1.0, 1.0, 1.0, 0.0)


def _rand(gen):
    return float(torch.rand(1, generator=gen))


def _uniform(lo, hi, gen):
    return float(torch.empty(1).uniform_(lo, hi, generator=gen))


def draw_params(B, generator=None):
    """-> (prm (B, 8) float32 CPU tensor {hue, sat, bri, con, sigma, 0, 0, 0}, flip (B,) uint8 CPU tensor)."""
    prm = torch.zeros((B, 8), dtype=torch.float32)
    flip = torch.zeros((B,), dtype=torch.uint8)
    for i in range(B):
        row = list(IDENTITY)
        for slot, rng in ((0, HUE), (1, SAT), (2, BRI), (3, CON)):
            if not (P_APPLY < _rand(generator)):
                torch.randperm(4, generator=generator)
                row[slot] = _uniform(rng[0], rng[1], generator)
        if not (P_APPLY < _rand(generator)):
            row[4] = _uniform(SIGMA[0], SIGMA[1], generator)
        flip[i] = 1 if _rand(generator) < 0.5 else 0
        prm[i, :5] = torch.tensor(row)
    return prm, flip


def is_identity(prm):
    return bool((prm[:, :5] == torch.tensor(IDENTITY)).all())


def apply(frames, prm, crop=None):
    """frames (B,T,3,H,W) uint8 or fp32 0..255 on the GPU; prm (B,8) fp32 (host or device); crop (top,left,h,w) or None.
    Returns fp32 0..255 frames (B,T,3,h,w) of the crop window with the colour / blur stages applied."""
    if not frames.is_cuda or not frames.is_contiguous():
        raise RuntimeError("augment.apply: frames must be a contiguous GPU tensor")
    if frames.dtype not in (torch.uint8, torch.float32):
        raise TypeError(f"augment.apply: uint8 or float32 frames expected, got {frames.dtype}")
    B, T, _, H, W = frames.shape
    top, left, ch, cw = crop if crop is not None else (0, 0, H, W)
    dev = frames.device
    prm_d = prm.to(device=dev, dtype=torch.float32).contiguous()
    if tuple(prm_d.shape) != (B, 8):
        raise ValueError(f"augment.apply: prm must be ({B}, 8), got {tuple(prm_d.shape)}")
    out = torch.empty((B, T, 3, ch, cw), dtype=torch.float32, device=dev)
    tmp = torch.empty_like(out) if bool((prm[:, 4] > 0).any()) else out[:0].new_empty(1)
    part = torch.empty(_lib.load().tdeed_augment_scratch_floats(B * T), dtype=torch.float32, device=dev)
    call("tdeed_augment_clips", ptr(frames), int(frames.dtype == torch.float32), B, T, H, W, top, left, ch, cw, ptr(prm_d),
         ptr(part), ptr(out), ptr(tmp), stream_ptr())
    return out


def crop_only(frames, crop):
    """An `Impl.augment_fn` that switches the train-time augmentation off: just the batch's crop window."""
    if crop is None:
        return frames
    top, left, ch, cw = crop
    return frames[..., top:top + ch, left:left + cw].contiguous()
