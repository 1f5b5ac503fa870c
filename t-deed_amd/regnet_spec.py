"""Static description of the two RegNetY backbones T-DEED uses.

The reference obtains them from the un-vendored ``timm==1.0.3`` package
(``/root/reference/model/model.py:38-45``, ``requirements.txt:39``):
``regnety_002`` (200 MF) and ``regnety_008`` (800 MF).  The numbers below are the
published RegNetY design-space results (widths / depths / group width, SE ratio
0.25 of the *block input* width, bottleneck ratio 1, stem width 32) and reproduce
the parameter counts 3,162,996 / 6,263,168 (with the 1000-way fc) that timm
reports for those models.

Every block is::

    conv1 1x1 (in->w) BN ReLU      <- GatedShift wraps this one in s3/s4
    conv2 3x3 grouped (w/gw groups, stride s) BN ReLU
    SE    mean(H,W) -> 1x1 (w->rd)+b ReLU -> 1x1 (rd->w)+b -> sigmoid -> scale
    conv3 1x1 (w->w) BN
    + shortcut (identity | 1x1 stride-s conv + BN)   -> ReLU
"""
from dataclasses import dataclass
from typing import List
import math


@dataclass(frozen=True)
class BlockSpec:
    name: str          # "s3.b2"
    stage: int         # 1..4
    index: int         # 1..depth
    cin: int
    cout: int
    stride: int
    groups: int
    gw: int
    se_rd: int
    has_downsample: bool
    gsf_fold: int      # 0 when the block's conv1 is not wrapped by GatedShift
    hin: int           # input spatial size for a 224 crop (informative)


@dataclass(frozen=True)
class RegNetSpec:
    arch: str
    stem_w: int
    widths: tuple
    depths: tuple
    gw: int
    blocks: tuple

    @property
    def feat_dim(self) -> int:
        return self.widths[-1]


_ARCH = {
    # arch: (widths, depths, group width)
    "rny002": ((24, 56, 152, 368), (1, 1, 4, 7), 8),
    "rny008": ((64, 128, 320, 768), (1, 3, 8, 2), 16),
}


def gsf_fold_dim(channels: int, n_div: int = 4) -> int:
    """``GatedShift.fold_dim`` (/root/reference/model/shift.py:79)."""
    return math.ceil(channels // n_div / 4) * 4


def regnet_spec(feature_arch: str, crop: int = 224) -> RegNetSpec:
    """feature_arch is the reference's string, e.g. ``rny002_gsf`` / ``rny008_gsm`` / ``rny002``."""
    base = feature_arch.rsplit("_", 1)[0] if "_" in feature_arch else feature_arch
    if base not in _ARCH:
        raise NotImplementedError(feature_arch)
    shifted = feature_arch.endswith(("_gsf", "_gsm"))
    widths, depths, gw = _ARCH[base]
    blocks: List[BlockSpec] = []
    cin = 32
    h = (crop + 1) // 2  # after the stride-2 stem
    for si, (w, d) in enumerate(zip(widths, depths), start=1):
        for bi in range(1, d + 1):
            stride = 2 if bi == 1 else 1
            blocks.append(BlockSpec(
                name=f"s{si}.b{bi}", stage=si, index=bi, cin=cin, cout=w, stride=stride,
                groups=w // gw, gw=gw, se_rd=int(round(cin * 0.25)),
                has_downsample=(cin != w or stride != 1),
                gsf_fold=(gsf_fold_dim(cin) if (shifted and si >= 3) else 0),
                hin=h))
            if stride == 2:
                h = (h + 1) // 2
            cin = w
    return RegNetSpec(arch=base, stem_w=32, widths=widths, depths=depths, gw=gw, blocks=tuple(blocks))


def sgp_up_size(ks: int, r) -> int:
    """Wide depthwise kernel size of an SGP block (/root/reference/model/modules.py:119-120)."""
    up = round((ks + 1) * r)
    return up + 1 if up % 2 == 0 else up


def pyramid_lengths(clip_len: int, n_layers: int):
    """Temporal lengths of the encoder levels: [L, ceil(L/2), ceil(L/4), ...]
    (/root/reference/model/modules.py:64,66)."""
    return [math.ceil(clip_len / (2 ** i)) for i in range(n_layers + 1)]
