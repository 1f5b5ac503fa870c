"""Build-container-only harness: lets ``/root/reference/model/model.py`` import and run.

The reference needs ``timm`` and ``torchvision`` (neither installed, no network).
This file registers minimal stand-ins in ``sys.modules`` *for the golden-vector
script only* (SURVEY.md section 8c recipe).  The stand-in ``timm`` RegNet is an
nn.Module with timm's attribute / state_dict names (stem, s1..s4, b1..bN,
conv1/conv2/conv3/se/downsample, head.fc) so that the reference's own
``make_temporal_shift`` / ``GatedShift`` / ``_GSF`` / SGP / heads / loss code runs
unmodified on top of it.  Nothing here is shipped to the GPU box as a dependency
of tests or product; it never travels as "reference code" either -- it is ours.
"""
import sys
import types
import torch
from torch import nn

sys.path.insert(0, __file__.rsplit("/", 2)[0])
import tdeed_amd  # noqa: E402
from tdeed_amd.regnet_spec import regnet_spec  # noqa: E402


class ConvBnAct(nn.Module):
    def __init__(self, cin, cout, k=1, stride=1, groups=1, act=True):
        super().__init__()
        self.conv = nn.Conv2d(cin, cout, k, stride=stride, padding=k // 2, groups=groups, bias=False)
        self.bn = nn.BatchNorm2d(cout)
        self.act = act

    def forward(self, x):
        x = self.bn(self.conv(x))
        return torch.relu(x) if self.act else x


class _SE(nn.Module):
    def __init__(self, c, rd):
        super().__init__()
        self.fc1 = nn.Conv2d(c, rd, 1)
        self.fc2 = nn.Conv2d(rd, c, 1)

    def forward(self, x):
        s = x.mean((2, 3), keepdim=True)
        return x * torch.sigmoid(self.fc2(torch.relu(self.fc1(s))))


class _Block(nn.Module):
    def __init__(self, b):
        super().__init__()
        self.conv1 = ConvBnAct(b.cin, b.cout, 1)
        self.conv2 = ConvBnAct(b.cout, b.cout, 3, stride=b.stride, groups=b.groups)
        self.se = _SE(b.cout, b.se_rd)
        self.conv3 = ConvBnAct(b.cout, b.cout, 1, act=False)
        self.downsample = ConvBnAct(b.cin, b.cout, 1, stride=b.stride, act=False) if b.has_downsample else None

    def forward(self, x):
        s = x if self.downsample is None else self.downsample(x)
        return torch.relu(self.conv3(self.se(self.conv2(self.conv1(x)))) + s)


class _Head(nn.Module):
    def __init__(self, c):
        super().__init__()
        self.fc = nn.Linear(c, 1000)

    def forward(self, x):
        return self.fc(x.mean((2, 3)))


class RegNet(nn.Module):
    def __init__(self, arch):
        super().__init__()
        spec = regnet_spec(arch)            # no _gsf suffix: plain trunk, reference adds the shifts
        self.stem = ConvBnAct(3, 32, 3, stride=2)
        for si in range(1, 5):
            blocks = [(f"b{b.index}", _Block(b)) for b in spec.blocks if b.stage == si]
            stage = nn.Sequential()
            for n, m in blocks:
                stage.add_module(n, m)
            setattr(self, f"s{si}", stage)
        self.head = _Head(spec.feat_dim)

    def forward(self, x):
        x = self.stem(x)
        for si in range(1, 5):
            x = getattr(self, f"s{si}")(x)
        return self.head(x)


def install():
    timm = types.ModuleType("timm")
    timm.create_model = lambda name, pretrained=False: RegNet({"regnety_002": "rny002", "regnety_008": "rny008"}[name])
    timm.models = types.ModuleType("timm.models")
    timm.models.regnet = types.ModuleType("timm.models.regnet")
    timm.models.regnet.RegNet = RegNet
    timm.layers = types.ModuleType("timm.layers")
    timm.layers.conv_bn_act = types.ModuleType("timm.layers.conv_bn_act")
    timm.layers.conv_bn_act.ConvBnAct = ConvBnAct
    for n, m in [("timm", timm), ("timm.models", timm.models), ("timm.models.regnet", timm.models.regnet),
                 ("timm.layers", timm.layers), ("timm.layers.conv_bn_act", timm.layers.conv_bn_act)]:
        sys.modules[n] = m

    tv = types.ModuleType("torchvision")
    tv.models = types.ModuleType("torchvision.models")
    tv.models.ResNet = type("ResNet", (), {})
    tv.models.resnet = types.ModuleType("torchvision.models.resnet")
    tv.models.resnet.BasicBlock = type("BasicBlock", (), {})
    tv.ops = types.ModuleType("torchvision.ops")
    tv.ops.misc = types.ModuleType("torchvision.ops.misc")
    tv.ops.misc.ConvNormActivation = type("ConvNormActivation", (), {})
    T = types.ModuleType("torchvision.transforms")

    class Compose:
        def __init__(self, ts):
            self.ts = ts

        def __call__(self, x):
            for t in self.ts:
                x = t(x)
            return x

    class Normalize:
        def __init__(self, mean, std):
            self.mean = torch.tensor(mean).view(-1, 1, 1)
            self.std = torch.tensor(std).view(-1, 1, 1)

        def __call__(self, x):
            return (x - self.mean) / self.std

    class CenterCrop:
        def __init__(self, size):
            self.size = size

        def __call__(self, x):
            h, w = x.shape[-2:]
            ch, cw = self.size
            top, left = int(round((h - ch) / 2.0)), int(round((w - cw) / 2.0))
            return x[..., top:top + ch, left:left + cw]

    class RandomHorizontalFlip:
        def __init__(self, p=0.5):
            self.p = p

        def __call__(self, x):
            assert self.p == 1.0, "only the deterministic test-time flip is exercised"
            return x.flip(-1)

    def _dummy(name):
        return type(name, (), {"__init__": lambda self, *a, **k: None})

    T.Compose, T.Normalize, T.CenterCrop, T.RandomHorizontalFlip = Compose, Normalize, CenterCrop, RandomHorizontalFlip
    for n in ("RandomApply", "ColorJitter", "GaussianBlur", "RandomCrop"):
        setattr(T, n, _dummy(n))
    tv.transforms = T
    # torchvision.io.read_image (dataset/frame.py:271, 555) over Pillow's libjpeg: uint8 (3,H,W) RGB; a missing or
    # unreadable file raises RuntimeError like torchvision's decoder does (frame.py:611-615 relies on that)
    tvio = types.ModuleType("torchvision.io")

    def read_image(path):
        import numpy as np
        from PIL import Image
        try:
            with Image.open(path) as im:
                a = np.asarray(im.convert("RGB"))
        except (OSError, FileNotFoundError) as e:
            raise RuntimeError(str(e))
        return torch.from_numpy(np.ascontiguousarray(np.moveaxis(a, 2, 0)))
    tvio.read_image = read_image
    tv.io = tvio
    sys.modules["torchvision.io"] = tvio
    for n, m in [("torchvision", tv), ("torchvision.models", tv.models), ("torchvision.models.resnet", tv.models.resnet),
                 ("torchvision.ops", tv.ops), ("torchvision.ops.misc", tv.ops.misc), ("torchvision.transforms", T)]:
        sys.modules[n] = m
    if "/root/reference" not in sys.path:
        sys.path.insert(0, "/root/reference")
