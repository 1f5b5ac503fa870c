"""Time the fused stage-1 front launch alone (uint8 frames -> conv2 output, shortcut, squeeze sums).
    TDEED_FRONT_ROLL=<rows per strip | 0> python tools/bench_front.py [frames]      (regnety_002 stage-1 widths, 224 x 224)"""
import importlib
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pkg = importlib.import_module("t-deed_amd")
from tdeed_amd import ops                                   # noqa: E402
from tdeed_amd.engine import pack_front_weights            # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 800
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(0)
    C1, gw = 24, 8
    r = lambda *s: torch.randn(*s, generator=g) * 0.2                                   # noqa: E731
    aff = lambda c: (torch.rand(c, generator=g) + 0.5, r(c))                             # noqa: E731
    (ss, sh), (s1, h1), (sd, hd), (s2, h2) = aff(32), aff(C1), aff(C1), aff(C1)
    fw = pack_front_weights(r(32, 3, 3, 3), ss, sh, r(C1, 32), s1, h1, r(C1, 32), sd, hd, r(C1, gw, 3, 3), gw, s2, h2, dev)
    fr = torch.randint(0, 256, (n, 3, 224, 224), dtype=torch.uint8, generator=g).to(dev)
    outs = ops.s1_front(fr, fw, None, False)
    y2, sc, pooled = outs
    for _ in range(3):
        ops.s1_front(fr, fw, None, False, y2, sc, pooled)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = 20
    e0.record()
    for _ in range(reps):
        ops.s1_front(fr, fw, None, False, y2, sc, pooled)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / reps
    byt = fr.numel() + (y2.numel() + sc.numel()) * 2
    print(f"s1_front frames {n} roll {os.environ.get('TDEED_FRONT_ROLL', 'default')}: {us:.1f} us  "
          f"{byt / us / 1e3:.0f} GB/s algorithmic  checksum {float(y2.float().sum()):.3f}")


if __name__ == "__main__":
    main()
