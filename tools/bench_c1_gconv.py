"""GPU tool: conv1 + grouped 3x3 in one launch (tdeed_c1_gconv_fwd) alone at the shapes of the shipped models, with its phase
time stamps (tdeed_c1_gconv_set_debug).
    python tools/bench_c1_gconv.py"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from tdeed_amd import ops, _lib
from tdeed_amd.engine import pack_mfma_frags, pack_gconv_frags

DEV = "cuda"


def timeit(fn, reps=30):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3


SHAPES = [("200MF s2.b1", 800, 56, 24, 56, 8, 2), ("200MF s3.b1", 800, 28, 56, 152, 8, 2), ("200MF s4.b1", 800, 14, 152, 368, 8, 2),
          ("800MF s2.b1", 1600, 56, 64, 128, 16, 2), ("800MF s2.b2", 1600, 28, 128, 128, 16, 1), ("800MF s3.b1", 1600, 28, 128, 320, 16, 2)]
for name, N, Hi, Cin, C, gw, stride in SHAPES:
    g = torch.Generator().manual_seed(0)
    if not ops.c1_gconv_fits(Hi, Hi, Cin, C, stride):
        print(f"{name}: not served")
        continue
    x = torch.relu(torch.randn(N, Hi, Hi, Cin, generator=g)).to(torch.bfloat16).to(DEV)
    W1 = torch.randn(C, Cin, generator=g) / Cin ** 0.5
    W2 = torch.randn(C, gw, 3, 3, generator=g) / (gw * 9) ** 0.5
    vec = lambda n, s=0.1, o=0.0: (torch.randn(n, generator=g) * s + o).to(DEV)      # noqa: E731
    s1, h1, s2, h2 = vec(C, .1, 1.), vec(C), vec(C, .1, 1.), vec(C)
    rows = 16 * ops.c1_gconv_slab_tiles(Hi, Hi, C, stride)
    w1f = pack_mfma_frags(W1.numpy(), DEV, rows=rows)
    w2f = pack_gconv_frags(W2.numpy(), gw, DEV)
    Ho = (Hi - 1) // stride + 1
    out = torch.empty((N, Ho, Ho, C), dtype=torch.bfloat16, device=DEV)
    parts = ops.gconv3x3_parts(Hi, Hi, C, stride, torch.bfloat16)
    pooled = torch.empty((N, parts, C), device=DEV)

    def run():
        ops.c1_gconv(x, w1f, s1, h1, w2f, s2, h2, gw, stride, C, out=out, pooled=pooled)

    us = timeit(run)
    nslabs = rows // 16 // max(1, (min(C, 64) + 15) // 16) if C >= 64 else 1
    nwg = N * parts * ((C + 63) // 64)
    dbg = torch.zeros((nwg + 64, 8), dtype=torch.int64, device=DEV)
    _lib.call("tdeed_c1_gconv_set_debug", dbg.data_ptr())
    run()
    torch.cuda.synchronize()
    _lib.call("tdeed_c1_gconv_set_debug", None)
    d = dbg.cpu().numpy().astype(np.float64)[:nwg] * 10.0 / 1e3          # us
    ok = d[:, 6] > 0
    d = d[ok]
    byt = (x.numel() + out.numel()) * 2
    names = ["weights + halo", "conv1 tiles (wave 0)", "barrier", "grouped conv setup", "grouped conv tiles + stores", "squeeze sums"]
    ph = [np.median(d[:, i + 1] - d[:, i]) for i in range(6)]
    print(f"{name} N={N} {Hi}x{Hi} {Cin}->{C} stride {stride}: {us:7.1f} us per launch, {byt / us / 1e3:6.0f} GB/s algorithmic, {nwg} workgroups "
          f"({int(ok.sum())} stamped); first start -> last end {(d[:, 6].max() - d[:, 0].min()):.1f} us; workgroup median "
          f"{np.median(d[:, 6] - d[:, 0]):.2f} us: " + ", ".join(f"{n_} {v:.2f}" for n_, v in zip(names, ph)), flush=True)
