"""GPU tool: the SGP block front launch (LayerNorm + depthwise branches) alone at the shipped geometries, with its phase
time stamps (tdeed_sgp_front_set_debug).
    python tools/bench_sgp_front.py"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from tdeed_amd import ops, _lib
from tdeed_amd.regnet_spec import sgp_up_size

DEV = "cuda"


def timeit(fn, reps=200):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3


for (B, T, C, ks, parts) in [(8, 100, 368, 7, 0), (8, 100, 368, 7, 6), (8, 50, 368, 7, 6), (16, 100, 768, 7, 12), (4, 250, 768, 9, 12)]:
    g = torch.Generator().manual_seed(0)
    up = sgp_up_size(ks, 4)
    x = torch.randn(B, T, C, generator=g).to(DEV)
    wlen = 2 * ks + up + 2
    ln_w, ln_b = torch.ones(C, device=DEV), torch.zeros(C, device=DEV)
    dw = (torch.randn(C, wlen, generator=g) * 0.1).to(DEV)
    db = (torch.randn(5, C, generator=g) * 0.1).to(DEV)
    y = torch.empty_like(x)
    chs = torch.empty(B, C, 2, device=DEV)
    if parts == 0:
        rowstat = torch.stack([x.mean(-1), 1.0 / torch.sqrt(x.var(-1, unbiased=False) + 1e-5)], -1).reshape(B * T, 2).contiguous()
    else:
        xf = x.reshape(B * T, C)                     # (sum, sum of squares) per 64-column tile, as sgp_gemm MODE 1 leaves them
        tiles = [xf[:, 64 * i:min(64 * (i + 1), C)] for i in range(parts)]
        rowstat = torch.stack([torch.stack([tl.sum(-1), (tl * tl).sum(-1)], -1) for tl in tiles], 0).contiguous()   # (parts, B*T, 2)

    def run():
        ops.sgp_front(x, ks, up, ln_w, ln_b, dw, db, out=y, chsum=chs, rowstat=rowstat)

    us = timeit(run)
    nwg = B * ((C + 15) // 16)
    dbg = torch.zeros((nwg, 16), dtype=torch.int64, device=DEV)
    _lib.call("tdeed_sgp_front_set_debug", dbg.data_ptr())
    run()
    torch.cuda.synchronize()
    _lib.call("tdeed_sgp_front_set_debug", None)
    d = dbg.cpu().numpy().astype(np.float64) * 10.0          # ns
    t0 = d[:, 0].min()
    names = ["issue + row stats", "commit + barrier", "LN apply + barrier", "branches + barrier", "store + sums"]
    ph = [np.median(d[:, i + 1] - d[:, i]) for i in range(5)]
    print(f"B={B} T={T} C={C} ks={ks} rowstat parts={parts}: {us:6.2f} us per launch ({nwg} workgroups); first start -> last end "
          f"{(d[:, 5].max() - t0) / 1e3:.2f} us, starts spread {(d[:, 0].max() - t0) / 1e3:.2f} us, workgroup median "
          f"{np.median(d[:, 5] - d[:, 0]) / 1e3:.2f} us: " + ", ".join(f"{n} {v / 1e3:.2f}" for n, v in zip(names, ph)), flush=True)
