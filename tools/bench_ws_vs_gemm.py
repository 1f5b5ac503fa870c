"""GPU tool: the narrow 1x1 convs of the training step on the tiled gemm kernel vs the weight-stationary one.
    python tools/bench_ws_vs_gemm.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import numpy as np
import tdeed_amd  # noqa
from tdeed_amd import ops
from tdeed_amd.engine import pack_ws_weights

def tm(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3

shapes = [(800 * 112 * 112, 32, 24), (800 * 112 * 112, 24, 32), (800 * 56 * 56, 24, 24), (800 * 56 * 56, 24, 56), (800 * 28 * 28, 56, 56),
          (800 * 28 * 28, 56, 152), (800 * 14 * 14, 152, 152), (1600 * 112 * 112, 32, 64), (1600 * 56 * 56, 64, 64), (1600 * 56 * 56, 64, 144),
          (1600 * 28 * 28, 144, 144)]
for M, K, N in shapes:
    A = torch.randn((M, K), device="cuda", dtype=torch.bfloat16)
    W = (np.random.randn(N, K) * 0.1).astype(np.float32)
    Wd = torch.from_numpy(W).cuda().to(torch.bfloat16)
    out = torch.empty((M, N), device="cuda", dtype=torch.bfloat16)
    t_g = tm(lambda: ops.gemm(A, Wd, None, None, ops.ACT_NONE, out=out, M=M))
    byts = (M * K + M * N) * 2
    line = f"M={M:9d} K={K:3d} N={N:3d}  gemm {t_g:8.1f} us {byts/t_g/1e6:6.2f} TB/s"
    if ops.gemm_ws_fits_mode(K, N, torch.bfloat16) == 1:
        Wf = pack_ws_weights(W, torch.bfloat16, "cuda")
        t_w = tm(lambda: ops.gemm_ws(A, Wf, K, N, None, None, ops.ACT_NONE, out=out, M=M))
        line += f" | gemm_ws {t_w:8.1f} us {byts/t_w/1e6:6.2f} TB/s"
    print(line, flush=True)
    del A, out
