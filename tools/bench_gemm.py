"""GPU micro-benchmark of the contraction kernels on the trunk's shapes."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tdeed_amd import ops
from tdeed_amd.engine import pack_ws_weights

def timeit(fn, reps=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3

dev = "cuda"
for (M, K, N) in [(39200, 368, 368), (39200, 768, 768), (156800, 152, 368), (156800, 152, 152), (800, 368, 1472), (800, 1472, 368), (800, 2208, 368)]:
    A = torch.randn(M, K, device=dev).bfloat16()
    W = (torch.randn(N, K, device=dev) / K ** 0.5).bfloat16()
    R = torch.randn(M, N, device=dev).bfloat16()
    sh = torch.randn(N, device=dev)
    fr = max(1, M // 49)
    gate = torch.rand(M // 49 if M % 49 == 0 else 1, K, device=dev)
    out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    res = {}
    res["plain"] = timeit(lambda: ops.gemm(A, W, None, sh, 1, out=out))
    res["+res"] = timeit(lambda: ops.gemm(A, W, None, sh, 1, residual=R, out=out))
    if M % 49 == 0:
        res["+res+se"] = timeit(lambda: ops.gemm(A, W, None, sh, 1, residual=R, a_scale=gate, a_scale_rows=49, out=out))
    if ops.gemm_ws_fits(K, N, torch.bfloat16):
        Wf = pack_ws_weights(W.float().cpu().numpy(), torch.bfloat16, dev)
        res["ws"] = timeit(lambda: ops.gemm_ws(A, Wf, K, N, None, sh, 1, out=out))
        res["ws+res"] = timeit(lambda: ops.gemm_ws(A, Wf, K, N, None, sh, 1, residual=R, out=out))
        if M % 49 == 0:
            res["ws+res+se"] = timeit(lambda: ops.gemm_ws(A, Wf, K, N, None, sh, 1, residual=R, a_scale=gate, a_scale_rows=49, out=out))
    flops = 2 * M * K * N
    byts = (M * K + N * K + M * N) * 2
    print(f"M={M} K={K} N={N}: " + "  ".join(f"{k}: {v:.1f}us ({flops / v / 1e6:.0f} TF/s, {byts / v / 1e3:.0f} GB/s)" for k, v in res.items()))
