"""GPU tool: time gemm_splitk vs gemm on the SGP shapes (run under rocprofv3 --kernel-trace --stats for the split)."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from tdeed_amd import ops

for (M, K, N) in [(800, 1472, 368), (800, 368, 1472), (800, 2208, 368), (400, 1472, 368)]:
    A = torch.randn(M, K, device="cuda").to(torch.bfloat16)
    W = (torch.randn(N, K, device="cuda") * 0.05).to(torch.bfloat16)
    b = torch.randn(N, device="cuda")
    out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    ws = ops.gemm_splitk_workspace(M, K, N, "cuda")
    for name, fn in [("splitk", lambda: ops.gemm_splitk(A, W, None, b, 2, out=out, workspace=ws)),
                     ("tiled", lambda: ops.gemm(A, W, None, b, 2, out=out))]:
        for _ in range(5):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(200):
            fn()
        e1.record()
        torch.cuda.synchronize()
        print(f"M={M} K={K} N={N} {name:7s} {e0.elapsed_time(e1) / 200 * 1e3:8.1f} us")
