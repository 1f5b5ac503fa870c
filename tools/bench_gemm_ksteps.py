"""GPU tool: fixed cost vs per-K-slab cost of the tiled contraction (M=19600, N=368: the s4 1x1 convs of a half batch)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tdeed_amd import ops
def timeit(fn, reps=50):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3
M, N = 19600, 368
for K in (64, 128, 192, 256, 368):
    A = torch.randn(M, K, device="cuda").bfloat16(); W = torch.randn(N, K, device="cuda").bfloat16()
    R = torch.randn(M, N, device="cuda").bfloat16(); sh = torch.randn(N, device="cuda")
    out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    print(K, "plain %.1f us" % timeit(lambda: ops.gemm(A, W, None, sh, 1, out=out)), " +res %.1f us" % timeit(lambda: ops.gemm(A, W, None, sh, 1, residual=R, out=out)))
