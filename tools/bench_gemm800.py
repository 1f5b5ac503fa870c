"""GPU micro-benchmark: the tiled contraction on the RegNetY-800MF shapes (sub-batch of 8 clips)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tdeed_amd import ops

def timeit(fn, reps=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3

for (M, K, N) in [(156800, 320, 320), (156800, 144, 320), (39200, 784, 784), (39200, 320, 784), (627200, 144, 144), (800, 784, 3136),
                  (800, 3136, 784), (1254400, 144, 144), (313600, 320, 320), (78400, 784, 784)]:
    A = torch.randn(M, K, device="cuda").bfloat16()
    W = (torch.randn(N, K, device="cuda") / K ** 0.5).bfloat16()
    R = torch.randn(M, N, device="cuda").bfloat16()
    sh = torch.randn(N, device="cuda")
    out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    t0 = timeit(lambda: ops.gemm(A, W, None, sh, 1, out=out))
    t1 = timeit(lambda: ops.gemm(A, W, None, sh, 1, residual=R, out=out))
    flops, byts = 2 * M * K * N, (M * K + N * K + M * N) * 2
    print(f"M={M:8d} K={K:4d} N={N:4d}: plain {t0:8.1f} us {flops/t0/1e6:6.0f} TF/s {byts/t0/1e3:6.0f} GB/s | +res {t1:8.1f} us {flops/t1/1e6:6.0f} TF/s", flush=True)
