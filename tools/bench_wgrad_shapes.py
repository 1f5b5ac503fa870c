"""GPU micro-benchmark: weight gradients of the 1x1 convs of an 800MF B=16 training step against the time their two operand
reads cost at 5.5 TB/s."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tdeed_amd import ops_bwd as B_

def timeit(fn, reps=10):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3
tot = 0.0
for (M, N, K, cnt) in [(20070400, 64, 32, 1), (5017600, 64, 64, 1), (5017600, 128, 64, 1), (1254400, 128, 128, 5), (1254400, 320, 128, 1),
                       (313600, 320, 320, 15), (313600, 768, 320, 1), (78400, 768, 768, 3)]:
    dY = torch.randn(M, N, device="cuda").bfloat16()
    X = torch.randn(M, K, device="cuda").bfloat16()
    t = timeit(lambda: B_.wgrad(dY, X, with_bias=False))
    byts = M * (N + K) * 2
    tot += t * cnt
    print(f"M={M:9d} N={N:4d} K={K:4d} x{cnt:2d}: {t:8.1f} us  {byts/t/1e3:6.0f} GB/s  (reads at 5.5 TB/s: {byts/5.5e6:7.1f} us)  {2*M*N*K/t/1e6:6.0f} TF/s", flush=True)
    del dY, X
print("sum over the step's layers: %.2f ms" % (tot / 1e3))
