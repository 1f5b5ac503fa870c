"""GPU micro-benchmark: weight-stationary vs tiled contraction on the narrow / mid-width 1x1 convs of the inference forward
(sub-batch of 4 clips at 200MF, 8 clips at 800MF), plain and as conv3 (SE gate on the operand + residual + ReLU)."""
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
for (M, K, N, hw) in [(2508800, 64, 64, 3136), (2508800, 64, 128, 3136), (627200, 128, 128, 784), (627200, 128, 320, 784), (156800, 320, 320, 196),
                      (1254400, 24, 24, 3136), (313600, 56, 56, 784), (313600, 56, 152, 784), (78400, 152, 152, 196), (78400, 152, 368, 196)]:
    if not ops.gemm_ws_fits(K, N, torch.bfloat16):
        print(f"M={M} K={K} N={N}: does not fit the weight-stationary kernel"); continue
    A = torch.randn(M, K, device="cuda").bfloat16()
    W = (torch.randn(N, K, device="cuda") / K ** 0.5).bfloat16()
    Wf = pack_ws_weights(W.float().cpu().numpy(), torch.bfloat16, "cuda")
    sc, sh = torch.rand(N, device="cuda") + 0.5, torch.randn(N, device="cuda")
    out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    t0 = timeit(lambda: ops.gemm(A, W, sc, sh, 1, out=out))
    t1 = timeit(lambda: ops.gemm_ws(A, Wf, K, N, sc, sh, 1, out=out))
    line = f"M={M:8d} K={K:4d} N={N:4d}: conv1  tiled {t0:7.1f} us  ws {t1:7.1f} us"
    if K == N:
        R = torch.randn(M, N, device="cuda").bfloat16()
        gate = torch.rand(M // hw, K, device="cuda")
        t2 = timeit(lambda: ops.gemm(A, W, sc, sh, 1, residual=R, a_scale=gate, a_scale_rows=hw, out=out))
        t3 = timeit(lambda: ops.gemm_ws(A, Wf, K, N, sc, sh, 1, residual=R, a_scale=gate, a_scale_rows=hw, out=out))
        line += f" | conv3  tiled {t2:7.1f} us  ws {t3:7.1f} us"
    print(line, flush=True)
