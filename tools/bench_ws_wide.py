"""GPU micro-benchmark: the sliced weight-stationary contraction (W slice in LDS, activations streamed, gemm_ws mode 2) against
the tiled one on the wide 1x1 convs of RegNetY-800MF at the training / inference row counts (B=16 / B=8 clips of 100 frames).
    TDEED_WS_CAP_KB=96|128|150 python tools/bench_ws_wide.py"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tdeed_amd import ops
from tdeed_amd.engine import pack_ws_weights


def timeit(fn, reps=20):
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


print("TDEED_WS_CAP_KB =", os.environ.get("TDEED_WS_CAP_KB", "96 (default)"))
for (M, K, N) in [(313600, 320, 320), (156800, 320, 320), (313600, 128, 320), (313600, 320, 128), (1254400, 128, 128),
                  (78400, 320, 768), (39200, 368, 368)]:
    mode = ops.gemm_ws_fits_mode(K, N, torch.bfloat16)
    A = torch.randn(M, K, device="cuda").bfloat16()
    W = (torch.randn(N, K, device="cuda") / K ** 0.5).bfloat16()
    sc, sh = torch.rand(N, device="cuda") + 0.5, torch.randn(N, device="cuda")
    out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    t0 = timeit(lambda: ops.gemm(A, W, sc, sh, 1, out=out))
    line = f"M={M:8d} K={K:4d} N={N:4d}: tiled {t0:7.1f} us ({2.0 * M * K * N / t0 / 1e6:6.1f} TF)"
    if mode:
        Wf = pack_ws_weights(W.float().cpu().numpy(), torch.bfloat16, "cuda")
        ref = out.clone()
        t1 = timeit(lambda: ops.gemm_ws(A, Wf, K, N, sc, sh, 1, out=out))
        err = float((out.float() - ref.float()).abs().max())
        line += f"   ws(mode {mode}) {t1:7.1f} us ({2.0 * M * K * N / t1 / 1e6:6.1f} TF)  max diff {err:.3g}"
    else:
        line += "   ws: does not fit"
    if ops.gemm_rs_fits(M, K, N):
        Wf = pack_ws_weights(W.float().cpu().numpy(), torch.bfloat16, "cuda")
        ops.gemm(A, W, sc, sh, 1, out=out)
        ref = out.clone()
        t2 = timeit(lambda: ops.gemm_rs(A, Wf, K, N, sc, sh, 1, out=out))
        line += f"   rs {t2:7.1f} us ({2.0 * M * K * N / t2 / 1e6:6.1f} TF)  max diff {float((out.float() - ref.float()).abs().max()):.3g}"
        R = torch.randn(M, N, device="cuda").bfloat16()
        gate = torch.rand(M // 196, K, device="cuda")
        t3 = timeit(lambda: ops.gemm(A, W, sc, sh, 1, residual=R, a_scale=gate, a_scale_rows=196, out=out))
        ref = out.clone()
        t4 = timeit(lambda: ops.gemm_rs(A, Wf, K, N, sc, sh, 1, residual=R, a_scale=gate, a_scale_rows=196, out=out))
        line += f" | conv3: tiled {t3:7.1f} us  rs {t4:7.1f} us  max diff {float((out.float() - ref.float()).abs().max()):.3g}"
        t5 = timeit(lambda: ops.gemm_rs(A, Wf, K, N, sc, sh, 1, residual=R, out=out))
        t6 = timeit(lambda: ops.gemm_rs(A, Wf, K, N, sc, sh, 1, a_scale=gate, a_scale_rows=196, out=out))
        line += f" (rs residual only {t5:.1f}, gate only {t6:.1f})"
    print(line, flush=True)
