"""GPU micro-benchmark: the 1x1 contractions of a 200MF / 800MF training step (B=8 / 16, T=100) per column-tile width
(TDEED_GEMM_BN, read once per process -> child processes) and in the weight-stationary kernel."""
import sys, os, subprocess
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
SHAPES = [(156800, 152, 152), (156800, 56, 152), (627200, 56, 56), (2508800, 24, 24), (2508800, 24, 56), (10035200, 32, 24),
          (10035200, 24, 32), (39200, 368, 368), (1254400, 128, 128), (313600, 320, 320), (5017600, 64, 64)]
if len(sys.argv) > 1 and sys.argv[1] == "child":
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
    for (M, K, N) in SHAPES:
        A = torch.randn(M, K, device="cuda").bfloat16()
        W = (torch.randn(N, K, device="cuda") / K ** 0.5).bfloat16()
        out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
        cp = torch.empty(((M + 127) // 128, 2, N), device="cuda")
        t0 = timeit(lambda: ops.gemm(A, W, None, None, 0, out=out))
        t1 = timeit(lambda: ops.gemm(A, W, None, None, 0, out=out, colpart=cp))
        byts = (M * K + N * K + M * N) * 2
        line = f"M={M:9d} K={K:4d} N={N:4d}: {t0:8.1f} us {byts/t0/1e3:6.0f} GB/s | +stats {t1:8.1f} us"
        if os.environ.get("WS") == "1" and ops.gemm_ws_fits(K, N, torch.bfloat16):
            Wf = pack_ws_weights(W.float().cpu().numpy(), torch.bfloat16, "cuda")
            t2 = timeit(lambda: ops.gemm_ws(A, Wf, K, N, None, None, 0, out=out))
            line += f" | ws {t2:8.1f} us {byts/t2/1e3:6.0f} GB/s"
        print(line, flush=True)
else:
    for env in ({"WS": "1"}, {"TDEED_GEMM_BN": "32"}, {"TDEED_GEMM_BN": "64"}, {"TDEED_GEMM_BN": "128"}):
        print(env, flush=True)
        subprocess.run([sys.executable, os.path.abspath(__file__), "child"], env={**os.environ, **env})
