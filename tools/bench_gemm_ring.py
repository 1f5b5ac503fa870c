"""GPU micro-benchmark: register-staged tile (TDEED_GEMM_RING=0) against the LDS-DMA ring forms, run as child processes per
setting (the switch is read once per process).  The ring kernels are parked in experiments/gemm_ring.hip: without them plugged
into gemm.hip (see the header of that file) every setting measures the shipped kernel.  python tools/bench_gemm_ring.py [child]"""
import sys, os, subprocess
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
SHAPES = [(19600, 368, 368), (19600, 152, 368), (78400, 152, 152), (156800, 320, 320), (39200, 784, 784), (78400, 784, 784),
          (39200, 320, 784), (313600, 320, 320), (627200, 144, 144), (2508800, 64, 64), (9800, 368, 368)]
if len(sys.argv) > 1 and sys.argv[1] == "child":
    import torch
    from tdeed_amd import ops

    def timeit(fn, reps=30):
        for _ in range(5): fn()
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(reps): fn()
        b.record(); torch.cuda.synchronize()
        return a.elapsed_time(b) / reps * 1e3
    for (M, K, N) in SHAPES:
        A = torch.randn(M, K, device="cuda").bfloat16()
        W = (torch.randn(N, K, device="cuda") / K ** 0.5).bfloat16()
        R = torch.randn(M, N, device="cuda").bfloat16()
        sh = torch.randn(N, device="cuda")
        out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
        t0 = timeit(lambda: ops.gemm(A, W, None, sh, 1, out=out))
        t1 = timeit(lambda: ops.gemm(A, W, None, sh, 1, residual=R, out=out))
        ref = torch.relu(A[-4096:].float() @ W.float().T + sh + R[-4096:].float())
        err = float((out[-4096:].float() - ref).abs().max() / ref.abs().max())
        flops, byts = 2 * M * K * N, (M * K + N * K + M * N) * 2
        print(f"M={M:8d} K={K:4d} N={N:4d}: plain {t0:8.1f} us {flops/t0/1e6:6.0f} TF/s {byts/t0/1e3:6.0f} GB/s | +res {t1:8.1f} us  err {err:.1e}", flush=True)
else:
    for env in ({"TDEED_GEMM_RING": "0"}, {"TDEED_GEMM_RING": "2", "TDEED_GEMM_RING_NS": "3"}, {"TDEED_GEMM_RING": "2", "TDEED_GEMM_RING_NS": "4"}):
        print(env, flush=True)
        subprocess.run([sys.executable, os.path.abspath(__file__), "child"], env={**os.environ, **env})
