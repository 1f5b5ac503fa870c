"""Device time of every tile form of the sgp_gemm.hip contractions at the shapes of the shipped configs (each launch
replayed back to back on one stream: kernel + boundary), and of the whole SGP stage as a captured HIP graph.
    python tools/bench_sgp_gemm.py [--stage-only] [--profile]"""
import ctypes
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
import tdeed_amd  # noqa: F401
from tdeed_amd import ops, _lib
from tdeed_amd.engine import SgpBuilder, pack_sgp_block, pack_sgp_mixer, pack_mfma_frags, _Pool
from helpers import module_state

DEV = "cuda"
FORMS = [(4, 2), (4, 1), (2, 2), (2, 1), (1, 2), (1, 1)]


def timeit(fn, reps=100):
    st = torch.cuda.current_stream()
    for _ in range(5):
        fn()
    st.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(st)
    for _ in range(reps):
        fn()
    b.record(st)
    st.synchronize()
    return a.elapsed_time(b) / reps * 1e3


def graph_time(steps, reps=100):
    st = torch.cuda.current_stream()
    for s in steps:
        s.fn()
    st.synchronize()
    h = ctypes.c_void_p()
    _lib.call("tdeed_graph_begin", st.cuda_stream)
    try:
        for s in steps:
            s.fn()
    finally:
        _lib.call("tdeed_graph_end", st.cuda_stream, ctypes.byref(h))
    us = timeit(lambda: _lib.call("tdeed_graph_launch", h, st.cuda_stream), reps)
    _lib.call("tdeed_graph_destroy", h)
    return us


def stage(B, T, C, n, ks, stream_dt):
    sd = module_state("pyramid", "_temp_fine", 5, C=C, ks=ks, r=4, n=n)
    dt = torch.bfloat16
    x = torch.randn((B, T, C), device=DEV).to(stream_dt)
    sgp = [pack_sgp_block(sd, f"_temp_fine._sgp.{i}", C, dt, DEV) for i in range(2 * n + 1)]
    mix = [pack_sgp_mixer(sd, f"_temp_fine._sgpMixer.{i}", C, dt, DEV) for i in range(n)]
    steps, keep = [], {}
    sb = SgpBuilder(_Pool(DEV), steps, keep, set(), B, dt)
    sb.pyramid(x, T, n, sgp, mix)
    return steps


with torch.cuda.stream(torch.cuda.Stream()):
    prof = "--profile" in sys.argv
    for (B, T, C, n, ks) in ((8, 100, 368, 2, 7), (16, 100, 768, 3, 7), (4, 250, 768, 2, 9)):
        for sdt in (torch.float32,) if prof else (torch.float32, torch.bfloat16):
            steps = stage(B, T, C, n, ks, sdt)
            print(f"stage B={B} T={T} C={C} n={n} stream={str(sdt)[6:]}: {len(steps)} steps, one HIP graph "
                  f"{graph_time(steps, 300 if prof else 100):.1f} us", flush=True)
            if not prof and sdt == torch.float32:
                for s in steps:
                    print(f"   {s.name:36s} {s.kernel:12s} {timeit(s.fn):7.1f} us")
        if prof:
            break
    if "--stage-only" in sys.argv or prof:
        sys.exit(0)
    for (B, C) in ((8, 368), (16, 768)):
        for T in (100, 50, 25):
            y = torch.randn((B, T, C), device=DEV)
            yb = y.to(torch.bfloat16)
            chs = torch.stack([y.sum(1), (y * y).sum(1)], -1).contiguous()
            gw, gb = torch.ones(C, device=DEV), torch.zeros(C, device=DEV)
            W1 = pack_mfma_frags(torch.randn(4 * C, C).numpy() * 0.05, DEV, ks_mult=12)
            W2 = pack_mfma_frags(torch.randn(C, 4 * C).numpy() * 0.05, DEV, ks_mult=12)
            Wc = pack_mfma_frags(torch.randn(C, 6 * C).numpy() * 0.05, DEV, ks_mult=12)
            b1, b2 = torch.zeros(4 * C, device=DEV), torch.zeros(C, device=DEV)
            H = torch.randn((B, T, 4 * C), device=DEV).to(torch.bfloat16)
            cat = torch.randn((B, T, 6 * C), device=DEV).to(torch.bfloat16)
            out = torch.empty_like(y)
            rows = []
            for f in FORMS:
                NJ, nct = ops.sgp_gemm_tiles(T, C, f)
                rsp = torch.empty((nct, B * T, 2), device=DEV)
                chso = torch.empty((NJ, B, C, 2), device=DEV)
                t0 = timeit(lambda: ops.sgp_gemm_gn_gelu(y, chs, gw, gb, W1, b1, 4 * C, out=H, form=f)) if f != (4, 2) else float("nan")
                t0b = timeit(lambda: ops.sgp_gemm_gn_gelu(yb, chs, gw, gb, W1, b1, 4 * C, out=H, form=f))
                t1 = timeit(lambda: ops.sgp_gemm_residual(H, W2, b2, y, out=out, rowstat_part=rsp, form=f))
                t2 = timeit(lambda: ops.sgp_gemm_gelu_chsum(cat, Wc, b2, C, out, chso, form=f))
                rows.append((f, t0, t0b, t1, t2))
            pick = [ops.sgp_gemm_form(m, B, T, N, K) for m, N, K in ((3, 4 * C, C), (0, 4 * C, C), (1, C, 4 * C), (2, C, 6 * C))]
            print(f"B={B} T={T} C={C}   form: fc1(f32 rows) fc1(bf16 rows) fc2 concat [us]; launcher picks {pick}")
            for f, t0, t0b, t1, t2 in rows:
                print(f"   {f}: {t0:7.1f} {t0b:7.1f} {t1:7.1f} {t2:7.1f}", flush=True)
