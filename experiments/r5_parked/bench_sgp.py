"""Per-launch device time of the SGP stage pieces (narrow-C MLP form / fused / launch-per-op), each step replayed back to
back on one stream, and the whole chain as a captured HIP graph (device time incl. kernel boundaries, no host gaps).
    python tools/bench_sgp.py [B] [T] [C] [n_layers]"""
import ctypes
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
import tdeed_amd  # noqa: F401
from tdeed_amd import ops, _lib
from tdeed_amd.engine import SgpBuilder, pack_sgp_block, pack_sgp_mixer, _Pool
from helpers import module_state

PROFILE = "--profile" in sys.argv      # only the default chain as a graph, replayed (for rocprofv3 --kernel-trace --stats)
sys.argv = [a for a in sys.argv if a != "--profile"]
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
T = int(sys.argv[2]) if len(sys.argv) > 2 else 100
C = int(sys.argv[3]) if len(sys.argv) > 3 else 368
n = int(sys.argv[4]) if len(sys.argv) > 4 else 2
DEV = "cuda"


def timeit(fn, reps=50):
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


def graph_time(steps, reps=50):
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


sd = module_state("pyramid", "_temp_fine", 5, C=C, ks=7, r=4, n=n)
dt = torch.bfloat16
x = torch.randn((B, T, C), device=DEV).to(dt)
with torch.cuda.stream(torch.cuda.Stream()):
    for fused, mlp2 in ((("1", "1"),) if PROFILE else (("1", "1"), ("1", "0"), ("0", "0"))):
        os.environ["TDEED_SGP_FUSED"] = fused
        os.environ["TDEED_SGP_MLP2"] = mlp2
        sgp = [pack_sgp_block(sd, f"_temp_fine._sgp.{i}", C, dt, DEV) for i in range(2 * n + 1)]
        mix = [pack_sgp_mixer(sd, f"_temp_fine._sgpMixer.{i}", C, dt, DEV) for i in range(n)]
        steps, keep = [], {}
        sb = SgpBuilder(_Pool(DEV), steps, keep, set(), B, dt)
        sb.pyramid(x, T, n, sgp, mix)
        print(f"fused={fused} mlp2={mlp2}: {len(steps)} steps, chain as one HIP graph {graph_time(steps, 200 if PROFILE else 50):.1f} us")
        if PROFILE:
            break
        for s in steps:
            print(f"   {s.name:40s} {s.kernel:14s} {timeit(s.fn):7.1f} us")
