"""Per-launch device time of the SGP stage pieces (fused vs launch-per-op), back to back on one stream.
    python tools/bench_sgp.py [B] [T] [C]"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
import tdeed_amd  # noqa: F401
from tdeed_amd import ops
from tdeed_amd.engine import SgpBuilder, pack_sgp_block, pack_sgp_mixer, _Pool
from helpers import module_state

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
T = int(sys.argv[2]) if len(sys.argv) > 2 else 100
C = int(sys.argv[3]) if len(sys.argv) > 3 else 368
DEV = "cuda"
n = 2


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


sd = module_state("pyramid", "_temp_fine", 5, C=C, ks=7, r=4, n=n)
dt = torch.bfloat16
x = torch.randn((B, T, C), device=DEV).to(dt)
with torch.cuda.stream(torch.cuda.Stream()):
    for fused, merge in (("1", None), ("0", None)):
        os.environ["TDEED_SGP_FUSED"] = fused
        sgp = [pack_sgp_block(sd, f"_temp_fine._sgp.{i}", C, dt, DEV) for i in range(2 * n + 1)]
        mix = [pack_sgp_mixer(sd, f"_temp_fine._sgpMixer.{i}", C, dt, DEV) for i in range(n)]
        steps, keep = [], {}
        sb = SgpBuilder(_Pool(DEV), steps, keep, set(), B, dt)
        sb.pyramid(x, T, n, sgp, mix)
        tot = timeit(lambda: [s.fn() for s in steps], 20)
        print(f"fused={fused}: {len(steps)} launches, chain {tot:.1f} us")
        for s in steps:
            print(f"   {s.name:40s} {s.kernel:14s} {timeit(s.fn):7.1f} us")
    o = sgp[0]
    y = torch.randn((B, T, C), device=DEV).to(dt)
    for S in (1, 2, 4):
        os.environ["TDEED_SGP_MLP_SPLIT"] = str(S)
    # the split is read once per process: report the automatic one
    print("auto split:", ops.sgp_mlp_partial_shape(B * T, C)[0])
