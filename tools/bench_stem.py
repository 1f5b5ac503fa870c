"""GPU micro-benchmark: the training stem forward (MFMA) and its weight gradient at B=8 x T=100 x 224^2."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tdeed_amd import ops, ops_bwd
from tdeed_amd.engine import stem_frags_on_device
N = int(sys.argv[1]) if len(sys.argv) > 1 else 800
fr = ops.fill_u8_hash((N, 3, 224, 224), 7, "cuda")
w = torch.randn(32, 3, 3, 3, device="cuda") * 0.2
wf = stem_frags_on_device(w)
dz = torch.randn(N, 112, 112, 32, device="cuda").bfloat16()
def timeit(fn, reps=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3
print("stem_mfma %.1f us   stem_wgrad %.1f us   (write / read of the 112^2 x 32 map at 5.5 TB/s: %.0f us)" % (
    timeit(lambda: ops.stem_mfma(fr, wf)), timeit(lambda: ops_bwd.stem_wgrad(fr, dz)), N * 112 * 112 * 64 / 5.5e6))
