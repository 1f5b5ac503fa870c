"""Device time of the fused GroupNorm+MLP launch alone (rocprofv3 --kernel-trace --stats -- python3 tools/bench_mlp.py)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
import tdeed_amd  # noqa: F401
from tdeed_amd import ops
from tdeed_amd.engine import pack_sgp_block
from helpers import module_state
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
T = int(sys.argv[2]) if len(sys.argv) > 2 else 100
C = int(sys.argv[3]) if len(sys.argv) > 3 else 368
sd = module_state("sgp_block", "blk", 5, C=C, ks=7, r=4)
o = pack_sgp_block(sd, "blk", C, torch.bfloat16, "cuda")
y = torch.randn((B, T, C), device="cuda").to(torch.bfloat16)
out = torch.empty_like(y)
ws = torch.empty((4, B * T, C), dtype=torch.float32, device="cuda")
with torch.cuda.stream(torch.cuda.Stream()):
    for _ in range(100):
        ops.sgp_mlp(y, o.gn_w, o.gn_b, o.w1f, o.b_fc1, o.w2f, o.b_fc2, out=out, partial=ws)
    torch.cuda.synchronize()
