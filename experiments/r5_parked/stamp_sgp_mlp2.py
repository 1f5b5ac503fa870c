"""Where a workgroup of sgp_mlp2_kernel spends its time: in-kernel s_memtime stamps (diagnostic instantiation,
tdeed_sgp_mlp2_stamped), reported as the median / max over workgroups of each phase in shader cycles and as the start skew.
    python tools/stamp_sgp_mlp2.py [B] [T] [C] [rows]"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import torch
import tdeed_amd  # noqa: F401
from tdeed_amd import ops, _lib
from tdeed_amd.engine import pack_sgp_block
from helpers import module_state

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
T = int(sys.argv[2]) if len(sys.argv) > 2 else 100
C = int(sys.argv[3]) if len(sys.argv) > 3 else 368
rows = int(sys.argv[4]) if len(sys.argv) > 4 else 64
DEV = "cuda"
sd = module_state("pyramid", "_temp_fine", 5, C=C, ks=7, r=4, n=2)
o = pack_sgp_block(sd, "_temp_fine._sgp.0", C, torch.bfloat16, DEV)
R = B * T
y = torch.randn((B, T, C), device=DEV).to(torch.bfloat16)
chs = torch.stack([y.float().sum(1), (y.float() ** 2).sum(1)], -1).contiguous()
S = ops.sgp_mlp2_slices(C)
part = torch.empty((S, R, C), dtype=torch.float32, device=DEV)
nwg = S * ((R + rows - 1) // rows)
st = torch.zeros((nwg, 8), dtype=torch.int64, device=DEV)
# some cache-cold traffic in front, like the stage sees it
junk = torch.empty(512 << 20, dtype=torch.uint8, device=DEV)
for rep, dbg in ((0, 0), (1, 0), (2, 0), (3, 1), (4, 1), (5, 3), (6, 3)):
    if dbg == 0:
        junk.fill_(rep)
    st.zero_()
    _lib.call("tdeed_sgp_mlp2_stamped", y.data_ptr(), R, T, C, 16, o.gn_w.data_ptr(), o.gn_b.data_ptr(), 1e-5,
              o.w1p.data_ptr(), o.b1p.data_ptr(), o.w2p.data_ptr(), part.data_ptr(), chs.data_ptr(), st.data_ptr(), rows, dbg,
              torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    a = st.cpu().numpy().astype(np.int64)
    a = a[a[:, 6] != 0]                               # ids past the last slice exit without stamps
    names = ["issue loads", "GN statistics", "affine+A tile", "fc1", "fc2+1st store", "stores land"]
    d = np.diff(a[:, :7], axis=1)
    tot = a[:, 6] - a[:, 0]
    rt = (a[:, 7] - a[:, 7].min()) * 10.0          # s_memrealtime ticks at 100 MHz -> ns
    print(f"rep {rep} dbg {dbg} (1: all workgroups read slice 0's weights, 2: and row tile 0): {len(a)} workgroups; total cycles median {np.median(tot):.0f} max {tot.max()}; start skew median "
          f"{np.median(rt):.0f} ns max {rt.max():.0f} ns")
    for i, nm in enumerate(names):
        print(f"    {nm:16s} median {np.median(d[:, i]):8.0f} max {d[:, i].max():8d} cycles")
