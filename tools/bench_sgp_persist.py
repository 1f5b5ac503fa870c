"""GPU experiment: one level of the SGP pyramid (front -> fc1 -> fc2) as ONE persistent launch with two
device-scope grid barriers against its three launches, at cfg2's geometry (B = 8, T = 100 / 50 / 25, C = 368).  The persistent
kernel (experiments/r6_parked/sgp_level_persist.hip) runs the product kernels' own bodies, so its outputs must equal the
three launches' bit for bit.
    python tools/bench_sgp_persist.py --build        (here, no GPU: hipcc -> experiments/r6_parked/libsgp_level_persist.so)
    python tools/bench_sgp_persist.py                (on the GPU box)"""
import ctypes
import os
import subprocess
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
LIB = os.path.join(ROOT, "experiments", "r6_parked", "libsgp_level_persist.so")
if "--build" in sys.argv:
    subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-fPIC", "-std=c++17", "-ffp-contract=fast", "-shared",
                    "-I", os.path.join(ROOT, "t-deed_amd", "csrc"), "-I", os.path.join(ROOT, "include"), "-o", LIB,
                    os.path.join(ROOT, "experiments", "r6_parked", "sgp_level_persist.hip")], check=True)
    print(LIB)
    sys.exit(0)

import numpy as np
import torch
from tdeed_amd import ops, _lib
from tdeed_amd.engine import pack_mfma_frags
from tdeed_amd.regnet_spec import sgp_up_size

DEV = "cuda"
X = ctypes.CDLL(LIB)
P, I, F = ctypes.c_void_p, ctypes.c_int, ctypes.c_float
X.exp_level_persist.argtypes = [P, I, I, I, I, I, P, P, F, P, P, P, P, P, I, P, P, P, P, P, P, P, P, P, P, I, P, P]
X.exp_level_persist.restype = I
X.exp_level_max_grid.argtypes = [I]
X.exp_level_smem.argtypes = [I, I, I, I]


def ptr(t):
    return None if t is None else ctypes.c_void_p(t.data_ptr())


def graph_time(fn, reps=200):
    st = torch.cuda.current_stream()
    fn()
    st.synchronize()
    h = ctypes.c_void_p()
    _lib.call("tdeed_graph_begin", st.cuda_stream)
    try:
        fn()
    finally:
        _lib.call("tdeed_graph_end", st.cuda_stream, ctypes.byref(h))
    for _ in range(5):
        _lib.call("tdeed_graph_launch", h, st.cuda_stream)
    st.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(st)
    for _ in range(reps):
        _lib.call("tdeed_graph_launch", h, st.cuda_stream)
    b.record(st)
    st.synchronize()
    _lib.call("tdeed_graph_destroy", h)
    return a.elapsed_time(b) / reps * 1e3


side = torch.cuda.Stream()
with torch.cuda.stream(side):
    for (B, T, C, ks) in [(8, 100, 368, 7), (8, 50, 368, 7), (8, 25, 368, 7)]:
        g = torch.Generator().manual_seed(0)
        up = sgp_up_size(ks, 4)
        N1 = 4 * C
        x = torch.randn(B, T, C, generator=g).to(DEV)
        wlen = 2 * ks + up + 2
        ln_w, ln_b = (1 + 0.1 * torch.randn(C, generator=g)).to(DEV), (0.1 * torch.randn(C, generator=g)).to(DEV)
        dw, db = (torch.randn(C, wlen, generator=g) * 0.1).to(DEV), (torch.randn(5, C, generator=g) * 0.1).to(DEV)
        gn_w, gn_b = (1 + 0.1 * torch.randn(C, generator=g)).to(DEV), (0.1 * torch.randn(C, generator=g)).to(DEV)
        W1, W2 = torch.randn(N1, C, generator=g) / C ** 0.5, torch.randn(C, N1, generator=g) / N1 ** 0.5
        b1, b2 = (0.1 * torch.randn(N1, generator=g)).to(DEV), (0.1 * torch.randn(C, generator=g)).to(DEV)
        W1p, W2p = pack_mfma_frags(W1.numpy(), DEV, ks_mult=12), pack_mfma_frags(W2.numpy(), DEV, ks_mult=12)
        rowstat = torch.stack([x.mean(-1), 1.0 / torch.sqrt(x.var(-1, unbiased=False) + 1e-5)], -1).reshape(B * T, 2).contiguous()
        form = (2, 1)
        nct = ops.sgp_gemm_tiles(T, C, form)[1]

        def bufs():
            return dict(y=torch.empty_like(x), chs=torch.empty(B, C, 2, device=DEV), H=torch.empty(B, T, N1, dtype=torch.bfloat16, device=DEV),
                        out=torch.empty_like(x), rsp=torch.empty(nct, B * T, 2, device=DEV))
        a, b = bufs(), bufs()

        def three():
            ops.sgp_front(x, ks, up, ln_w, ln_b, dw, db, out=a["y"], chsum=a["chs"], rowstat=rowstat)
            ops.sgp_gemm_gn_gelu(a["y"], a["chs"], gn_w, gn_b, W1p, b1, N1, out=a["H"], form=form)
            ops.sgp_gemm_residual(a["H"], W2p, b2, a["y"], out=a["out"], rowstat_part=a["rsp"], form=form)

        counter = torch.zeros(4, dtype=torch.int32, device=DEV)
        smem = X.exp_level_smem(T, ks, up, ops.sgp_gemm_ksteps(C))
        gmax = X.exp_level_max_grid(smem)
        res = {}
        for grid in sorted({256, min(512, gmax), gmax}):
            if grid <= 0 or grid > gmax:
                continue
            dbg = torch.zeros((grid, 8), dtype=torch.int64, device=DEV)

            def one(dbgp=None):
                rc = X.exp_level_persist(ptr(x), B, T, C, ks, up, ptr(ln_w), ptr(ln_b), 1e-5, ptr(dw), ptr(db), ptr(b["y"]), ptr(b["chs"]),
                                         ptr(rowstat), 0, ptr(gn_w), ptr(gn_b), ptr(W1p), ptr(b1), ptr(b["H"]), ptr(W2p), ptr(b2),
                                         ptr(b["out"]), ptr(b["rsp"]), ptr(counter), grid, dbgp, ctypes.c_void_p(side.cuda_stream))
                assert rc == 0, rc
            three()
            one()
            side.synchronize()
            eq = {k: bool(torch.equal(a[k], b[k])) for k in a}
            t3, t1 = graph_time(three), graph_time(one)
            one(ptr(dbg))
            side.synchronize()
            d = dbg.cpu().numpy().astype(np.float64) * 10.0 / 1e3       # us
            t0 = d[:, 0].min()
            ph = ["front", "barrier 1", "fc1", "barrier 2", "fc2"]
            spans = [np.median(d[:, i + 1] - d[:, i]) for i in range(5)]
            print(f"B={B} T={T} C={C}: three launches {t3:6.2f} us   one persistent launch ({grid} workgroups, {smem} B of LDS, up to "
                  f"{gmax} resident) {t1:6.2f} us   outputs equal {eq}\n      first start -> last end {(d[:, 5].max() - t0):.2f} us; "
                  f"workgroup medians: " + ", ".join(f"{n} {v:.2f}" for n, v in zip(ph, spans)), flush=True)
