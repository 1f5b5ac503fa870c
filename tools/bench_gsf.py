"""GPU tool: one gate-shift site at the three geometries of the cfg2 forward (N = 800 frames), as HIP graphs of REPS serial
copies: the whole site (3 launches), the gate half (tap maps + gate / sums), the blend alone.
    python tools/bench_gsf.py [N] [shape ...]        shape = h,C,F"""
import sys, os, ctypes
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from tdeed_amd import ops, _lib
from tdeed_amd.engine import pack_gsf_q_frags

N = int(sys.argv[1]) if len(sys.argv) > 1 else 800
shapes = [tuple(int(v) for v in a.split(",")) for a in sys.argv[2:]] or [(28, 56, 16), (14, 152, 40), (7, 368, 92)]
T = 100
B = N // T
REPS = 20
dev = "cuda"


def graph_time(fn, st):
    fn(); st.synchronize()
    h = ctypes.c_void_p()
    _lib.call("tdeed_graph_begin", st.cuda_stream)
    try:
        for _ in range(REPS):
            fn()
    finally:
        _lib.call("tdeed_graph_end", st.cuda_stream, ctypes.byref(h))
    for _ in range(3):
        _lib.call("tdeed_graph_launch", h, st.cuda_stream)
    st.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(st)
    for _ in range(10):
        _lib.call("tdeed_graph_launch", h, st.cuda_stream)
    b.record(st)
    st.synchronize()
    _lib.call("tdeed_graph_destroy", h)
    return a.elapsed_time(b) / 10 / REPS * 1e3


st = torch.cuda.Stream()
with torch.cuda.stream(st):
    for (h, C, F) in shapes:
        g = torch.Generator(device="cpu").manual_seed(h * 1000 + F)
        x = (torch.randn((N, h, h, C), generator=g) * 0.5).to(dev).to(torch.bfloat16)
        Fp = (F + 7) // 8 * 8
        w3d = torch.randn((2, F // 2, 3, 3, 3), generator=g) * 0.1
        wqf = pack_gsf_q_frags(w3d, dev)
        f32 = lambda *s: (torch.randn(s, generator=g) * 0.3).to(dev)
        bn_s, bn_b, b3d = f32(F).abs() + 0.5, f32(F), f32(2)
        cw1, cb1, cw2, cb2 = f32(2, 3, 3), f32(1), f32(2, 3, 3), f32(1)
        bufs = dict(gate=torch.empty((N, h, h, 2), device=dev), ysum=torch.empty((N, F), device=dev),
                    xsum=torch.empty((N, F), device=dev), out=torch.empty((N * h * h, Fp), device=dev, dtype=torch.bfloat16),
                    q=torch.empty((N, h, h, 6), device=dev))
        dc = ops.dtype_code(x.dtype)
        P = ops.ptr

        def gate():
            _lib.call("tdeed_gsf_gate_fwd", P(x), B, T, h, h, C, F, P(bn_s), P(bn_b), None, P(wqf), P(b3d), P(bufs["q"]),
                      P(bufs["gate"]), P(bufs["ysum"]), P(bufs["xsum"]), dc, st.cuda_stream)

        def apply():
            _lib.call("tdeed_gsf_apply_fused_fwd", P(x), P(bufs["gate"]), P(bufs["ysum"]), P(bufs["xsum"]), P(cw1), P(cb1),
                      P(cw2), P(cb2), B, T, h, h, C, F, Fp, P(bufs["out"]), dc, st.cuda_stream)

        def blend_src():
            _lib.call("tdeed_gsf_blend_src_fwd", P(x), P(bufs["gate"]), P(bufs["ysum"]), P(bufs["xsum"]), P(cw1), P(cb1),
                      P(cw2), P(cb2), B, T, h, h, C, F, Fp, P(bufs["out"]), dc, st.cuda_stream)

        def site():
            gate(); apply()

        def site_src():
            gate(); blend_src()

        site(); st.synchronize()
        chk = float(bufs["out"].float().abs().sum())
        print(f"{h}x{h} C={C} F={F} N={N}: site {graph_time(site, st):6.1f} us   tap maps + gate/sums {graph_time(gate, st):6.1f} us   "
              f"blend {graph_time(apply, st):6.1f} us   (checksum {chk:.6e})")
        print(f"      source-order blend {graph_time(blend_src, st):6.1f} us   site with it {graph_time(site_src, st):6.1f} us")
