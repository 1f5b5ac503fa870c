"""GPU micro-benchmark: a stride-1 bottleneck as one launch (tdeed_bneck_fwd) against the four launches it replaces (conv1 ->
grouped 3x3 -> SE -> conv3), at the sub-batch sizes of cfg2 (400 frames), with the kernel's phase time stamps.
    python tools/bench_bneck.py [frames]"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from tdeed_amd import ops, _lib
from tdeed_amd.engine import pack_mfma_frags, pack_gconv_frags, pack_se_mfma, pack_gsf_q_frags, pack_gsf_p_frags
from tdeed_amd.regnet_spec import gsf_fold_dim

N = int(sys.argv[1]) if len(sys.argv) > 1 else 400
DEV = "cuda"


def timeit(fn, reps=30):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3


for (h, w, C, gw, R, Fp) in [(7, 7, 368, 8, 92, 96), (14, 14, 152, 8, 38, 40)]:
    g = torch.Generator().manual_seed(1)
    hw, M = h * w, N * h * w
    x = torch.relu(torch.randn(N, h, w, C, generator=g)).to(torch.bfloat16).to(DEV)
    G = torch.randn(M, Fp, generator=g).to(torch.bfloat16).to(DEV)
    W1, W3 = torch.randn(C, C, generator=g) / C ** 0.5, torch.randn(C, C, generator=g) / C ** 0.5
    W2 = torch.randn(C, gw, 3, 3, generator=g) / (gw * 9) ** 0.5
    fc1, fc2 = torch.randn(R, C, generator=g) / C ** 0.5, torch.randn(C, R, generator=g) / R ** 0.5
    vec = lambda n, s=0.1, o=0.0: (torch.randn(n, generator=g) * s + o).to(DEV)      # noqa: E731
    s1, h1, s2, h2, s3, h3, b1, b2 = vec(C, .1, 1.), vec(C), vec(C, .1, 1.), vec(C), vec(C, .1, .5), vec(C), vec(R), vec(C)
    W1d, W3d = W1.to(torch.bfloat16).to(DEV), W3.to(torch.bfloat16).to(DEV)
    w1f, w3f = pack_mfma_frags(W1.numpy(), DEV), pack_mfma_frags(W3.numpy(), DEV)
    w2f = pack_gconv_frags(W2.numpy(), gw, DEV)
    w2f_tm = pack_gconv_frags(W2.numpy(), gw, DEV, tap_major=True)      # the one-launch form's k-slot order
    se = pack_se_mfma(fc1.numpy(), fc2.numpy(), DEV)
    y1 = torch.empty((M, C), dtype=torch.bfloat16, device=DEV)
    y2 = torch.empty((N, h, w, C), dtype=torch.bfloat16, device=DEV)
    parts = ops.gconv3x3_parts(h, w, C, 1, torch.bfloat16)
    pooled = torch.empty((N, parts, C), device=DEV)
    gate = torch.empty((N, C), device=DEV)
    out = torch.empty((M, C), dtype=torch.bfloat16, device=DEV)
    out2 = torch.empty((M, 96), dtype=torch.bfloat16, device=DEV)
    outb = torch.empty((N, h, w, C), dtype=torch.bfloat16, device=DEV)

    def chain():
        ops.gemm(x.view(M, C), W1d, s1, h1, ops.ACT_RELU, A0=G, k0=Fp, out=y1)
        ops.gconv3x3(y1.view(N, h, w, C), None, s2, h2, gw, 1, wfrag=w2f, out=y2, pooled=pooled)
        ops.se_gate_mfma(pooled, 1.0 / hw, se["w1f"], b1, se["w2f"], b2, R, out=gate)
        ops.gemm(y2.view(M, C), W3d, s3, h3, ops.ACT_RELU, residual=x.view(M, C), a_scale=gate, a_scale_rows=hw, out=out,
                 out2=out2)

    def fused():
        ops.bneck(x, w1f, s1, h1, w2f_tm, s2, h2, se["w1f"], b1, se["w2f"], b2, R, w3f, s3, h3, G=G, out=outb, out2=out2)

    t0, t1 = timeit(chain), timeit(fused)
    chain(); ref = out.clone(); fused(); torch.cuda.synchronize()
    ob = outb.view(M, C)
    ndiff = int((ref != ob).sum())
    print(f"{h}x{w}x{C} N={N}: chain {t0:7.1f} us   one launch {t1:7.1f} us   equal {torch.equal(ref, ob)}"
          f"   ({ndiff} of {ref.numel()} elements differ, max abs {float((ref.float() - ob.float()).abs().max()):.3g})", flush=True)
    nwg = (N + (1 if hw > 64 else 2) - 1) // (1 if hw > 64 else 2)
    dbg = torch.zeros((nwg, 16), dtype=torch.int64, device=DEV)
    _lib.call("tdeed_bneck_set_debug", dbg.data_ptr())
    fused(); torch.cuda.synchronize()
    _lib.call("tdeed_bneck_set_debug", None)
    d = dbg.cpu().numpy().astype(np.float64)
    # slots 8..11: wave 0's conv2 unit stamps; a form whose wave 0 owns fewer than three units leaves the later slots
    # unwritten (zero): only differences between two written stamps are reported
    def gap(a, b):
        ok = (d[:, a] > 0) & (d[:, b] > 0)
        return float(np.median(d[ok, b] - d[ok, a])) if ok.any() else None
    sub = [gap(2, 8)] + [gap(8 + j, 9 + j) for j in range(3)]
    print("   conv2 of wave 0, cycles: tap offsets + weight requests %s, units %s"
          % ("%.0f" % sub[0] if sub[0] is not None else "-", ", ".join("%.0f" % v for v in sub[1:] if v is not None)))
    d = d[(d[:, 0] > 0) & (d[:, 6] > 0)]            # rows of workgroups that ran (the row count above is an upper bound)
    # slots 12..14: inside the SE phase (stamp 3 = its start, 4 = its end): means staged, hidden units done, gates done
    se = [np.median(d[:, 12] - d[:, 3]), np.median(d[:, 13] - d[:, 12]), np.median(d[:, 14] - d[:, 13]), np.median(d[:, 4] - d[:, 14])]
    print("   SE phase, cycles: weights requested + means staged %.0f, hidden units %.0f, gates %.0f, y2 *= gate pass %.0f" % tuple(se))
    ph = np.diff(d[:, :7], axis=1) / 100.0          # clock64 ticks (shader clock, ~2.4 GHz) / 100
    # clock64 is a per-XCD counter: a span across workgroups means something only while the eight counters agree
    sp = (d[:, 6].max() - d[:, 0].min()) / 100.0
    span = f"{sp:.2f}" if sp < 100 * np.median(d[:, 6] - d[:, 0]) / 100.0 else "n/a (the XCDs' counters are not aligned)"
    names = ["load x", "conv1", "conv2", "SE + gate", "conv3", "store"]
    print("   phase cycles / 100 (median over workgroups): " + ", ".join(f"{n} {np.median(ph[:, i]):.2f}" for i, n in enumerate(names))
          + f"; workgroup total {np.median(d[:, 6] - d[:, 0]) / 100.0:.2f}; first start -> last end {span}", flush=True)

    # ---- behind a gate-shift-fuse site: blend launch + bottleneck against the bottleneck that blends in its frame load
    T = 100 if N % 100 == 0 else N
    B, F = N // T, gsf_fold_dim(C)
    xs = x[..., :Fp].contiguous()
    w3d = torch.randn(2, F // 2, 3, 3, 3, generator=g) * 0.1
    f32 = lambda *s_: (torch.randn(s_, generator=g) * 0.3).to(DEV)      # noqa: E731
    bn_s, bn_b, b3d = f32(F).abs() + 0.5, f32(F), f32(2)
    cw = [f32(18), f32(1), f32(18), f32(1)]
    wq, wqf = w3d.reshape(F, 27).t().contiguous().to(DEV), pack_gsf_q_frags(w3d.numpy(), DEV)
    bufs = dict(gate=torch.empty((N, h, w, 2), device=DEV), q=torch.empty((N, h, w, 6), device=DEV), ysum=torch.empty((N, F), device=DEV),
                xsum=torch.empty((N, F), device=DEV), out=G)
    gates = lambda: ops.gate_shift(xs, B, T, F, Fp, bn_s, bn_b, wq, b3d, *cw, wqf=wqf, bufs=bufs, gates_only=True)      # noqa: E731
    gates()

    def two():
        _lib.call("tdeed_gsf_blend_src_fwd", xs.data_ptr(), bufs["gate"].data_ptr(), bufs["ysum"].data_ptr(), bufs["xsum"].data_ptr(),
                      *[c.data_ptr() for c in cw], B, T, h, w, Fp, F, Fp, G.data_ptr(), ops.dtype_code(torch.bfloat16), ops.stream_ptr())
        fused()

    def one():
        ops.bneck_gs(x, xs, bufs["gate"], bufs["ysum"], bufs["xsum"], *cw, T, F, Fp, w1f, s1, h1, w2f_tm, s2, h2, se["w1f"], b1,
                     se["w2f"], b2, R, w3f, s3, h3, out=outb, out2=out2)

    se = pack_se_mfma(fc1.numpy(), fc2.numpy(), DEV)
    tg, t2, t1 = timeit(gates), timeit(two), timeit(one)
    two(); r2 = outb.clone(); one(); torch.cuda.synchronize()
    dbg.zero_()
    _lib.call("tdeed_bneck_set_debug", dbg.data_ptr())
    one(); torch.cuda.synchronize()
    _lib.call("tdeed_bneck_set_debug", None)
    d = dbg.cpu().numpy().astype(np.float64)
    d = d[(d[:, 0] > 0) & (d[:, 6] > 0)]
    print(f"   gate-shift-fuse site: gate launches {tg:.1f} us; blend + bottleneck {t2:.1f} us; bottleneck with the blend inside {t1:.1f} us"
          f" (equal {torch.equal(r2, outb)}); its load phase {np.median(d[:, 1] - d[:, 0]) / 100.0:.2f}, workgroup total"
          f" {np.median(d[:, 6] - d[:, 0]) / 100.0:.2f}", flush=True)
    # ... and with the next site's tap maps in the tail (stamp 7), against the gate launch's first kernel it replaces
    Qb = torch.empty((N, h, w, 6), device=DEV)
    bnq, wpf = ops.gsq_bn_table(bn_s, bn_b), pack_gsf_p_frags(w3d.numpy(), DEV)

    def one_q():
        ops.bneck_gs(x, xs, bufs["gate"], bufs["ysum"], bufs["xsum"], *cw, T, F, Fp, w1f, s1, h1, w2f_tm, s2, h2, se["w1f"], b1,
                     se["w2f"], b2, R, w3f, s3, h3, out=outb, out2=out2, qtail=(wpf, bnq, F, Qb))

    sums = lambda: ops.gate_shift(xs, B, T, F, Fp, bn_s, bn_b, wq, b3d, *cw, wqf=wqf, bufs=bufs, gates_only=True, q_given=True)   # noqa: E731
    tq, ts = timeit(one_q), timeit(sums)
    dbg.zero_()
    _lib.call("tdeed_bneck_set_debug", dbg.data_ptr())
    one_q(); torch.cuda.synchronize()
    _lib.call("tdeed_bneck_set_debug", None)
    d = dbg.cpu().numpy().astype(np.float64)
    d = d[(d[:, 0] > 0) & (d[:, 6] > 0)]
    print("   tail stamps (cycles): staging + barrier %.0f, contraction + barrier %.0f, nine-tap sums + stores %.0f" % tuple(
        np.median(d[:, b_] - d[:, a_]) for a_, b_ in ((5, 8), (8, 9), (9, 7))))
    print(f"   with the tap-map tail {tq:.1f} us (gate launches without their first kernel {ts:.1f} us); tail {np.median(d[:, 7] - d[:, 5]) / 100.0:.2f},"
          f" stores behind it {np.median(d[:, 6] - d[:, 7]) / 100.0:.2f}, workgroup total {np.median(d[:, 6] - d[:, 0]) / 100.0:.2f}", flush=True)
