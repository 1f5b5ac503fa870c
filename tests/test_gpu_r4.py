"""Round-4 parity tests on the MI355X (through the C ABI).  -m gpu only.

* the execution shape bench.py times (one sub-batch, three plans = three buffer sets / HIP graphs rotated over three
  streams) gives, per slot, the logits of the per-clip B=1 forwards and the reference's golden logits;
* the restructured training backward (SE + BatchNorm backward from per-frame sums; ReLU mask and BatchNorm statistics in
  the producing contraction's epilogue) against the launch-per-op chain it replaces."""
import numpy as np
import pytest
import torch

from helpers import load_golden, model_state, t, max_abs
from tdeed_amd import synth

pytestmark = pytest.mark.gpu
DEV = "cuda"


def test_three_rotated_slots_of_the_timed_shape_equal_single_clip_forwards():
    """BASELINE configs[1] as bench.py runs it (bench.py main(): n_split=1, plan(B, H, W, slot=i) for i < 3, replayed
    rotated on three streams): after 7 rotated steps every slot still holds the logits of ITS OWN clips -- equal to the
    B=1 forward of each clip (bf16 <= 2e-2) and, for the golden clip in slot 0, within the bf16 bound of the reference's
    logits (model/model.py:105-149)."""
    from tdeed_amd.engine import ForwardEngine
    meta, g = load_golden("finediving_small")
    cfg = meta["cfg"]
    sd = model_state(cfg, meta["seed_w"])
    T, H, W = cfg["clip_len"], meta["H"], meta["W"]
    B, depth = 8, 3
    K1 = cfg["num_classes"] + 1
    streams = [torch.cuda.Stream() for _ in range(depth)]
    clips = [np.concatenate([synth.uint8_clip(meta["seed_x"] + 100 * s + i, (1, T, 3, H, W)) for i in range(B)], 0)
             for s in range(depth)]
    with torch.cuda.stream(streams[0]):
        eng = ForwardEngine(cfg, sd, torch.bfloat16, DEV, n_split=1)
        plans = [eng.plan(B, H, W, slot=i) for i in range(depth)]
        assert all(len(p.subs) == 1 for p in plans)
        assert len({p.head_out.data_ptr() for p in plans}) == depth
        for p, c in zip(plans, clips):
            eng.set_frames(p, t(c).to(DEV))
    torch.cuda.synchronize()
    for i in range(7 * depth):                                       # the loop of bench.py's run(n)
        with torch.cuda.stream(streams[i % depth]):
            eng.run_plan(plans[i % depth])
    torch.cuda.synchronize()
    assert all(p.graph is not None for p in plans)
    heads = [p.head_out.float().cpu().view(B, T, -1).clone() for p in plans]
    for s in range(depth):
        assert torch.isfinite(heads[s]).all()
        for o in range(s):
            assert max_abs(heads[s], heads[o]) > 1e-2                 # different clips: different logits (no aliasing)
    # per-clip B=1 forwards on a fresh engine (its own buffers)
    eng1 = ForwardEngine(cfg, sd, torch.bfloat16, DEV, n_split=1)
    st = torch.cuda.Stream()
    for s in range(depth):
        for i in (0, 3, 7):
            with torch.cuda.stream(st):
                h1, _ = eng1.forward(t(clips[s][i:i + 1]).to(DEV))
                st.synchronize()
            assert max_abs(h1.float().cpu().view(T, -1), heads[s][i]) <= 2e-2, (s, i)
    err = max_abs(heads[0][0, :, :K1], g["logits"][0])
    assert err < 0.08 * max(1.0, float(np.abs(g["logits"]).max()))


def _rand(shape, seed, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(shape, generator=g) * scale)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("N,h,w,C,R", [(6, 7, 7, 48, 12), (5, 14, 14, 152, 38), (3, 9, 5, 768, 192), (2, 56, 56, 64, 8)])
def test_se_and_conv2_batchnorm_backward_from_frame_sums_equals_the_pass_per_op_chain(dtype, N, h, w, C, R):
    """tdeed_se_bn_bwd_{sums,finalize,apply} (five per-frame sums, no d_y2 map) against the chain it replaces
    (tdeed_pool_rows -> tdeed_se_train_bwd -> tdeed_scale_rows -> tdeed_bn_train_bwd) and, in fp32, against autograd on
    the plain torch expression of timm's SEModule + BatchNorm2d + ReLU."""
    from tdeed_amd import ops_bwd as B_
    z = _rand((N, h, w, C), 1).to(DEV).to(dtype)
    d = _rand((N, h, w, C), 2, 0.5).to(DEV).to(dtype)
    bw = (_rand((C,), 3, 0.3) + 1.0).to(DEV)
    bb = _rand((C,), 4, 0.3).to(DEV)
    w1, b1 = _rand((R, C), 5, 0.2).to(DEV), _rand((R,), 6, 0.1).to(DEV)
    w2, b2 = _rand((C, R), 7, 0.2).to(DEV), _rand((C,), 8, 0.1).to(DEV)
    mean, rstd, fa, fb = B_.bn_stats(z, bw, bb)
    bn = (mean, rstd, fa, fb)
    p = B_.pool_rows(z, affine=(fa, fb))
    hid, gate = B_.se_train_fwd(p, w1.t().contiguous(), b1, w2.t().contiguous(), b2)
    # the chain
    d_gate = B_.pool_rows(d, z, affine=(fa, fb), affine_on=2)
    d_pre2, d_hid, d_p = B_.se_train_bwd(d_gate, gate, hid, w1, w2)
    d_y2 = B_.scale_rows(d, gate, add=d_p, add_scale=1.0 / (h * w))
    dz_ref, _, dw_ref, db_ref = B_.bn_train_bwd(z, d_y2, None, bn, bw, relu=True)
    # the fused form
    dz, dw, db, d_pre2_f, d_hid_f = B_.se_bn_bwd(d, z, bn, bw, gate, hid, w1, w2)
    torch.cuda.synchronize()
    tol = 2e-5 if dtype == torch.float32 else 2e-2

    def close(a, b, name, t_=tol):
        a, b = a.float().cpu(), b.float().cpu()
        sc = max(1e-6, float(b.abs().max()))
        assert float((a - b).abs().max()) <= t_ * sc, (name, float((a - b).abs().max()), sc)

    close(d_pre2_f, d_pre2, "d_pre2", 2e-5 if dtype == torch.float32 else 1e-4)
    close(d_hid_f, d_hid, "d_hid", 2e-5 if dtype == torch.float32 else 1e-4)
    close(dz, dz_ref, "dz")
    # (bf16: the chain rounds d_y2 to bf16 before the sums, the fused form keeps it in fp32)
    close(dw, dw_ref, "dw", tol if dtype == torch.float32 else 1.5e-2)
    close(db, db_ref, "db", tol if dtype == torch.float32 else 1.5e-2)
    if dtype == torch.float32:
        zc = z.cpu().double().requires_grad_(True)
        wc, bc = bw.cpu().double().requires_grad_(True), bb.cpu().double().requires_grad_(True)
        mu = zc.mean((0, 1, 2))
        var = zc.var((0, 1, 2), unbiased=False)
        y = torch.relu((zc - mu) / torch.sqrt(var + 1e-5) * wc + bc)
        pp = y.mean((1, 2))
        gt = torch.sigmoid(torch.relu(pp @ w1.cpu().double().t() + b1.cpu().double()) @ w2.cpu().double().t() + b2.cpu().double())
        (y * gt[:, None, None, :] * d.cpu().double()).sum().backward()
        close(dz, zc.grad, "dz vs autograd", 1e-4)
        close(dw, wc.grad, "dw vs autograd", 1e-4)
        close(db, bc.grad, "db vs autograd", 1e-4)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("frames,hi,wi,K,N,n2,res", [(9, 7, 5, 64, 32, 0, "s2"), (4, 14, 14, 152, 152, 40, "full"),
                                                     (3, 9, 9, 320, 128, 0, "s2"), (2, 28, 28, 128, 128, 0, "full"),
                                                     (5, 6, 6, 48, 56, 16, "none")])
def test_input_gradient_contraction_with_the_gradient_sink_epilogue(dtype, frames, hi, wi, K, N, n2, res):
    """tdeed_gemm_dgrad: residual (full / on the even pixels of a stride-2 block), compact second output before the residual,
    ReLU mask of the block in front and the per-tile column sums of the BatchNorm backward behind it, against torch; then
    tdeed_gsf_add_cols_sink on top and tdeed_bn_bwd_from_parts against the masked-statistics pass (tdeed_bn_train_bwd)."""
    from tdeed_amd import ops_bwd as B_
    M = frames * hi * wi
    f32 = dtype == torch.float32
    dz = _rand((M, K), 11, 0.5).to(DEV).to(dtype)
    wt = _rand((N, K), 12, 0.2).to(DEV).to(dtype)
    mask = _rand((M, N), 13).to(DEV).to(dtype)
    z = _rand((M, N), 14).to(DEV).to(dtype)
    zd = _rand((M, N), 15).to(DEV).to(dtype)
    mean, mean_d = _rand((N,), 16, 0.2).to(DEV), _rand((N,), 17, 0.2).to(DEV)
    ho, wo = (hi + 1) // 2, (wi + 1) // 2
    R = r_hw = None
    if res == "full":
        R = _rand((M, N), 18, 0.5).to(DEV).to(dtype)
    elif res == "s2":
        R = _rand((frames * ho * wo, N), 18, 0.5).to(DEV).to(dtype)
        r_hw = (hi, wi)
    out2 = torch.empty((M, n2), dtype=dtype, device=DEV) if n2 else None
    sink = B_.GradSink(mask, z, mean, zd=zd, mean_d=mean_d)
    dx = B_.gemm_dgrad(dz, wt, sink=sink, residual=R, r_hw=r_hw, out2=out2)
    torch.cuda.synchronize()
    acc = dz.double() @ wt.double().t()
    if n2:
        ref2 = acc[:, :n2].clone()
        acc[:, :n2] = 0
        assert max_abs(out2, ref2.to(dtype)) <= (1e-4 if f32 else 3e-2) * max(1.0, float(ref2.abs().max()))
    if res == "full":
        acc = acc + R.double()
    elif res == "s2":
        full = torch.zeros((frames, hi, wi, N), dtype=torch.float64, device=DEV)
        full[:, ::2, ::2] = R.double().view(frames, ho, wo, N)
        acc = acc + full.view(M, N)
    ref = torch.where(mask.double() > 0, acc, torch.zeros_like(acc))
    tol = (1e-4 if f32 else 3e-2) * max(1.0, float(ref.abs().max()))
    assert max_abs(dx, ref) <= tol
    # column sums of what was stored
    st = dx.double()
    sums = sink.partA.double().sum(0)
    want = torch.stack([st.sum(0), (st * (z.double() - mean.double())).sum(0), (st * (zd.double() - mean_d.double())).sum(0)])
    assert max_abs(sums, want) <= 1e-3 * max(1.0, float(want.abs().max()))
    # the gate-shift columns joining afterwards
    C, Fp = N, 16
    a = _rand((M, Fp), 19, 0.3).to(DEV).to(dtype)
    b = _rand((M, Fp), 20, 0.3).to(DEV).to(dtype)
    dx0 = dx.clone()
    B_.gsf_add_cols_sink(a, b, dx, Fp, sink)
    torch.cuda.synchronize()
    add = torch.where(mask[:, :Fp].double() > 0, a.double() + b.double(), torch.zeros((M, Fp), dtype=torch.float64, device=DEV))
    ref_cols = (dx0[:, :Fp].double() + add).to(dtype)
    assert max_abs(dx[:, :Fp], ref_cols) <= (1e-6 if f32 else 2e-2) * max(1.0, float(ref_cols.abs().max()))
    assert max_abs(dx[:, Fp:], dx0[:, Fp:]) == 0.0
    # BatchNorm backward from the two partial sets == the statistics pass over the finished map
    bw = (_rand((C,), 21, 0.3) + 1.0).to(DEV)
    rstd = (_rand((C,), 22, 0.1).abs() + 0.5).to(DEV)
    for q, zz, mm in ((1, z, mean), (2, zd, mean_d)):
        dz_f, dw_f, db_f = B_.bn_bwd_from_parts(zz, dx, (mm, rstd), bw, sink, q=q)
        dz_r, _, dw_r, db_r = B_.bn_train_bwd(zz, dx, None, (mm, rstd), bw, relu=False)
        torch.cuda.synchronize()
        sc = max(1.0, float(dw_r.abs().max()), float(db_r.abs().max()))
        assert max_abs(dw_f, dw_r) <= 2e-4 * sc and max_abs(db_f, db_r) <= 2e-4 * sc, q
        assert max_abs(dz_f, dz_r) <= (1e-4 if f32 else 2e-2) * max(1.0, float(dz_r.float().abs().max())), q


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_shortcut_batchnorm_inside_conv3s_apply_pass_equals_two_passes(dtype):
    """tdeed_bn_apply2 (y = relu(z a + b + (zd ra + rb))) == tdeed_bn_apply(zd) followed by tdeed_bn_apply(z, res), bitwise."""
    from tdeed_amd import ops_bwd as B_
    from tdeed_amd._lib import call, ptr, stream_ptr, dtype_code
    M, C = 1000, 152
    z, zd = _rand((M, C), 31).to(DEV).to(dtype), _rand((M, C), 32).to(DEV).to(dtype)
    a, b, ra, rb = (_rand((C,), 33 + i, 0.5).to(DEV) for i in range(4))
    sc = B_.bn_apply(zd, ra, rb, relu=False)
    ref = B_.bn_apply(z, a, b, res=sc, relu=True)
    out = torch.empty_like(z)
    call("tdeed_bn_apply2", ptr(z), M, C, ptr(a), ptr(b), ptr(zd), ptr(ra), ptr(rb), 1, ptr(out), dtype_code(dtype), stream_ptr())
    torch.cuda.synchronize()
    assert torch.equal(out, ref)


def test_register_stationary_contraction_with_the_statistics_epilogue_equals_the_tiled_kernel():
    """tdeed_gemm_rs_stats_fwd (K = N = 320, W in registers, per-workgroup column sums) against tdeed_gemm_fwd(colpart): the
    same rounded outputs (same k order per MFMA chain is not guaranteed: <= 1 bf16 ulp) and the same BatchNorm statistics;
    weights packed by repack.pack_ws == engine.pack_ws_weights."""
    from tdeed_amd import ops, repack as R
    from tdeed_amd.engine import pack_ws_weights
    M, K = 70000 + 37, 320
    A = _rand((M, K), 41).to(DEV).to(torch.bfloat16)
    G = _rand((M, 80), 42).to(DEV).to(torch.bfloat16)
    W = _rand((K, K), 43, 0.05).to(DEV)
    wf = R.pack_ws(W).to(torch.bfloat16)
    assert torch.equal(wf.cpu(), pack_ws_weights(W.cpu().numpy(), torch.bfloat16, "cpu"))
    Wb = W.to(torch.bfloat16)
    for A0, k0 in ((None, 0), (G, 80)):
        z, (ps, pq, stride, P) = ops.gemm_rs_stats(A, wf, K, K, M=M, A0=A0, k0=k0)
        cp = torch.empty((ops.gemm_colpart_rows(M), 2, K), dtype=torch.float32, device=DEV)
        zr = ops.gemm(A, Wb, None, None, ops.ACT_NONE, M=M, colpart=cp, A0=A0, k0=k0)
        torch.cuda.synchronize()
        assert max_abs(z, zr) <= 2e-2 * max(1.0, float(zr.float().abs().max()))
        s1 = ps.view(P, 2 * K)[:, :K].double().sum(0)
        s2 = ps.view(P, 2 * K)[:, K:].double().sum(0)
        zf = z.double()
        assert max_abs(s1, zf.sum(0)) <= 1e-3 * max(1.0, float(zf.sum(0).abs().max()))
        assert max_abs(s2, (zf * zf).sum(0)) <= 1e-3 * float((zf * zf).sum(0).abs().max())
    # the plain form serves conv3's input gradient
    d = ops.gemm_rs(A, wf, K, K, M=M)
    torch.cuda.synchronize()
    assert max_abs(d, ops.gemm(A, Wb, None, None, ops.ACT_NONE, M=M)) <= 2e-2 * max(1.0, float(d.float().abs().max()))


def test_weight_gradient_on_exact_160_wide_tiles_for_the_320_wide_layers():
    """tdeed_wgrad at N = K = 320 (wgrad_tr160_kernel: 2 x 2 exact tiles) against the fp64 product, with and without the
    gate-shift splice of the X operand, on a row count that is not a multiple of the 64-row chunk."""
    from tdeed_amd import ops_bwd as B_
    M, N = 9000 + 13, 320
    dY = _rand((M, N), 51, 0.5).to(DEV).to(torch.bfloat16)
    X = _rand((M, N), 52).to(DEV).to(torch.bfloat16)
    X0 = _rand((M, 80), 53).to(DEV).to(torch.bfloat16)
    for x0, k0 in ((None, 0), (X0, 80)):
        dW, _ = B_.wgrad(dY, X, with_bias=False, M=M, X0=x0, k0=k0)
        torch.cuda.synchronize()
        Xe = X.double().clone()
        if x0 is not None:
            Xe[:, :k0] = x0.double()
        ref = dY.double().t() @ Xe
        assert max_abs(dW, ref) <= 2e-4 * float(ref.abs().max())


@pytest.mark.parametrize("N,h,w,C,gw", [(5, 14, 14, 320, 16), (3, 28, 28, 128, 16), (4, 7, 7, 368, 8), (2, 14, 14, 152, 8)])
def test_conv1_batchnorm_backward_statistics_from_conv2s_input_gradient_launch(N, h, w, C, gw):
    """tdeed_gconv3x3_dgrad_stats + tdeed_bn_bwd_masked_from_parts against the separate launches they replace (the grouped conv
    of dz2 with the flipped weights, then tdeed_bn_train_bwd's masked statistics pass + apply pass over (d_y1, z1))."""
    from tdeed_amd import ops, ops_bwd as B_
    from tdeed_amd.engine import gconv_frags_on_device
    G = C // gw
    w2 = _rand((C, gw, 3, 3), 61, 0.2).to(DEV)
    wt = w2.reshape(G, gw, gw, 3, 3).transpose(1, 2).flip(3, 4).reshape(C, gw, 3, 3).contiguous()
    wfrag_t = gconv_frags_on_device(wt, gw)
    one, zero = torch.ones(C, device=DEV), torch.zeros(C, device=DEV)
    dz2 = _rand((N, h, w, C), 62, 0.5).to(DEV).to(torch.bfloat16)
    z1 = _rand((N, h, w, C), 63).to(DEV).to(torch.bfloat16)
    bw = (_rand((C,), 64, 0.3) + 1.0).to(DEV)
    bb = _rand((C,), 65, 0.3).to(DEV)
    bn1 = B_.bn_stats(z1, bw, bb)
    d_ref, _ = ops.gconv3x3(dz2, None, one, zero, gw, 1, wfrag=wfrag_t, relu=False)
    dz_ref, _, dw_ref, db_ref = B_.bn_train_bwd(z1, d_ref, None, bn1, bw, relu=True)
    d_y1, part = B_.gconv3x3_dgrad_stats(dz2, wfrag_t, one, zero, gw, z1, bn1)
    dz, dw, db = B_.bn_bwd_masked_from_parts(z1, d_y1, bn1, bw, part)
    torch.cuda.synchronize()
    assert torch.equal(d_y1, d_ref)
    sc = max(1.0, float(dw_ref.abs().max()), float(db_ref.abs().max()))
    assert max_abs(dw, dw_ref) <= 2e-4 * sc and max_abs(db, db_ref) <= 2e-4 * sc
    assert max_abs(dz, dz_ref) <= 1e-2 * max(1.0, float(dz_ref.float().abs().max()))


@pytest.mark.parametrize("N,Hi,Wi,C,gw", [(3, 28, 28, 128, 16), (2, 56, 56, 64, 16), (4, 14, 14, 368, 8), (2, 30, 18, 56, 8)])
def test_stride2_input_gradient_launch_leaves_conv1_batchnorm_backward_statistics(N, Hi, Wi, C, gw):
    """tdeed_gconv3x3_bwd_stats (stride 2): dx and dw equal tdeed_gconv3x3_bwd's, and the partial rows fold to the sums of
    tdeed_bn_train_bwd's masked statistics pass over (dx, z1)."""
    from tdeed_amd import ops_bwd as B_
    assert B_.gconv3x3_bwd_stats_fits(N, Hi, Wi, C, gw)
    G = C // gw
    Ho, Wo = (Hi - 1) // 2 + 1, (Wi - 1) // 2 + 1
    w2p = _rand((G, 9, gw, gw), 71, 0.2).to(DEV)
    x = torch.relu(_rand((N, Hi, Wi, C), 72)).to(DEV).to(torch.bfloat16)
    dy = _rand((N, Ho, Wo, C), 73, 0.5).to(DEV).to(torch.bfloat16)
    z1 = _rand((N, Hi, Wi, C), 74).to(DEV).to(torch.bfloat16)
    bw = (_rand((C,), 75, 0.3) + 1.0).to(DEV)
    bb = _rand((C,), 76, 0.3).to(DEV)
    bn1 = B_.bn_stats(z1, bw, bb)
    dx_ref, dw_ref = B_.gconv3x3_bwd(x, dy, w2p, gw, 2)
    dz_ref, _, dwb_ref, dbb_ref = B_.bn_train_bwd(z1, dx_ref, None, bn1, bw, relu=True)
    dx, dw, part = B_.gconv3x3_bwd_stats(x, dy, w2p, gw, z1, bn1)
    dz, dwb, dbb = B_.bn_bwd_masked_from_parts(z1, dx, bn1, bw, part)
    torch.cuda.synchronize()
    assert torch.equal(dx, dx_ref) and torch.equal(dw, dw_ref)
    sc = max(1.0, float(dwb_ref.abs().max()), float(dbb_ref.abs().max()))
    assert max_abs(dwb, dwb_ref) <= 2e-4 * sc and max_abs(dbb, dbb_ref) <= 2e-4 * sc
    assert max_abs(dz, dz_ref) <= 1e-2 * max(1.0, float(dz_ref.float().abs().max()))


@pytest.mark.parametrize("Co,Ci,frames,hi,wi,res", [(64, 32, 3, 10, 10, "s2"), (128, 64, 2, 12, 8, "s2"), (128, 128, 5, 7, 9, "full"),
                                                     (128, 128, 2, 16, 16, "full"), (64, 32, 1, 8, 8, "none")])
def test_narrow_conv1_backward_in_one_launch_equals_the_three_launch_chain(Co, Ci, frames, hi, wi, res):
    """tdeed_narrow_conv1_bwd (BatchNorm + ReLU backward, input gradient with the gradient-sink epilogue, weight gradient; dz1
    never stored) against tdeed_bn_bwd_masked_from_parts -> tdeed_gemm_dgrad -> tdeed_wgrad on the same operands."""
    from tdeed_amd import ops_bwd as B_
    assert B_.narrow_conv1_bwd_fits(Co, Ci, torch.bfloat16)
    M = frames * hi * wi
    bf = torch.bfloat16
    d_y1 = _rand((M, Co), 81, 0.5).to(DEV).to(bf)
    z1 = _rand((M, Co), 82).to(DEV).to(bf)
    x = torch.relu(_rand((M, Ci), 83)).to(DEV).to(bf)
    W1 = _rand((Co, Ci), 84, 0.2).to(DEV).to(bf)
    wt = W1.t().contiguous()
    bw = (_rand((Co,), 85, 0.3) + 1.0).to(DEV)
    bb = _rand((Co,), 86, 0.3).to(DEV)
    bn1 = B_.bn_stats(z1, bw, bb)
    zp, zdp = _rand((M, Ci), 87).to(DEV).to(bf), _rand((M, Ci), 88).to(DEV).to(bf)
    mp, mdp = _rand((Ci,), 89, 0.2).to(DEV), _rand((Ci,), 90, 0.2).to(DEV)
    ho, wo = (hi + 1) // 2, (wi + 1) // 2
    R = r_hw = None
    if res == "full":
        R = _rand((M, Ci), 91, 0.5).to(DEV).to(bf)
    elif res == "s2":
        R, r_hw = _rand((frames * ho * wo, Ci), 91, 0.5).to(DEV).to(bf), (hi, wi)
    # the chain
    dz_ref, _, dwb_ref, dbb_ref = B_.bn_train_bwd(z1, d_y1, None, bn1, bw, relu=True)
    s_ref = B_.GradSink(x, zp, mp, zd=zdp, mean_d=mdp)
    dx_ref = B_.gemm_dgrad(dz_ref, wt, sink=s_ref, residual=R, r_hw=r_hw)
    dW_ref, _ = B_.wgrad(dz_ref, x, with_bias=False, M=M)
    # one launch; the statistics partials as conv2's input-gradient launch would leave them (one row here)
    part = (dbb_ref.view(1, Co).contiguous(), (dwb_ref / bn1[1]).view(1, Co).contiguous(), Co, 1)
    s_new = B_.GradSink(x, zp, mp, zd=zdp, mean_d=mdp)
    dx, dW, dwb, dbb = B_.narrow_conv1_bwd(d_y1, z1, bn1, bw, part, x, wt, sink=s_new, residual=R, r_hw=r_hw)
    dW = B_.materialize(dW)
    torch.cuda.synchronize()
    assert max_abs(dwb, dwb_ref) <= 1e-4 * max(1.0, float(dwb_ref.abs().max()))
    assert max_abs(dbb, dbb_ref) <= 1e-4 * max(1.0, float(dbb_ref.abs().max()))
    assert max_abs(dx, dx_ref) <= 2e-2 * max(1.0, float(dx_ref.float().abs().max()))
    assert max_abs(dW, dW_ref) <= 5e-3 * max(1.0, float(dW_ref.abs().max()))
    a, b = s_new.partA.double().sum(0), s_ref.partA.double().sum(0)
    assert max_abs(a, b) <= 5e-3 * max(1.0, float(b.abs().max()))
    # the sums are those of what was stored
    st = dx.double()
    want = torch.stack([st.sum(0), (st * (zp.double() - mp.double())).sum(0), (st * (zdp.double() - mdp.double())).sum(0)])
    assert max_abs(a, want) <= 1e-3 * max(1.0, float(want.abs().max()))
    # z1 recomputed inside the launch (Z = NULL) when it is the raw product of the operands the launch holds: the same results
    # as reading that map
    if not (Co == 128 and Ci == 128):
        from tdeed_amd import ops
        z_raw = ops.gemm(x, W1, None, None, ops.ACT_NONE)
        bn_r = B_.bn_stats(z_raw, bw, bb)
        _, _, dwb_r, dbb_r = B_.bn_train_bwd(z_raw, d_y1, None, bn_r, bw, relu=True)
        part_r = (dbb_r.view(1, Co).contiguous(), (dwb_r / bn_r[1]).view(1, Co).contiguous(), Co, 1)
        outs = []
        for rc in (False, True):
            sk = B_.GradSink(x, zp, mp, zd=zdp, mean_d=mdp)
            dx_r, dW_r, _, _ = B_.narrow_conv1_bwd(d_y1, z_raw, bn_r, bw, part_r, x, wt, sink=sk, residual=R, r_hw=r_hw,
                                                   recompute=rc)
            outs.append((dx_r, B_.materialize(dW_r), sk.partA.clone()))
        torch.cuda.synchronize()
        assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1]) and torch.equal(outs[0][2], outs[1][2])


@pytest.mark.parametrize("n2", [0, 80])
def test_register_stationary_input_gradient_with_the_sink_epilogue_equals_the_tiled_kernel(n2):
    """tdeed_gemm_dgrad_rs (K = N = 320 over many rows: residual, mask, compact second output before the residual, one-map sink
    statistics per persistent workgroup) against tdeed_gemm_dgrad on the same operands."""
    from tdeed_amd import ops_bwd as B_, repack as R
    M, K = 70000 + 21, 320
    bf = torch.bfloat16
    dz = _rand((M, K), 61, 0.5).to(DEV).to(bf)
    Wt = _rand((K, K), 62, 0.05).to(DEV)                       # the [N][K] matrix of the product
    wt, wt_ws = Wt.to(bf), R.pack_ws(Wt).to(bf)
    res = _rand((M, K), 63, 0.5).to(DEV).to(bf)
    mask = torch.relu(_rand((M, K), 64)).to(DEV).to(bf)
    z, mean = _rand((M, K), 65).to(DEV).to(bf), _rand((K,), 66, 0.2).to(DEV)
    outs = []
    for ws in (None, wt_ws):
        sink = B_.GradSink(mask, z, mean)
        o2 = torch.empty((M, n2), dtype=bf, device=DEV) if n2 else None
        dx = B_.gemm_dgrad(dz, wt, sink=sink, residual=res, out2=o2, wt_ws=ws)
        outs.append((dx, o2, sink.partA.double().sum(0)))
    torch.cuda.synchronize()
    (dx0, o20, p0), (dx1, o21, p1) = outs
    assert p1.shape == p0.shape
    assert max_abs(dx1, dx0) <= 2e-2 * max(1.0, float(dx0.float().abs().max()))
    if n2:
        assert max_abs(o21, o20) <= 2e-2 * max(1.0, float(o20.float().abs().max()))
        assert torch.equal(dx1[:, :n2], (res[:, :n2].float() * (mask[:, :n2] > 0)).to(bf))
    # the sums are those of what was stored
    st = dx1.double()
    want = torch.stack([st.sum(0), (st * (z.double() - mean.double())).sum(0)])
    assert max_abs(p1[:2], want) <= 1e-3 * max(1.0, float(want.abs().max()))
    assert float(p1[2].abs().max()) == 0.0
