"""Round 6, GPU: the one-launch bottleneck with the gate-shift-fuse blend inside its frame load (tdeed_bneck_gs_fwd) against the
two launches it replaces (tdeed_gsf_blend_src_fwd -> tdeed_bneck_fwd)."""
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _block(g, h, w, C, gw, R):
    from tdeed_amd.engine import pack_mfma_frags, pack_gconv_frags, pack_se_mfma
    W1 = torch.randn(C, C, generator=g) / C ** 0.5
    W3 = torch.randn(C, C, generator=g) / C ** 0.5
    W2 = torch.randn(C, gw, 3, 3, generator=g) / (gw * 9) ** 0.5
    fc1 = torch.randn(R, C, generator=g) / C ** 0.5
    fc2 = torch.randn(C, R, generator=g) / R ** 0.5
    vec = lambda n, s_=0.1, o=0.0: (torch.randn(n, generator=g) * s_ + o).to(DEV)          # noqa: E731
    s1, h1, s2, h2, s3, h3 = vec(C, 0.1, 1.0), vec(C), vec(C, 0.1, 1.0), vec(C), vec(C, 0.1, 0.5), vec(C)
    b1, b2 = vec(R), vec(C)
    se = pack_se_mfma(fc1.numpy(), fc2.numpy(), DEV)
    return (pack_mfma_frags(W1.numpy(), DEV), s1, h1, pack_gconv_frags(W2.numpy(), gw, DEV, tap_major=True), s2, h2,
            se["w1f"], b1, se["w2f"], b2, R, pack_mfma_frags(W3.numpy(), DEV), s3, h3)


@pytest.mark.parametrize("compact", [True, False])
@pytest.mark.parametrize("h,w,C,R,B,T", [(7, 7, 368, 92, 3, 7), (14, 14, 152, 38, 2, 5), (7, 7, 368, 92, 8, 100), (14, 14, 152, 38, 1, 1),
                                         (7, 7, 368, 92, 5, 1), (5, 7, 152, 38, 3, 3), (13, 7, 368, 92, 2, 3)])
def test_bottleneck_with_the_blend_inside_equals_blend_then_bottleneck(h, w, C, R, B, T, compact):
    """out == bneck(x, G = gate_shift(slice, src_order=True)) bit for bit: clips of odd length (a workgroup's two frames in
    different clips), an odd frame count (a last workgroup with one frame), one-frame clips (no neighbour on either side), the
    slice read from the block input itself or from its compact copy, all four instances of the launch, the cfg2 batch."""
    from tdeed_amd import ops
    from tdeed_amd.engine import pack_gsf_q_frags
    from tdeed_amd.regnet_spec import gsf_fold_dim
    g = torch.Generator().manual_seed(h * 1000 + C + B * 10 + T)
    F = gsf_fold_dim(C)
    Fp = (F + 7) // 8 * 8
    N = B * T
    assert ops.bneck_fits(h, w, C, R)
    x = torch.relu(torch.randn(N, h, w, C, generator=g)).to(torch.bfloat16).to(DEV)
    xs = x[..., :Fp].contiguous() if compact else x
    w3d = torch.randn(2, F // 2, 3, 3, 3, generator=g) * 0.1
    f32 = lambda *s: (torch.randn(s, generator=g) * 0.3).to(DEV)      # noqa: E731
    bn_s, bn_b, b3d = f32(F).abs() + 0.5, f32(F), f32(2)
    cw = [f32(18), f32(1), f32(18), f32(1)]
    wq, wqf = w3d.reshape(F, 27).t().contiguous().to(DEV), pack_gsf_q_frags(w3d.numpy(), DEV)
    blk = _block(g, h, w, C, 8, R)
    G = ops.gate_shift(xs, B, T, F, Fp, bn_s, bn_b, wq, b3d, *cw, wqf=wqf, src_order=True)
    assert float((G.float() - xs.reshape(-1, xs.shape[-1])[:, :Fp].float()).abs().max()) > 0.05      # the blend does something
    out2_ref = torch.empty((N * h * w, 40), dtype=torch.bfloat16, device=DEV)
    ref = ops.bneck(x, *blk, G=G, out2=out2_ref)
    gate, ysum, xsum = ops.gate_shift(xs, B, T, F, Fp, bn_s, bn_b, wq, b3d, *cw, wqf=wqf, gates_only=True)
    out2 = torch.empty_like(out2_ref)
    out = ops.bneck_gs(x, xs, gate, ysum, xsum, *cw, T, F, Fp, *blk, out2=out2)
    torch.cuda.synchronize()
    assert torch.equal(out, ref) and torch.equal(out2, out2_ref)


@pytest.mark.parametrize("h,w,C,R,B,T", [(7, 7, 368, 92, 3, 7), (14, 14, 152, 38, 2, 5), (7, 7, 368, 92, 8, 100), (13, 14, 152, 38, 3, 3),
                                         (13, 7, 368, 92, 2, 3), (14, 14, 152, 38, 8, 100)])
def test_tap_maps_from_the_bottlenecks_tail_equal_the_gate_launch(h, w, C, R, B, T):
    """tdeed_bneck_gs_fwd's Q: the tap maps of the NEXT site, made from the output rows in LDS (1x1 contraction to per-tap sums +
    nine adds), against the first launch of tdeed_gsf_gate_fwd run on the block's output (implicit GEMM over (tap, channel)):
    the same bf16 products in fp32, another summation order -- equal to 2e-6 of the largest map value; the site's second launch
    alone (tdeed_gsf_gate_sums_fwd) on the launch's own Q equals the two-launch call bit for bit; the block's own outputs are
    unchanged by the tail."""
    from tdeed_amd import ops
    from tdeed_amd.engine import pack_gsf_q_frags, pack_gsf_p_frags
    from tdeed_amd.regnet_spec import gsf_fold_dim
    g = torch.Generator().manual_seed(h * 77 + C + B + T)
    F = gsf_fold_dim(C)
    Fp = (F + 7) // 8 * 8
    N = B * T
    assert ops.bneck_fits(h, w, C, R) and ops.bneck_qtail_fits(h, w, C, F)
    x = torch.relu(torch.randn(N, h, w, C, generator=g)).to(torch.bfloat16).to(DEV)
    f32 = lambda *s: (torch.randn(s, generator=g) * 0.3).to(DEV)      # noqa: E731

    def site():
        w3d = torch.randn(2, F // 2, 3, 3, 3, generator=g) * 0.1
        return dict(bn_s=f32(F).abs() + 0.5, bn_b=f32(F), b3d=f32(2), cw=[f32(18), f32(1), f32(18), f32(1)],
                    wq=w3d.reshape(F, 27).t().contiguous().to(DEV), wqf=pack_gsf_q_frags(w3d.numpy(), DEV),
                    wpf=pack_gsf_p_frags(w3d.numpy(), DEV))
    s0, s1 = site(), site()
    blk = _block(g, h, w, C, 8, R)
    gate, ysum, xsum = ops.gate_shift(x, B, T, F, Fp, s0["bn_s"], s0["bn_b"], s0["wq"], s0["b3d"], *s0["cw"], wqf=s0["wqf"],
                                      gates_only=True)
    o2a, o2b = (torch.empty((N * h * w, Fp), dtype=torch.bfloat16, device=DEV) for _ in range(2))
    ref = ops.bneck_gs(x, x, gate, ysum, xsum, *s0["cw"], T, F, Fp, *blk, out2=o2a)
    Q = torch.full((N, h, w, 6), float("nan"), device=DEV)
    out = ops.bneck_gs(x, x, gate, ysum, xsum, *s0["cw"], T, F, Fp, *blk, out2=o2b,
                       qtail=(s1["wpf"], ops.gsq_bn_table(s1["bn_s"], s1["bn_b"]), F, Q))
    assert torch.equal(out, ref) and torch.equal(o2a, o2b)
    bufs = dict(q=torch.empty((N, h, w, 6), device=DEV))
    g1, y1, x1 = ops.gate_shift(o2a.view(N, h, w, Fp), B, T, F, Fp, s1["bn_s"], s1["bn_b"], s1["wq"], s1["b3d"], *s1["cw"],
                                wqf=s1["wqf"], bufs=bufs, gates_only=True)
    assert bool(torch.isfinite(Q).all())
    assert float((Q - bufs["q"]).abs().max()) <= 2e-6 * float(bufs["q"].abs().max()), float((Q - bufs["q"]).abs().max())
    g2, y2, x2 = ops.gate_shift(o2a.view(N, h, w, Fp), B, T, F, Fp, s1["bn_s"], s1["bn_b"], s1["wq"], s1["b3d"], *s1["cw"],
                                wqf=s1["wqf"], bufs=dict(q=bufs["q"].clone()), gates_only=True, q_given=True)
    assert torch.equal(g1, g2) and torch.equal(y1, y2) and torch.equal(x1, x2)


def test_bottleneck_blend_rejects_what_it_cannot_hold():
    from tdeed_amd import ops
    x = torch.zeros((4, 7, 7, 368), dtype=torch.bfloat16, device=DEV)
    g = torch.Generator().manual_seed(1)
    blk = _block(g, 7, 7, 368, 8, 92)
    z = lambda *s: torch.zeros(s, device=DEV)      # noqa: E731
    cw = [z(18), z(1), z(18), z(1)]
    with pytest.raises(RuntimeError, match="whole clips"):
        ops.bneck_gs(x, x, z(4, 7, 7, 2), z(4, 92), z(4, 92), *cw, 3, 92, 96, *blk)
    assert not ops.bneck_qtail_fits(5, 7, 152, 40)          # two 35-pixel frames of 152 channels: region A is too small for the tail
    with pytest.raises(RuntimeError, match="bad fold"):
        ops.bneck_gs(x, x, z(4, 7, 7, 2), z(4, 92), z(4, 92), *cw, 2, 90, 96, *blk)


def test_whole_forward_with_the_site_fusions_equals_the_forward_without_them(monkeypatch):
    """RegNetY-200MF + GSF at 224 x 224 (7 x 7 and 14 x 14 one-launch bottlenecks): the plan with the blend inside the
    bottleneck's frame load equals the plan with the blend launch bit for bit; with the tap maps in the tail as well (another
    fp32 summation order for seven sites' gates) the logits stay within the bf16 noise of a single rounding flip."""
    from tdeed_amd import synth, state_layout, engine
    from tdeed_amd.engine import ForwardEngine
    cfg = dict(feature_arch="rny002_gsf", clip_len=6, crop_dim=None, n_layers=2, sgp_ks=5, sgp_r=2, num_classes=3,
               radi_displacement=2)
    sd = synth.make_state(state_layout.model_state_shapes(cfg), 3)
    clip = torch.from_numpy(synth.uint8_clip(11, (3, 6, 3, 224, 224))).to(DEV)
    outs = {}
    for name, blend, qtail in (("chain", False, False), ("blend", True, False), ("both", True, True)):
        monkeypatch.setattr(engine, "BNECK_BLEND", blend)
        monkeypatch.setattr(engine, "BNECK_QTAIL", qtail)
        with torch.cuda.stream(torch.cuda.Stream()):
            eng = ForwardEngine(cfg, sd, torch.bfloat16, DEV, use_graph=False)
            plan = eng.plan(3, 224, 224)
            kinds = [s.name for s in plan.steps if s.kernel == "bneck"]
            assert len(kinds) == 9
            out, _ = eng.forward(clip)
        torch.cuda.synchronize()
        outs[name] = out.float().cpu()
        del eng, plan
    assert torch.equal(outs["blend"], outs["chain"])
    scale = float(outs["chain"].abs().max())
    assert float((outs["both"] - outs["chain"]).abs().max()) < 2e-2 * scale
