"""Parked with experiments/r4_parked/bneck_with_gs_apply.hip (the test of the in-kernel gate-shift apply step)."""
@pytest.mark.parametrize("h,w,C,F,B,T,compact", [(7, 7, 368, 92, 2, 7, True), (14, 14, 152, 40, 1, 5, True),
                                                  (7, 7, 368, 92, 1, 6, False), (5, 5, 152, 36, 3, 4, True)])
def test_gate_shift_apply_inside_the_one_launch_bottleneck(h, w, C, F, B, T, compact):
    """tdeed_bneck_fwd with the gate-shift-fuse APPLY step in its load phase (impl/gsf.py:66-93: fusion weights from the
    per-frame spatial sums, gated temporal shifts, blend) against the launch it replaces (tdeed_gsf_apply_fused_fwd producing
    G, spliced by the same bottleneck launch): bitwise -- clips of odd and even length (a workgroup's two frames may belong to
    different clips), slice read from the compact copy or from the block input itself."""
    from tdeed_amd import ops
    from tdeed_amd.engine import pack_mfma_frags, pack_gconv_frags, pack_se_mfma
    g = torch.Generator().manual_seed(h * 100 + C + F + T)
    N, hw = B * T, h * w
    M = N * hw
    Fp = (F + 7) // 8 * 8
    R, gw = (92 if C == 368 else 38), 8
    assert ops.bneck_fits(h, w, C, R)
    x = torch.relu(torch.randn(N, h, w, C, generator=g)).to(torch.bfloat16).to(DEV)
    xs = x[..., :Fp].contiguous() if compact else x
    vec = lambda n, s=0.1, o=0.0: (torch.randn(n, generator=g) * s + o).to(DEV)          # noqa: E731
    # gate-shift module parameters (BatchNorm3d folded, conv3D 2 x F/2 x 3x3x3, fusion convs 2 -> 1 x 3x3)
    bn_s, bn_h = vec(F, 0.1, 1.0), vec(F)
    w3d = torch.randn(2, F // 2, 3, 3, 3, generator=g) * 0.2
    wq = w3d.reshape(F, 27).t().contiguous().to(DEV)
    b3d = vec(2)
    cw1, cw2, cb1, cb2 = vec(18, 0.5), vec(18, 0.5), vec(1), vec(1)
    bufs = dict(gate=torch.empty((N, h, w, 2), device=DEV), q=torch.empty((N, h, w, 6), device=DEV),
                ysum=torch.empty((N, F), device=DEV), xsum=torch.empty((N, F), device=DEV),
                out=torch.empty((M, Fp), dtype=torch.bfloat16, device=DEV))
    G = ops.gate_shift(xs, B, T, F, Fp, bn_s, bn_h, wq, b3d, cw1, cb1, cw2, cb2, bufs=bufs)
    W1, W3 = torch.randn(C, C, generator=g) / C ** 0.5, torch.randn(C, C, generator=g) / C ** 0.5
    W2 = torch.randn(C, gw, 3, 3, generator=g) / (gw * 9) ** 0.5
    fc1, fc2 = torch.randn(R, C, generator=g) / C ** 0.5, torch.randn(C, R, generator=g) / R ** 0.5
    s1, h1, s2, h2, s3, h3, b1, b2 = vec(C, .1, 1.), vec(C), vec(C, .1, 1.), vec(C), vec(C, .1, .5), vec(C), vec(R), vec(C)
    w1f, w3f = pack_mfma_frags(W1.numpy(), DEV), pack_mfma_frags(W3.numpy(), DEV)
    w2f = pack_gconv_frags(W2.numpy(), gw, DEV)
    se = pack_se_mfma(fc1.numpy(), fc2.numpy(), DEV)
    args = (w1f, s1, h1, w2f, s2, h2, se["w1f"], b1, se["w2f"], b2, R, w3f, s3, h3)
    ref = ops.bneck(x, *args, G=G)
    got = ops.bneck(x, *args, gs=dict(x=xs, Fp=Fp, F=F, T=T, gate=bufs["gate"], ysum=bufs["ysum"], xsum=bufs["xsum"],
                                      cw1=cw1, cb1=cb1, cw2=cw2, cb2=cb2))
    torch.cuda.synchronize()
    assert torch.isfinite(got.float()).all()
    assert torch.equal(got, ref), float((got.float() - ref.float()).abs().max())


