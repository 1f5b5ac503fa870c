"""Round-5 parity tests on the MI355X (through the C ABI).  -m gpu only."""
import numpy as np
import pytest
import torch

from helpers import load_golden, model_state, t, max_abs
from tdeed_amd import synth, state_layout

pytestmark = pytest.mark.gpu
DEV = "cuda"


def test_backward_after_a_repack_is_refused_where_it_would_recompute_from_the_new_weights():
    """the narrow one-launch backward (trunk_bwd3.hip, recompute=True) re-derives z from the packed weights;
    a repack between forward and backward would silently change what it differentiates.  The engine now refuses."""
    from tdeed_amd.trainer import TrainEngine
    cfg = dict(feature_arch="rny008_gsf", clip_len=4, crop_dim=None, n_layers=2, sgp_ks=5, sgp_r=2, num_classes=3,
               radi_displacement=0)
    B, T, H, W = 2, cfg["clip_len"], 64, 64
    sd0 = {k: t(v) for k, v in synth.make_state(state_layout.model_state_shapes(cfg), 9).items()}
    frames = t(synth.uint8_clip(561, (B, T, 3, H, W))).to(DEV)
    lab = t(synth.labels(562, B, T, cfg["num_classes"], 1, fg_frac=0.4)[0]).long().to(DEV)
    eng = TrainEngine(cfg, {k: v.clone() for k, v in sd0.items()}, act_dtype=torch.bfloat16, lr=1e-3)
    head, ctx = eng.forward_train(frames)
    _, dhead = eng.temporal.loss_fwd_bwd(head, B, T, lab.reshape(-1).contiguous())
    eng.repack()
    with pytest.raises(RuntimeError, match="packed weights changed"):
        eng.backward_train(ctx, dhead)
    # and the ordinary order still works
    head, ctx = eng.forward_train(frames)
    _, dhead = eng.temporal.loss_fwd_bwd(head, B, T, lab.reshape(-1).contiguous())
    eng.backward_train(ctx, dhead)
    torch.cuda.synchronize()


# ------------------------------------------------------------------------------------------------ sgp_gemm.hip
def _rnd(seed, name, shape, scale=1.0):
    return t((synth.normalish(seed, name, int(np.prod(shape))) * scale).reshape(shape).astype(np.float32))


FORMS = [(4, 2), (4, 1), (2, 2), (2, 1), (1, 2), (1, 1), None]


@pytest.mark.parametrize("adt", [torch.bfloat16, torch.float32])
@pytest.mark.parametrize("B,T,C", [(8, 100, 368), (3, 50, 368), (2, 25, 48), (2, 13, 768), (1, 250, 96)])
def test_sgp_gemm_groupnorm_fc1_gelu(B, T, C, adt):
    """MODE 0 of sgp_gemm.hip against torch: H = GELU(GroupNorm16(y) @ W1^T + b1) (modules.py:134-136, 186), every tile form,
    bf16 and fp32 rows, the GroupNorm statistics from per-channel sums handed in as 1 or 3 partials."""
    from tdeed_amd import ops
    from tdeed_amd.engine import pack_mfma_frags
    N = 4 * C
    y = _rnd(1, "y", (B, T, C), 1.5).to(DEV).to(adt)
    W = _rnd(2, "w", (N, C), 0.08).to(DEV)
    b1 = _rnd(3, "b", (N,), 0.1).to(DEV)
    gw, gb = (1.0 + _rnd(4, "gw", (C,), 0.2)).to(DEV), _rnd(5, "gb", (C,), 0.2).to(DEV)
    Wp = pack_mfma_frags(W.cpu().numpy(), DEV, ks_mult=12)
    yf = y.float()
    chs = torch.stack([yf.sum(1), (yf * yf).sum(1)], -1).contiguous()                 # (B, C, 2)
    parts3 = torch.stack([chs * 0.5, chs * 0.25, chs * 0.25]).contiguous()
    gn = torch.nn.functional.group_norm(yf.transpose(1, 2), 16, gw, gb, 1e-5).transpose(1, 2)
    ref = torch.nn.functional.gelu(gn.to(torch.bfloat16).float() @ W.to(torch.bfloat16).float().t() + b1)
    forms = FORMS if adt == torch.bfloat16 else [f for f in FORMS if f != (4, 2)]
    for form in forms:
        for cs in (chs, parts3):
            H = ops.sgp_gemm_gn_gelu(y, cs, gw, gb, Wp, b1, N, form=form)
            torch.cuda.synchronize()
            err = float((H.float() - ref).abs().max())
            assert err < 2e-2 * max(1.0, float(ref.abs().max())), (form, err)


@pytest.mark.parametrize("odt", [torch.bfloat16, torch.float32])
@pytest.mark.parametrize("B,T,C,pool", [(8, 100, 368, True), (3, 50, 368, True), (2, 25, 48, False), (2, 13, 768, False),
                                        (1, 250, 96, True)])
def test_sgp_gemm_fc2_residual_rowsums_pool(B, T, C, pool, odt):
    """MODE 1: out = y + H @ W2^T + b2 (modules.py:137, 186), the per-row sums the next LayerNorm is derived from (exactly
    the sums of the stored values) and the fused AdaptiveMaxPool1d(T/2) (modules.py:64, 77) with its row sums."""
    from tdeed_amd import ops
    from tdeed_amd.engine import pack_mfma_frags
    K = 4 * C
    H = _rnd(11, "h", (B, T, K), 0.7).to(DEV).to(torch.bfloat16)
    W = _rnd(12, "w", (C, K), 0.05).to(DEV)
    b2 = _rnd(13, "b", (C,), 0.1).to(DEV)
    y = _rnd(14, "y", (B, T, C), 1.0).to(DEV).to(odt)
    Wp = pack_mfma_frags(W.cpu().numpy(), DEV, ks_mult=12)
    ref = (y.float() + H.float() @ W.to(torch.bfloat16).float().t() + b2)
    for form in FORMS:
        f = form or ops.sgp_gemm_form(1, B, T, C, K)
        NJ, nct = ops.sgp_gemm_tiles(T, C, f)
        rsp = torch.full((nct, B * T, 2), float("nan"), device=DEV)
        pooled = torch.full((B, T // 2, C), float("nan"), device=DEV, dtype=odt) if pool else None
        rpp = torch.full((nct, B * (T // 2), 2), float("nan"), device=DEV) if pool else None
        out = ops.sgp_gemm_residual(H, Wp, b2, y, rowstat_part=rsp, pooled=pooled, rowstat_pool_part=rpp, form=f)
        torch.cuda.synchronize()
        tol = (2e-2 if odt == torch.bfloat16 else 2e-3) * max(1.0, float(ref.abs().max()))
        assert float((out.float() - ref).abs().max()) < tol, form
        of = out.float().reshape(B * T, C)
        s = rsp.sum(0)
        assert float((s[:, 0] - of.sum(1)).abs().max()) < 1e-3 * C ** 0.5 * max(1.0, float(of.abs().max())), form
        assert float((s[:, 1] - (of * of).sum(1)).abs().max()) < 1e-3 * float((of * of).sum(1).max()), form
        if pool:
            pr = torch.nn.functional.adaptive_max_pool1d(out.float().transpose(1, 2), T // 2).transpose(1, 2)
            assert torch.equal(pooled.float(), pr), form
            pf = pooled.float().reshape(B * (T // 2), C)
            sp = rpp.sum(0)
            assert float((sp[:, 0] - pf.sum(1)).abs().max()) < 1e-3 * C ** 0.5 * max(1.0, float(pf.abs().max())), form
            assert float((sp[:, 1] - (pf * pf).sum(1)).abs().max()) < 1e-3 * float((pf * pf).sum(1).max()), form


@pytest.mark.parametrize("odt", [torch.bfloat16, torch.float32])
@pytest.mark.parametrize("B,T,C", [(8, 100, 368), (3, 50, 368), (2, 25, 48), (2, 13, 768)])
def test_sgp_gemm_concat_gelu_channel_sums(B, T, C, odt):
    """MODE 2: mo = GELU(cat @ Wc^T + bc) (modules.py:307-308) and the per-channel sums of the stored rows per row tile
    (what the mixer's GroupNorm statistics are derived from)."""
    from tdeed_amd import ops
    from tdeed_amd.engine import pack_mfma_frags
    K = 6 * C
    A = _rnd(21, "a", (B, T, K), 0.8).to(DEV).to(torch.bfloat16)
    W = _rnd(22, "w", (C, K), 0.04).to(DEV)
    bc = _rnd(23, "b", (C,), 0.1).to(DEV)
    Wp = pack_mfma_frags(W.cpu().numpy(), DEV, ks_mult=12)
    ref = torch.nn.functional.gelu(A.float() @ W.to(torch.bfloat16).float().t() + bc)
    for form in FORMS:
        f = form or ops.sgp_gemm_form(2, B, T, C, K)
        NJ, nct = ops.sgp_gemm_tiles(T, C, f)
        out = torch.full((B, T, C), float("nan"), device=DEV, dtype=odt)
        chs = torch.full((NJ, B, C, 2), float("nan"), device=DEV)
        o16 = torch.full((B, T, C), float("nan"), device=DEV, dtype=torch.bfloat16)
        ops.sgp_gemm_gelu_chsum(A, Wp, bc, C, out, chs, form=f, out16=o16)
        torch.cuda.synchronize()
        assert torch.equal(o16, out.to(torch.bfloat16)), form
        tol = (2e-2 if odt == torch.bfloat16 else 2e-3) * max(1.0, float(ref.abs().max()))
        assert float((out.float() - ref).abs().max()) < tol, form
        of = out.float()
        s = chs.sum(0)
        assert float((s[..., 0] - of.sum(1)).abs().max()) < 1e-3 * T ** 0.5 * max(1.0, float(of.abs().max())), form
        assert float((s[..., 1] - (of * of).sum(1)).abs().max()) < 1e-3 * max(1.0, float((of * of).sum(1).max())), form


# ------------------------------------------------------------------------------------------------ the stage on sgp_gemm
def _run(steps):
    for s_ in steps:
        s_.fn()
    torch.cuda.synchronize()


@pytest.mark.parametrize("stream", [torch.bfloat16, torch.float32])
@pytest.mark.parametrize("name", ["sgp_block_c32_t25", "sgp_block_c368_t100", "sgp_block_c48_t13"])
def test_sgp_block_on_sgp_gemm_matches_the_reference(name, stream):
    """SGPBlock.forward (modules.py:159-188) with bf16 contractions on sgp_gemm.hip, residual stream bf16 or fp32 (the
    throughput mode), against the reference's output."""
    from tdeed_amd.engine import SgpBuilder, pack_sgp_block, _Pool
    from helpers import module_state, act
    meta, g = load_golden(name)
    B, C, T = meta["B"], meta["C"], meta["T"]
    sd = module_state("sgp_block", "blk", meta["seed"], **meta)
    x = t(act(meta["seed"], name + ":x", (B, C, T)).transpose(0, 2, 1).copy()).to(stream).to(DEV)
    steps, keep = [], {}
    sb = SgpBuilder(_Pool(DEV), steps, keep, set(), B, torch.bfloat16)
    assert sb.gemm
    out = sb.block(x, T, pack_sgp_block(sd, "blk", C, torch.bfloat16, DEV), "blk")
    assert any(s_.kernel == "sgp_gemm" for s_ in steps) and out.dtype == stream
    _run(steps)
    ref = t(g["y"]).permute(0, 2, 1)
    tol = 4e-2 if stream == torch.bfloat16 else 2e-2
    assert max_abs(out.float().cpu(), ref) < tol * max(1.0, float(ref.abs().max()))


@pytest.mark.parametrize("stream", [torch.bfloat16, torch.float32])
@pytest.mark.parametrize("name", ["sgp_mixer_c32_t25", "sgp_mixer_c368_t100"])
def test_sgp_mixer_on_sgp_gemm_matches_the_reference(name, stream):
    from tdeed_amd.engine import SgpBuilder, pack_sgp_mixer, _Pool
    from helpers import module_state, act
    meta, g = load_golden(name)
    B, C, Th, Tl = meta["B"], meta["C"], meta["T_hi"], meta["T_lo"]
    sd = module_state("sgp_mixer", "mix", meta["seed"], **meta)
    z = t(act(meta["seed"], name + ":z", (B, C, Th)).transpose(0, 2, 1).copy()).to(stream).to(DEV)
    x = t(act(meta["seed"], name + ":x", (B, C, Tl)).transpose(0, 2, 1).copy()).to(stream).to(DEV)
    steps, keep = [], {}
    sb = SgpBuilder(_Pool(DEV), steps, keep, set(), B, torch.bfloat16)
    out = sb.mixer(x, Tl, z, Th, pack_sgp_mixer(sd, "mix", C, torch.bfloat16, DEV), "mix")
    assert sum(s_.kernel == "sgp_gemm" for s_ in steps) == 3 and out.dtype == stream
    _run(steps)
    ref = t(g["y"]).permute(0, 2, 1)
    tol = 4e-2 if stream == torch.bfloat16 else 2e-2
    assert max_abs(out.float().cpu(), ref) < tol * max(1.0, float(ref.abs().max()))


@pytest.mark.parametrize("stream", [torch.bfloat16, torch.float32])
@pytest.mark.parametrize("name", ["pyramid_c32_l25_n2", "pyramid_c64_l100_n3", "pyramid_c48_l250_n2"])
def test_pyramid_on_sgp_gemm_matches_the_reference(name, stream):
    """EDSGPMIXERLayers.forward (modules.py:69-87): even levels pool inside the fc2 launch, odd ones (25 -> 13, 125 -> 63)
    through the max-pool launch + in-kernel LayerNorm statistics."""
    from tdeed_amd.engine import SgpBuilder, pack_sgp_block, pack_sgp_mixer, _Pool
    from helpers import module_state, act
    meta, g = load_golden(name)
    B, C, L, n = meta["B"], meta["C"], meta["L"], meta["n"]
    sd = module_state("pyramid", "_temp_fine", meta["seed"], **meta)
    x = t(act(meta["seed"], name + ":x", (B, L, C))).to(stream).to(DEV)
    sgp = [pack_sgp_block(sd, f"_temp_fine._sgp.{i}", C, torch.bfloat16, DEV) for i in range(2 * n + 1)]
    mix = [pack_sgp_mixer(sd, f"_temp_fine._sgpMixer.{i}", C, torch.bfloat16, DEV) for i in range(n)]
    steps, keep = [], {}
    sb = SgpBuilder(_Pool(DEV), steps, keep, set(), B, torch.bfloat16)
    out = sb.pyramid(x, L, n, sgp, mix)
    kinds = [s_.kernel for s_ in steps]
    assert "sgp_mlp2" not in kinds and "gemm_splitk" not in kinds and kinds.count("sgp_gemm") == 2 * (2 * n + 1) + 3 * n
    _run(steps)
    ref = t(g["y"])
    tol = 6e-2 if stream == torch.bfloat16 else 3e-2
    assert max_abs(out.float().cpu(), ref) < tol * max(1.0, float(ref.abs().max()))


@pytest.mark.parametrize("T", [4, 2, 6])
def test_tiny_clips_through_the_stage(T):
    """clip_len 4 / 2 / 6: levels of 2 and 1 rows (tiles far below 16 rows, T + 2 halo < 16 rows of LDS tile): the bf16 stage
    with the fp32 stream tracks the fp32 launch-per-op chain (regression: the front kernels' second reduction scratch
    overlapped their result tile for T + 2 halo < 16)."""
    from tdeed_amd.engine import SgpBuilder, pack_sgp_block, pack_sgp_mixer, _Pool
    from helpers import module_state, act
    B, C, n = 2, 368, 2 if T >= 4 else 1
    sd = module_state("pyramid", "_temp_fine", 5, C=C, ks=5, r=2, n=n)
    x = t(act(9, "x", (B, T, C)))
    outs = []
    for wdt, sdt in ((torch.float32, torch.float32), (torch.bfloat16, torch.float32), (torch.bfloat16, torch.bfloat16)):
        sgp = [pack_sgp_block(sd, f"_temp_fine._sgp.{i}", C, wdt, DEV) for i in range(2 * n + 1)]
        mix = [pack_sgp_mixer(sd, f"_temp_fine._sgpMixer.{i}", C, wdt, DEV) for i in range(n)]
        steps, keep = [], {}
        sb = SgpBuilder(_Pool(DEV), steps, keep, set(), B, wdt)
        out = sb.pyramid(x.to(sdt).to(DEV), T, n, sgp, mix)
        _run(steps)
        outs.append(out.float().cpu())
    scale = max(1.0, float(outs[0].abs().max()))
    assert max_abs(outs[1], outs[0]) < 3e-2 * scale and max_abs(outs[2], outs[0]) < 6e-2 * scale


# ------------------------------------------------------------------------------------------------ the timed sizes
@pytest.mark.parametrize("h,w,C,gw,R,Fp", [(7, 7, 368, 8, 92, 96), (14, 14, 152, 8, 38, 40)])
def test_one_launch_bottleneck_at_the_timed_size_is_within_one_ulp_of_the_chain(h, w, C, gw, R, Fp):
    """tdeed_bneck_fwd against gemm -> gconv3x3 -> se_gate_mfma -> gemm at N = 800 frames (B = 8 clips
    of 100: what bench.py times).  The squeeze sums of a workgroup's frames are folded in another order than the chain's,
    so the SE gate may differ in its last fp32 bit and a handful of outputs by ONE bf16 ulp: at most 1e-5 of the elements,
    none by more than one ulp of its own magnitude."""
    from tdeed_amd import ops
    from tdeed_amd.engine import pack_mfma_frags, pack_gconv_frags, pack_se_mfma
    N = 800
    g = torch.Generator().manual_seed(h * 100 + C)
    hw = h * w
    M = N * hw
    x = torch.relu(torch.randn(N, h, w, C, generator=g)).to(torch.bfloat16)
    G = torch.randn(M, Fp, generator=g).to(torch.bfloat16)
    W1 = torch.randn(C, C, generator=g) / C ** 0.5
    W3 = torch.randn(C, C, generator=g) / C ** 0.5
    W2 = torch.randn(C, gw, 3, 3, generator=g) / (gw * 9) ** 0.5
    fc1 = torch.randn(R, C, generator=g) / C ** 0.5
    fc2 = torch.randn(C, R, generator=g) / R ** 0.5
    vec = lambda n, s_=0.1, o=0.0: (torch.randn(n, generator=g) * s_ + o).to(DEV)          # noqa: E731
    s1, h1, s2, h2, s3, h3 = vec(C, 0.1, 1.0), vec(C), vec(C, 0.1, 1.0), vec(C), vec(C, 0.1, 0.5), vec(C)
    b1, b2 = vec(R), vec(C)
    W1d, W3d = W1.to(torch.bfloat16).to(DEV), W3.to(torch.bfloat16).to(DEV)
    w2f = pack_gconv_frags(W2.numpy(), gw, DEV)
    se = pack_se_mfma(fc1.numpy(), fc2.numpy(), DEV)
    xd, Gd = x.to(DEV), G.to(DEV)
    y1 = ops.gemm(xd.view(M, C), W1d, s1, h1, ops.ACT_RELU, A0=Gd, k0=Fp)
    y2, pooled = ops.gconv3x3(y1.view(N, h, w, C), None, s2, h2, gw, 1, wfrag=w2f)
    gate = ops.se_gate_mfma(pooled, 1.0 / hw, se["w1f"], b1, se["w2f"], b2, R)
    ref = ops.gemm(y2.view(M, C), W3d, s3, h3, ops.ACT_RELU, residual=xd.view(M, C), a_scale=gate, a_scale_rows=hw)
    out = ops.bneck(xd, pack_mfma_frags(W1.numpy(), DEV), s1, h1, pack_gconv_frags(W2.numpy(), gw, DEV, tap_major=(gw == 8)), s2, h2,
                    se["w1f"], b1, se["w2f"], b2, R,
                    pack_mfma_frags(W3.numpy(), DEV), s3, h3, G=Gd, w2_tap_major=(gw == 8)).view(M, C)
    torch.cuda.synchronize()
    a, b = out.float(), ref.float()
    diff = (a - b).abs()
    nd = int((diff > 0).sum())
    assert nd <= 1e-5 * a.numel(), nd
    ulp = torch.maximum(a.abs(), b.abs()) * 2.0 ** -7          # one bf16 ulp is at most 2^-7 of the magnitude
    assert bool((diff <= ulp + 1e-30).all()), float((diff - ulp).max())


def test_cfg5_geometry_at_its_real_size_tracks_the_fp32_engine():
    """BASELINE configs[4] per-GPU share as bench.py runs it (RegNetY-800MF, T = 250, 224 x 224, B = 4): the bf16 forward of
    the whole batch against the fp32 engine on the same clips (the fp32 engine is the one held to the reference's goldens,
    incl. snb_t250 at 160 x 160), finite everywhere, the arg-max of nearly every frame equal."""
    from tdeed_amd.engine import ForwardEngine
    cfg = dict(feature_arch="rny008_gsf", clip_len=250, crop_dim=None, n_layers=2, sgp_ks=9, sgp_r=4, num_classes=12,
               radi_displacement=4)
    B, T, H, W = 4, 250, 224, 224
    sd = model_state(cfg, 0)
    clip = t(synth.uint8_clip(4242, (B, T, 3, H, W))).to(DEV)
    outs = {}
    for dt in (torch.float32, torch.bfloat16):
        with torch.cuda.stream(torch.cuda.Stream()):
            eng = ForwardEngine(cfg, sd, dt, DEV, use_graph=(dt == torch.bfloat16), n_split=1)
            plan = eng.plan(B, H, W)
            eng.set_frames(plan, clip)
            eng.run_plan(plan)
        torch.cuda.synchronize()
        outs[dt] = plan.head_out.float().cpu().view(B, T, -1)
        del eng, plan
        torch.cuda.empty_cache()
    K1 = cfg["num_classes"] + 1
    a, b = outs[torch.bfloat16], outs[torch.float32]
    assert torch.isfinite(a).all() and torch.isfinite(b).all()
    scale = float(b[..., :K1].abs().max())
    assert float((a[..., :K1] - b[..., :K1]).abs().max()) < 0.08 * max(1.0, scale)
    assert float((a[..., :K1].argmax(-1) == b[..., :K1].argmax(-1)).float().mean()) > 0.9


@pytest.mark.parametrize("T,n", [(50, 2), (26, 2)])
def test_wide_stage_with_bf16_operand_copies_and_odd_levels(T, n):
    """C = 768 (RegNetY-800MF): the fp32-stream stage hands fc1 a bf16 copy of its rows (sgp_front / the concat launch write
    it beside the fp32 tensor) and pools odd lengths (25 -> 13, 13 -> 7) through tdeed_maxpool_rowstat_fwd; against the fp32
    launch-per-op chain on the same weights."""
    from tdeed_amd.engine import SgpBuilder, pack_sgp_block, pack_sgp_mixer, _Pool
    from helpers import module_state, act
    B, C = 2, 768
    sd = module_state("pyramid", "_temp_fine", 5, C=C, ks=7, r=4, n=n)
    x = t(act(9, "x", (B, T, C)))
    outs, kinds = [], None
    for wdt in (torch.float32, torch.bfloat16):
        sgp = [pack_sgp_block(sd, f"_temp_fine._sgp.{i}", C, wdt, DEV) for i in range(2 * n + 1)]
        mix = [pack_sgp_mixer(sd, f"_temp_fine._sgpMixer.{i}", C, wdt, DEV) for i in range(n)]
        steps, keep = [], {}
        sb = SgpBuilder(_Pool(DEV), steps, keep, set(), B, wdt)
        out = sb.pyramid(x.to(DEV), T, n, sgp, mix)
        _run(steps)
        outs.append(out.float().cpu())
        kinds = [s_.kernel for s_ in steps]
    assert "maxpool" in kinds and kinds.count("sgp_gemm") == 2 * (2 * n + 1) + 3 * n
    assert max_abs(outs[1], outs[0]) < 3e-2 * max(1.0, float(outs[0].abs().max()))


# ------------------------------------------------------------------------------------ gate-shift at the timed frame counts
@pytest.mark.parametrize("h,C,F,B,T", [(7, 96, 92, 9, 89), (14, 40, 40, 5, 161), (7, 368, 92, 8, 100)])
def test_gate_shift_two_frames_per_workgroup_equals_clip_by_clip(h, C, F, B, T):
    """Above 768 frames (3 resident workgroups per CU) tdeed_gsf_gate_fwd gives a workgroup of the tap-map launch TWO frames
    (the cfg2 batch is 800); the module is independent per clip, so the batch in one call must equal the clips one by one
    (each below the threshold: one frame per workgroup) bit for bit -- odd frame counts (a last workgroup with one frame),
    pairs that span two clips, both blends (module order and source order)."""
    from tdeed_amd import ops
    from tdeed_amd.engine import pack_gsf_q_frags
    g = torch.Generator().manual_seed(h * 1000 + F + B)
    N, Fp = B * T, (F + 7) // 8 * 8
    x = (torch.randn((N, h, h, C), generator=g) * 0.5).to(torch.bfloat16).to(DEV)
    w3d = torch.randn(2, F // 2, 3, 3, 3, generator=g) * 0.1
    f32 = lambda *s: (torch.randn(s, generator=g) * 0.3).to(DEV)      # noqa: E731
    bn_s, bn_b, b3d = f32(F).abs() + 0.5, f32(F), f32(2)
    cw = [f32(18), f32(1), f32(18), f32(1)]
    wq, wqf = w3d.reshape(F, 27).t().contiguous().to(DEV), pack_gsf_q_frags(w3d.numpy(), DEV)
    assert N > 768
    for src in (False, True):
        whole = ops.gate_shift(x, B, T, F, Fp, bn_s, bn_b, wq, b3d, *cw, wqf=wqf, src_order=src).view(B, T * h * h, Fp)
        for b in range(B):
            one = ops.gate_shift(x[b * T:(b + 1) * T], 1, T, F, Fp, bn_s, bn_b, wq, b3d, *cw, wqf=wqf, src_order=src)
            assert torch.equal(one, whole[b]), (src, b)


@pytest.mark.parametrize("h,w,C,F,B,T", [(5, 9, 24, 20, 2, 3), (3, 3, 16, 8, 1, 1), (9, 5, 48, 44, 3, 2), (14, 14, 88, 80, 2, 7),
                                         (7, 7, 200, 196, 2, 5), (6, 6, 40, 36, 4, 2), (11, 13, 64, 60, 2, 2),
                                         (7, 7, 104, 100, 1, 9), (4, 4, 32, 28, 5, 1)])
def test_gate_shift_odd_geometries(h, w, C, F, B, T):
    """The bf16 gate-shift launches on geometries no model config has: non-square maps, folds that are not multiples of 8
    (a chunk and a piece straddling F/2), clips of one frame, folds on both sides of the register-weight limit (F = 100, 196:
    tap weights read from LDS).  Source-order blend == module-order blend under the interleave, bit for bit; both within the
    bf16 tolerance of the fp32 kernels on the same (bf16-rounded) input."""
    from tdeed_amd import ops
    from tdeed_amd.engine import pack_gsf_q_frags
    g = torch.Generator().manual_seed(h * 131 + w * 7 + F)
    Fp = (F + 7) // 8 * 8
    x = (torch.randn((B * T, h, w, C), generator=g) * 0.5).to(torch.bfloat16).to(DEV)
    w3d = torch.randn(2, F // 2, 3, 3, 3, generator=g) * 0.1
    f32 = lambda *s: (torch.randn(s, generator=g) * 0.3).to(DEV)      # noqa: E731
    bn_s, bn_b, b3d = f32(F).abs() + 0.5, f32(F), f32(2)
    cw = [f32(18), f32(1), f32(18), f32(1)]
    wq, wqf = w3d.reshape(F, 27).t().contiguous().to(DEV), pack_gsf_q_frags(w3d.numpy(), DEV)
    mod = ops.gate_shift(x, B, T, F, Fp, bn_s, bn_b, wq, b3d, *cw, wqf=wqf)
    src = ops.gate_shift(x, B, T, F, Fp, bn_s, bn_b, wq, b3d, *cw, wqf=wqf, src_order=True)
    idx = torch.tensor(ops.gs_source_order(F) + list(range(F, Fp)), device=DEV)
    assert torch.equal(src[:, idx], mod)
    ref = ops.gate_shift(x.float(), B, T, F, Fp, bn_s, bn_b, wq, b3d, *cw)
    assert float((mod.float() - ref).abs().max()) < 3e-2 * float(ref.abs().max())
