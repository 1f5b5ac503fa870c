"""Host logic of repack.PackPlan (no GPU): the index tables recorded from the modules' own repack() reproduce every
kernel-layout weight copy -- checked against independent torch / numpy expressions of the same layouts."""
import numpy as np
import torch

import tdeed_amd  # noqa: F401
from tdeed_amd import repack, state_layout, synth
from tdeed_amd.engine import pack_gconv_frags, pack_gsf_q_frags
from tdeed_amd.optim import FlatParams
from tdeed_amd.regnet_spec import regnet_spec
from tdeed_amd.temporal_train import TemporalStack
from tdeed_amd.trunk_train import BottleneckTrain, GateShiftTrain

CFG = dict(feature_arch="rny002_gsf", temporal_arch="ed_sgp_mixer", n_layers=2, sgp_ks=9, sgp_r=4, clip_len=16,
           num_classes=4, radi_displacement=2)


def _shells(sd, cfg, dt):
    """The train modules without their constructors (those run HIP kernels): only what repack() reads."""
    mods = []
    for b in regnet_spec(cfg["feature_arch"]).blocks:
        m = object.__new__(BottleneckTrain)
        pre = "_features." + b.name
        m.sd, m.pre, m.blk, m.dt, m.gs, m.epi_stats = sd, pre, b, dt, None, True
        m.c1 = pre + (".conv1.net" if b.gsf_fold else ".conv1")
        if b.gsf_fold:
            g = object.__new__(GateShiftTrain)
            g.sd, g.pre, g.F, g.T, g.dt, g.fuse = sd, pre + ".conv1.gs", b.gsf_fold, cfg["clip_len"], dt, True
            g.Fp = (b.gsf_fold + 7) // 8 * 8
            m.gs = g
        mods.append(m)
    t = object.__new__(TemporalStack)
    t.sd, t.dt, t.pre, t._cls_w = sd, dt, "_temp_fine.", {}
    t.C, t.n, t.T = sd["_temp_fine._sgp.0.ln.weight"].numel(), cfg["n_layers"], cfg["clip_len"]
    t.K1, t.radi = cfg["num_classes"] + 1, cfg["radi_displacement"]
    mods.append(t)
    return mods


def _plan(dt):
    sd = {k: torch.from_numpy(v) for k, v in synth.make_state(state_layout.model_state_shapes(CFG), 3).items()}
    flat = FlatParams(sd, "cpu")
    mods = _shells(sd, CFG, dt)
    plan = repack.PackPlan(flat, "cpu")
    plan.run = lambda: None                                       # the gather launch is the GPU's; emulated below
    plan.build(mods)
    _emulate(flat, plan)
    return sd, flat, mods, plan


def _emulate(flat, plan):
    """what PackPlan.run() launches, in torch"""
    ext = torch.cat([torch.zeros(1), flat.flat])
    for d, tab in plan.tables.items():
        assert tab.dtype == torch.int32 and tab.numel() % 8 == 0
        plan.bufs[d].copy_(ext[tab.long()].to(d))
    for op in plan.direct:
        if op[0] == "cast":
            _, off, n, out = op
            out.copy_(flat.flat[off:off + n].view(out.shape).to(torch.bfloat16))
        else:
            _, src, out = op
            out.copy_(src.t())


def test_index_tables_reproduce_the_packed_weights_bf16():
    sd, flat, mods, plan = _plan(torch.bfloat16)
    assert not repack.recording()
    assert plan.n_packed > 100 and plan.n_alias > 100
    assert any(op[0] == "cast" for op in plan.direct) and any(op[0] == "transpose" for op in plan.direct)   # large weights
    bf = lambda w: w.reshape(w.shape[0], -1).to(torch.bfloat16)
    for m in mods[:-1]:
        blk, pre = m.blk, m.pre
        w1, w3 = sd[m.c1 + ".conv.weight"], sd[pre + ".conv3.conv.weight"]
        assert torch.equal(m.w1.w, bf(w1)) and torch.equal(m.w1.wt, bf(w1).t().contiguous())
        assert torch.equal(m.w3.w, bf(w3)) and torch.equal(m.w3.wt, bf(w3).t().contiguous())
        if blk.has_downsample:
            wd = sd[pre + ".downsample.conv.weight"]
            assert torch.equal(m.wd.wt, bf(wd).t().contiguous())
        w2 = sd[pre + ".conv2.conv.weight"]
        G, gw = blk.groups, blk.gw
        assert torch.equal(m.w2p, w2.reshape(G, gw, gw, 3, 3).permute(0, 3, 4, 2, 1).reshape(G, 9, gw, gw))
        assert m.w2p.dtype == torch.float32
        assert torch.equal(m.w2frag, pack_gconv_frags(w2, gw, "cpu"))
        if blk.stride == 1:
            wt = w2.reshape(G, gw, gw, 3, 3).transpose(1, 2).flip(3, 4).reshape(blk.cout, gw, 3, 3)
            assert torch.equal(m.w2frag_t, pack_gconv_frags(wt, gw, "cpu"))
        R, C = blk.se_rd, blk.cout
        assert torch.equal(m.se_w1t, sd[pre + ".se.fc1.weight"].reshape(R, C).t())
        # plain views of the master buffer stay views (no copy to refresh)
        assert m.se_w1.data_ptr() == sd[pre + ".se.fc1.weight"].data_ptr()
        if m.gs is not None:
            g, gp = m.gs, m.gs.pre
            assert torch.equal(g.wqf, pack_gsf_q_frags(sd[gp + ".conv3D.weight"], "cpu"))
            assert torch.equal(g.wq, sd[gp + ".conv3D.weight"].reshape(g.F, 27).t())
            assert g.w_pad.shape == (g.Fp,) and torch.equal(g.w_pad[:g.F], sd[gp + ".bn.weight"])
            assert float(g.w_pad[g.F:].abs().sum()) == 0.0 and float(g.b_pad[g.F:].abs().sum()) == 0.0
            assert g.b3.data_ptr() == sd[gp + ".conv3D.bias"].data_ptr()
    t = mods[-1]
    C = t.C
    for i, o in enumerate(t.blocks):
        pre = f"_temp_fine._sgp.{i}"
        wm = sd[pre + ".mlp.0.weight"]
        assert torch.equal(o.fc1.w, bf(wm)) and torch.equal(o.fc1.wt, bf(wm).t().contiguous())
        names = ["psi", "convw", "convkw", "fc", "global_fc"]
        assert torch.equal(o.dw, torch.cat([sd[f"{pre}.{n}.weight"].reshape(C, -1) for n in names], dim=1))
        assert torch.equal(o.db, torch.stack([sd[f"{pre}.{n}.bias"].reshape(C) for n in names], dim=0))
        assert o.ln_w.data_ptr() == sd[pre + ".ln.weight"].data_ptr()
    for i, o in enumerate(t.mixers):
        pre = f"_temp_fine._sgpMixer.{i}"
        wc = sd[pre + ".concat_fc.weight"]
        assert torch.equal(o.cat.w, bf(wc)) and torch.equal(o.cat.wt, bf(wc).t().contiguous())


def test_a_parameter_update_reaches_every_copy_through_the_tables():
    sd, flat, mods, plan = _plan(torch.bfloat16)
    flat.flat.mul_(1.5)                                          # "optimizer step"
    _emulate(flat, plan)
    m = mods[3]
    w1 = sd[m.c1 + ".conv.weight"]
    assert torch.equal(m.w1.wt, w1.reshape(w1.shape[0], -1).to(torch.bfloat16).t().contiguous())
    assert torch.equal(m.w2frag, pack_gconv_frags(sd[m.pre + ".conv2.conv.weight"], m.blk.gw, "cpu"))


def test_fp32_mode_keeps_dense_weights_as_views():
    sd, flat, mods, plan = _plan(torch.float32)
    m = mods[2]
    w1 = sd[m.c1 + ".conv.weight"]
    assert m.w1.w.data_ptr() == w1.data_ptr() and m.w1.w.dtype == torch.float32
    assert torch.equal(m.w1.wt, w1.reshape(w1.shape[0], -1).t())
    assert torch.bfloat16 not in plan.tables                     # nothing is cast in the fp32 (parity) mode


def test_tap_map_fragments_of_both_forms_restate_the_conv3d():
    """The two MFMA layouts of a gate-shift site's conv3D weights (engine.pack_gsf_q_frags: implicit GEMM over (tap, channel);
    engine.pack_gsf_p_frags: one 1x1 contraction to per-tap sums, then nine adds -- the tail of tdeed_bneck_gs_fwd), emulated
    in numpy as the kernels read them, against the conv itself (impl/gsf.py:49-52 per frame: Q[jg] = conv2d(a, w3d[g][:, j]))."""
    from tdeed_amd.engine import _gsf_q_frags_np, _gsf_p_frags_np
    rng = np.random.default_rng(5)
    for F, h, w in ((92, 7, 7), (40, 5, 6), (16, 4, 3)):
        Fh, nch = F // 2, (F + 7) // 8
        w3d = rng.standard_normal((2, Fh, 3, 3, 3)).astype(np.float32)
        a = np.maximum(rng.standard_normal((h, w, nch * 8)), 0).astype(np.float32)
        a[..., F:] = 0
        ap = np.pad(a, ((1, 1), (1, 1), (0, 0)))
        ref = np.zeros((h, w, 6))
        for jg in range(6):
            jt, g = divmod(jg, 2)
            for dy in range(3):
                for dx in range(3):
                    ref[..., jg] += ap[dy:dy + h, dx:dx + w, g * Fh:(g + 1) * Fh].astype(np.float64) @ w3d[g, :, jt, dy, dx]
        # implicit GEMM: k-slot s = 4 ks + q = tap * nch + chunk, lane = q * 16 + row
        fq = _gsf_q_frags_np(w3d)
        q1 = np.zeros((h, w, 6))
        for ks in range(fq.shape[0]):
            for q in range(4):
                tap, ck = divmod(4 * ks + q, nch)
                if tap >= 9:
                    assert not fq[ks, q * 16:(q + 1) * 16].any()
                    continue
                dy, dx = divmod(tap, 3)
                q1 += ap[dy:dy + h, dx:dx + w, ck * 8:ck * 8 + 8].astype(np.float64) @ fq[ks, q * 16:q * 16 + 6].T
        # per-tap sums: row r = tap * 6 + jg of row tile r // 16, k = (4 ks + q) * 8 + e = channel
        fp = _gsf_p_frags_np(w3d)
        assert fp.shape == (4, (nch + 3) // 4, 64, 8)
        P = np.zeros((h, w, 64))
        for rt in range(4):
            for ks in range(fp.shape[1]):
                for q in range(4):
                    ck = 4 * ks + q
                    blk = fp[rt, ks, q * 16:(q + 1) * 16]                       # [row][e]
                    if ck >= nch:
                        assert not blk.any()
                        continue
                    P[..., rt * 16:(rt + 1) * 16] += a[..., ck * 8:ck * 8 + 8].astype(np.float64) @ blk.T
        assert not P[..., 54:].any()
        Pp = np.pad(P, ((1, 1), (1, 1), (0, 0)))
        q2 = np.zeros((h, w, 6))
        for tap in range(9):
            dy, dx = divmod(tap, 3)
            q2 += Pp[dy:dy + h, dx:dx + w, tap * 6:tap * 6 + 6]
        assert np.allclose(q1, ref, rtol=0, atol=1e-9 * max(1.0, np.abs(ref).max()))
        assert np.allclose(q2, ref, rtol=0, atol=1e-9 * max(1.0, np.abs(ref).max()))
