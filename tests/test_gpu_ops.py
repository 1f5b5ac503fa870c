"""Op-level parity of the HIP kernels (through the C ABI) against the CPU oracle / torch fp32
references and the reference-generated golden vectors.  Needs an MI355X: run with -m gpu."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from helpers import load_golden, act, module_state, t, max_abs
from tdeed_amd import synth

pytestmark = pytest.mark.gpu

DEV = "cuda"
F32_TOL = 1e-4          # kernels in TDEED_F32 mode vs fp32 CPU (relative to output magnitude)
BF16_TOL = 3e-2         # bf16 storage, fp32 accumulation


@pytest.fixture(scope="module")
def ops():
    from tdeed_amd import ops as o, _lib
    _lib.load()
    return o


def rel_err(a, b):
    b = b.detach().cpu().double() if isinstance(b, torch.Tensor) else torch.as_tensor(b).double()
    a = a.detach().cpu().double()
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-6))


def rnd(seed, name, shape, scale=1.0):
    return t(act(seed, name, shape, scale))


# ----------------------------------------------------------------------------- GEMM
GEMM_SHAPES = [(300, 32, 24), (257, 24, 56), (1000, 56, 152), (129, 152, 368), (640, 368, 368),
               (200, 368, 1472), (200, 1472, 368), (75, 2208, 368), (128, 64, 64), (5000, 128, 320)]


@pytest.mark.parametrize("M,K,N", GEMM_SHAPES)
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_gemm_plain(ops, M, K, N, dtype):
    A = rnd(1, f"A{M}", (M, K)).to(dtype)
    W = rnd(2, f"W{N}", (N, K), 1.0 / np.sqrt(K)).to(dtype)
    sc = rnd(3, "sc", (N,)) * 0.2 + 1.0
    sh = rnd(4, "sh", (N,))
    ref = (A.float() @ W.float().T) * sc + sh
    out = ops.gemm(A.to(DEV), W.to(DEV), sc.to(DEV), sh.to(DEV), ops.ACT_NONE)
    assert rel_err(out.float(), ref) < (F32_TOL if dtype == torch.float32 else BF16_TOL)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("actn", [0, 1, 2])
def test_gemm_epilogue(ops, dtype, actn):
    M, K, N = 333, 152, 152
    A = rnd(5, "A", (M, K)).to(dtype)
    W = rnd(6, "W", (N, K), 0.1).to(dtype)
    R = rnd(7, "R", (M, N)).to(dtype)
    sh = rnd(8, "sh", (N,))
    ref = A.float() @ W.float().T + sh + R.float()
    ref = [ref, torch.relu(ref), F.gelu(ref)][actn]
    out = ops.gemm(A.to(DEV), W.to(DEV), None, sh.to(DEV), actn, residual=R.to(DEV))
    assert rel_err(out.float(), ref) < (F32_TOL if dtype == torch.float32 else BF16_TOL)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_gemm_se_scale_splice_gather(ops, dtype):
    # SE gate on A (per frame), gate-shift splice in front, and the stride-2 row gather
    Fr, hw, K, N = 6, 49, 152, 368
    M = Fr * hw
    A = rnd(9, "A", (M, K)).to(dtype)
    W = rnd(10, "W", (N, K), 0.1).to(dtype)
    gate = torch.sigmoid(rnd(11, "g", (Fr, K)))
    ref = (A.float().view(Fr, hw, K) * gate[:, None, :])
    if dtype == torch.bfloat16:
        ref = ref.to(torch.bfloat16).float()        # the kernel rounds the scaled operand to bf16
    ref = ref.view(M, K) @ W.float().T
    out = ops.gemm(A.to(DEV), W.to(DEV), a_scale=gate.to(DEV), a_scale_rows=hw)
    assert rel_err(out.float(), ref) < (F32_TOL if dtype == torch.float32 else BF16_TOL)

    k0 = 48
    A0 = rnd(12, "A0", (M, k0)).to(dtype)
    Asp = torch.cat([A0, A[:, k0:]], dim=1)
    ref = Asp.float() @ W.float().T
    out = ops.gemm(A.to(DEV), W.to(DEV), A0=A0.to(DEV), k0=k0)
    assert rel_err(out.float(), ref) < (F32_TOL if dtype == torch.float32 else BF16_TOL)

    hi, wi, s = 7, 9, 2
    ho, wo = (hi - 1) // s + 1, (wi - 1) // s + 1
    X = rnd(13, "X", (3, hi, wi, K)).to(dtype)
    ref = X.float()[:, ::s, ::s, :].reshape(-1, K) @ W.float().T
    out = ops.gemm(X.to(DEV), W.to(DEV), gather=(s, hi, wi, ho, wo))
    assert out.shape[0] == 3 * ho * wo
    assert rel_err(out.float(), ref) < (F32_TOL if dtype == torch.float32 else BF16_TOL)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("M,K,N,n2", [(333, 152, 152, 40), (4097, 152, 152, 40), (1999, 368, 368, 96), (640, 56, 56, 16),
                                      (257, 56, 152, 40)])
def test_gemm_second_compact_output(ops, M, K, N, n2, dtype):
    """out2: the epilogue also stores columns [0, n2) of C as a compact (M, n2) tensor -- the next bottleneck's gate-shift
    slice (shift.py:46-93 reads only those channels).  Bit-identical to slicing C, in the tiled and the weight-stationary
    kernel, with the full epilogue (residual + ReLU) in front; guard columns behind n2 stay untouched."""
    from tdeed_amd.engine import pack_ws_weights
    A = rnd(201, f"A{M}", (M, K)).to(dtype).to(DEV)
    W = rnd(202, f"W{N}", (N, K), 1.0 / np.sqrt(K)).to(dtype)
    R = rnd(203, "R", (M, N)).to(dtype).to(DEV)
    sh = rnd(204, "sh", (N,)).to(DEV)
    out2 = torch.full((M, n2), 7.0, dtype=dtype, device=DEV)
    out = ops.gemm(A, W.to(DEV), None, sh, ops.ACT_RELU, residual=R, out2=out2)
    assert torch.equal(out2, out[:, :n2])
    assert torch.equal(out, ops.gemm(A, W.to(DEV), None, sh, ops.ACT_RELU, residual=R))
    if ops.gemm_ws_fits(K, N, dtype):
        Wf = pack_ws_weights(W.float().numpy(), dtype, DEV)
        out2.fill_(7.0)
        out = ops.gemm_ws(A, Wf, K, N, None, sh, ops.ACT_RELU, residual=R, out2=out2)
        assert torch.equal(out2, out[:, :n2])
        assert torch.equal(out, ops.gemm_ws(A, Wf, K, N, None, sh, ops.ACT_RELU, residual=R))
    with pytest.raises(Exception, match="second output"):
        ops.gemm(A, W.to(DEV), None, sh, ops.ACT_RELU, out2=torch.empty((M, N + 8), dtype=dtype, device=DEV))
    # out2_pre (input gradient of a gate-shifted conv1): out2 = the contraction alone, C keeps only the residual in those columns
    plain = ops.gemm(A, W.to(DEV), None, None, ops.ACT_NONE)
    full = ops.gemm(A, W.to(DEV), None, None, ops.ACT_NONE, residual=R)
    out2.fill_(7.0)
    out = ops.gemm(A, W.to(DEV), None, None, ops.ACT_NONE, residual=R, out2=out2, out2_pre=True)
    assert torch.equal(out2, plain[:, :n2]) and torch.equal(out[:, :n2], R[:, :n2]) and torch.equal(out[:, n2:], full[:, n2:])
    out = ops.gemm(A, W.to(DEV), None, None, ops.ACT_NONE, out2=out2, out2_pre=True)
    assert torch.equal(out2, plain[:, :n2]) and float(out[:, :n2].abs().max()) == 0.0 and torch.equal(out[:, n2:], plain[:, n2:])


WS_SHAPES = [(300, 32, 24), (1000, 24, 24), (257, 24, 56), (640, 56, 56), (999, 56, 152), (4097, 152, 152),
             (500, 64, 128), (130, 128, 128), (777, 368, 368), (300, 152, 368)]


@pytest.mark.parametrize("M,K,N", WS_SHAPES)
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_gemm_ws(ops, M, K, N, dtype):
    """weight-stationary kernel == tiled kernel semantics (plain, epilogue, SE scale, splice, gather)."""
    from tdeed_amd.engine import pack_ws_weights
    if not ops.gemm_ws_fits(K, N, dtype):
        pytest.skip("does not fit the weight-stationary kernel in this dtype")
    tol = F32_TOL if dtype == torch.float32 else BF16_TOL
    A = rnd(71, f"A{M}", (M, K)).to(dtype)
    W = rnd(72, f"W{N}", (N, K), 1.0 / np.sqrt(K)).to(dtype)
    Wf = pack_ws_weights(W.float().numpy(), dtype, DEV)
    sc, sh = rnd(73, "sc", (N,)) * 0.2 + 1.0, rnd(74, "sh", (N,))
    R = rnd(75, "R", (M, N)).to(dtype)
    base = A.float() @ W.float().T
    out = ops.gemm_ws(A.to(DEV), Wf, K, N, sc.to(DEV), sh.to(DEV), ops.ACT_NONE)
    assert rel_err(out.float(), base * sc + sh) < tol
    for actn, fn in [(1, torch.relu), (2, F.gelu)]:
        out = ops.gemm_ws(A.to(DEV), Wf, K, N, None, sh.to(DEV), actn, residual=R.to(DEV))
        assert rel_err(out.float(), fn(base + sh + R.float())) < tol
    rows = 7
    Mp = (M // rows) * rows
    gate = torch.sigmoid(rnd(76, "g", (Mp // rows, K)))
    ref = A[:Mp].float().view(-1, rows, K) * gate[:, None, :]
    if dtype == torch.bfloat16:
        ref = ref.to(torch.bfloat16).float()
    out = ops.gemm_ws(A[:Mp].contiguous().to(DEV), Wf, K, N, a_scale=gate.to(DEV), a_scale_rows=rows)
    assert rel_err(out.float(), ref.view(Mp, K) @ W.float().T) < tol
    if K >= 16:
        k0 = 8
        A0 = rnd(77, "A0", (M, k0)).to(dtype)
        out = ops.gemm_ws(A.to(DEV), Wf, K, N, A0=A0.to(DEV), k0=k0)
        assert rel_err(out.float(), torch.cat([A0, A[:, k0:]], 1).float() @ W.float().T) < tol
    hi, wi, s_ = 6, 10, 2
    X = rnd(78, "X", (3, hi, wi, K)).to(dtype)
    out = ops.gemm_ws(X.to(DEV), Wf, K, N, gather=(s_, hi, wi, 3, 5))
    assert rel_err(out.float(), X.float()[:, ::2, ::2, :].reshape(-1, K) @ W.float().T) < tol


# ----------------------------------------------------------------------------- stem / grouped conv / SE / pool
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("geom", [(2, 64, 64, None, False), (1, 72, 80, (4, 8, 64, 64), True), (1, 50, 37, None, False)])
def test_stem(ops, dtype, geom):
    from oracle import tdeed_oracle as O
    N, H, W, crop, flip = geom
    fr = synth.uint8_clip(21, (N, 3, H, W))
    w = rnd(22, "w", (32, 3, 3, 3), 0.3)
    sc = rnd(23, "sc", (32,)) * 0.2 + 1.0
    sh = rnd(24, "sh", (32,)) * 0.1
    x = t(fr).float() / 255.0
    if crop:
        x = x[..., crop[0]:crop[0] + crop[2], crop[1]:crop[1] + crop[3]]
    if flip:
        x = x.flip(-1)
    mean = torch.tensor(O.IMAGENET_MEAN).view(1, 3, 1, 1)
    std = torch.tensor(O.IMAGENET_STD).view(1, 3, 1, 1)
    x = (x - mean) / std
    ref = torch.relu(F.conv2d(x, w, stride=2, padding=1) * sc.view(1, -1, 1, 1) + sh.view(1, -1, 1, 1))
    out = ops.stem(t(fr).to(DEV), w.reshape(32, 27).contiguous().to(DEV), sc.to(DEV), sh.to(DEV), dtype, crop, flip)
    assert rel_err(out.float().permute(0, 3, 1, 2), ref) < (F32_TOL if dtype == torch.float32 else 1e-2)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("C,gw,stride,H,W", [(24, 8, 2, 20, 22), (56, 8, 1, 9, 7), (64, 16, 2, 16, 16),
                                             (368, 8, 1, 7, 7), (128, 16, 1, 28, 28), (152, 8, 2, 28, 28),
                                             (24, 8, 2, 112, 112), (320, 16, 1, 14, 14), (768, 16, 2, 14, 14)])
def test_gconv3x3(ops, dtype, C, gw, stride, H, W):
    N = 3
    x = rnd(31, "x", (N, C, H, W)).to(dtype)
    w = rnd(32, "w", (C, gw, 3, 3), 0.2)
    sc = rnd(33, "sc", (C,)) * 0.2 + 1.0
    sh = rnd(34, "sh", (C,)) * 0.1
    ref = torch.relu(F.conv2d(x.float(), w, stride=stride, padding=1, groups=C // gw) * sc.view(1, -1, 1, 1)
                     + sh.view(1, -1, 1, 1))
    G = C // gw
    wp = w.reshape(G, gw, gw, 3, 3).permute(0, 3, 4, 2, 1).reshape(G, 9, gw, gw).contiguous()
    from tdeed_amd.engine import pack_gconv_frags
    xin = x.permute(0, 2, 3, 1).contiguous().to(DEV)
    tol = F32_TOL if dtype == torch.float32 else BF16_TOL
    variants = [None] if dtype == torch.float32 else [None, pack_gconv_frags(w, gw, DEV)]   # VALU, MFMA
    for wfrag in variants:
        y, pooled = ops.gconv3x3(xin, wp.to(DEV), sc.to(DEV), sh.to(DEV), gw, stride, wfrag=wfrag)
        assert rel_err(y.float().permute(0, 3, 1, 2), ref) < tol
        npix = y.shape[1] * y.shape[2]
        assert rel_err(pooled.sum(dim=1) / npix, ref.mean(dim=(2, 3))) < tol


@pytest.mark.parametrize("M,K,N", [(800, 1472, 368), (400, 368, 1472), (200, 2208, 368), (37, 40, 24), (1000, 4608, 768)])
@pytest.mark.parametrize("actn", [0, 1, 2])
def test_gemm_splitk(ops, M, K, N, actn):
    A, W = rnd(141, "A", (M, K)).to(torch.bfloat16), rnd(142, "W", (N, K), 0.05).to(torch.bfloat16)
    sc, sh, R = rnd(143, "sc", (N,)) * 0.2 + 1.0, rnd(144, "sh", (N,), 0.1), rnd(145, "R", (M, N)).to(torch.bfloat16)
    ref = (A.float() @ W.float().T) * sc + sh + R.float()
    ref = [ref, torch.relu(ref), F.gelu(ref)][actn]
    out = ops.gemm_splitk(A.to(DEV), W.to(DEV), sc.to(DEV), sh.to(DEV), actn, residual=R.to(DEV))
    assert rel_err(out.float(), ref) < BF16_TOL
    out2 = ops.gemm_splitk(A.to(DEV), W.to(DEV), None, None, actn)
    ref2 = A.float() @ W.float().T
    assert rel_err(out2.float(), [ref2, torch.relu(ref2), F.gelu(ref2)][actn]) < BF16_TOL


def test_se_gate(ops):
    N, C, R = 11, 152, 38
    p = rnd(41, "p", (N, C)).abs()
    w1, b1 = rnd(42, "w1", (R, C), 0.1), rnd(43, "b1", (R,), 0.1)
    w2, b2 = rnd(44, "w2", (C, R), 0.2), rnd(45, "b2", (C,), 0.1)
    ref = torch.sigmoid(torch.relu(p @ w1.T + b1) @ w2.T + b2)
    parts = torch.stack([p * 0.25 * 7, p * 0.75 * 7], dim=1).contiguous()      # two partial sums over 7 "pixels"
    out = ops.se_gate(parts.to(DEV), 1.0 / 7, w1.T.contiguous().to(DEV), b1.to(DEV), w2.T.contiguous().to(DEV), b2.to(DEV))
    assert rel_err(out, ref) < 1e-5


@pytest.mark.parametrize("N,C,R", [(11, 152, 38), (800, 368, 92), (5, 24, 8), (7, 768, 192), (9, 56, 6)])
def test_se_gate_bf16(ops, N, C, R):
    from tdeed_amd.engine import pack_se_bf16
    p = rnd(46, "p", (N, 3, C)).abs()
    w1, b1 = rnd(47, "w1", (R, C), 0.1).to(torch.bfloat16).float(), rnd(48, "b1", (R,), 0.1)
    w2, b2 = rnd(49, "w2", (C, R), 0.2).to(torch.bfloat16).float(), rnd(50, "b2", (C,), 0.1)
    ref = torch.sigmoid(torch.relu((p.sum(1) / 5.0) @ w1.T + b1) @ w2.T + b2)
    pk = pack_se_bf16(w1.numpy(), w2.numpy(), DEV)
    out = ops.se_gate_bf16(p.to(DEV), 1.0 / 5.0, pk["se_w1p"], b1.to(DEV), pk["se_w2p"], b2.to(DEV), R)
    assert rel_err(out, ref) < 1e-4


@pytest.mark.parametrize("N,C,R,parts", [(11, 152, 38, 1), (400, 368, 92, 1), (35, 24, 8, 56), (7, 56, 6, 10), (9, 152, 14, 2),
                                          (16, 368, 38, 1),
                                          # the 8-wave form of the wide layers (RegNetY-800MF s4: 768 channels, 80 / 192 hidden)
                                          (37, 768, 192, 1), (400, 768, 80, 1), (5, 520, 130, 3), (33, 384, 100, 1)])
def test_se_gate_mfma(ops, N, C, R, parts):
    from tdeed_amd.engine import pack_se_mfma
    assert ops.se_gate_mfma_fits(C, R) and not ops.se_gate_mfma_fits(776, 192) and not ops.se_gate_mfma_fits(768, 200)
    p = rnd(146, "p", (N, parts, C)).abs()
    w1, b1 = rnd(147, "w1", (R, C), 0.1).to(torch.bfloat16).float(), rnd(148, "b1", (R,), 0.1)
    w2, b2 = rnd(149, "w2", (C, R), 0.2).to(torch.bfloat16).float(), rnd(150, "b2", (C,), 0.1)
    ref = torch.sigmoid(torch.relu((p.sum(1) / 5.0) @ w1.T + b1) @ w2.T + b2)
    pk = pack_se_mfma(w1.numpy(), w2.numpy(), DEV)
    out = ops.se_gate_mfma(p.to(DEV), 1.0 / 5.0, pk["w1f"], b1.to(DEV), pk["w2f"], b2.to(DEV), R)
    assert rel_err(out, ref) < 1e-4


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_avgpool_posenc(ops, dtype):
    B, T, hw, C = 2, 5, 49, 368
    x = rnd(51, "x", (B * T, 7, 7, C)).to(dtype)
    te = rnd(52, "te", (T, C), 0.1)
    ref = x.float().mean(dim=(1, 2)).view(B, T, C) + te[None]
    out = ops.avgpool_posenc(x.to(DEV), B, T, te.to(DEV))
    assert rel_err(out.float(), ref) < (F32_TOL if dtype == torch.float32 else 1e-2)


@pytest.mark.parametrize("geom", [(2, 64, 64, None, False, 24, 8), (1, 72, 80, (4, 8, 64, 64), True, 24, 8),
                                  (1, 224, 224, None, False, 24, 8), (1, 96, 64, None, False, 64, 16),
                                  # pipelined strip kernel: aligned crop + flip, one-tile width (C1 = 16), ragged last strip,
                                  # odd crop (-> rolling kernel)
                                  (1, 96, 112, (16, 16, 64, 96), True, 24, 8), (2, 64, 64, None, True, 16, 8),
                                  (1, 240, 224, None, False, 24, 8), (1, 70, 90, (3, 5, 61, 77), True, 24, 8)])
def test_s1_front_fused_vs_unfused_reference(ops, geom):
    """stem -> conv1 -> conv2 (+ squeeze) and the stride-2 shortcut in one kernel == the same chain in torch fp32."""
    from tdeed_amd.engine import pack_front_weights
    from oracle import tdeed_oracle as O
    N, H, W, crop, flip, C1, gw = geom
    fr = synth.uint8_clip(81, (N, 3, H, W))
    sw = rnd(82, "sw", (32, 3, 3, 3), 0.3)
    w1, wd = rnd(83, "w1", (C1, 32), 0.25), rnd(84, "wd", (C1, 32), 0.25)
    w2 = rnd(85, "w2", (C1, gw, 3, 3), 0.15)
    aff = lambda i, c: (rnd(90 + i, "s", (c,)) * 0.2 + 1.0, rnd(95 + i, "h", (c,)) * 0.1)   # noqa: E731
    (ss, sh), (s1, h1), (sd_, hd), (s2, h2) = aff(0, 32), aff(1, C1), aff(2, C1), aff(3, C1)
    x = t(fr).float() / 255.0
    if crop:
        x = x[..., crop[0]:crop[0] + crop[2], crop[1]:crop[1] + crop[3]]
    if flip:
        x = x.flip(-1)
    x = (x - torch.tensor(O.IMAGENET_MEAN).view(1, 3, 1, 1)) / torch.tensor(O.IMAGENET_STD).view(1, 3, 1, 1)
    v = lambda a: a.view(1, -1, 1, 1)                                                       # noqa: E731
    st = torch.relu(F.conv2d(x, sw, stride=2, padding=1) * v(ss) + v(sh))
    y1 = torch.relu(F.conv2d(st, w1.view(C1, 32, 1, 1)) * v(s1) + v(h1))
    y2 = torch.relu(F.conv2d(y1, w2, stride=2, padding=1, groups=C1 // gw) * v(s2) + v(h2))
    scut = F.conv2d(st, wd.view(C1, 32, 1, 1), stride=2) * v(sd_) + v(hd)
    fw = pack_front_weights(sw, ss, sh, w1, s1, h1, wd, sd_, hd, w2, gw, s2, h2, DEV)
    g2, gs, gp = ops.s1_front(t(fr).to(DEV), fw, crop, flip)
    assert g2.shape[1:3] == y2.shape[2:]
    assert rel_err(g2.float().permute(0, 3, 1, 2), y2) < 4e-2
    assert rel_err(gs.float().permute(0, 3, 1, 2), scut) < 4e-2
    npix = y2.shape[2] * y2.shape[3]
    assert rel_err(gp.sum(dim=1) / npix, y2.mean(dim=(2, 3))) < 4e-2


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("name", ["gsf_f16", "gsf_f40", "gsf_f92", "gsm_f16"])
def test_gate_shift_golden(ops, name, dtype):
    meta, g = load_golden(name)
    Fd, T, B, h, w, mode = meta["F"], meta["T"], meta["B"], meta["h"], meta["w"], meta["mode"]
    sd = module_state("gate_shift", "gs", meta["seed"], F=Fd, mode=mode)
    x = act(meta["seed"], name + ":x", (B * T, Fd, h, w))
    C = (Fd + 7) // 8 * 8 + 8                      # a few pass-through channels behind the fold
    xin = np.zeros((B * T, h, w, C), np.float32)
    xin[..., :Fd] = x.transpose(0, 2, 3, 1)
    xin[..., Fd:] = act(99, "pad", (B * T, h, w, C - Fd))
    bnw, bnb = sd["gs.bn.weight"].astype(np.float64), sd["gs.bn.bias"].astype(np.float64)
    s = bnw / np.sqrt(sd["gs.bn.running_var"].astype(np.float64) + 1e-5)
    sh = bnb - sd["gs.bn.running_mean"].astype(np.float64) * s
    dev = lambda a: t(np.ascontiguousarray(a, dtype=np.float32)).to(DEV)   # noqa: E731
    kw = {}
    if mode == "gsf":
        kw = dict(cw1=dev(sd["gs.channel_conv1.weight"].reshape(18)), cb1=dev(sd["gs.channel_conv1.bias"]),
                  cw2=dev(sd["gs.channel_conv2.weight"].reshape(18)), cb2=dev(sd["gs.channel_conv2.bias"]))
    Fp = (Fd + 7) // 8 * 8
    if dtype == torch.bfloat16:
        from tdeed_amd.engine import pack_gsf_q_frags
        kw["wqf"] = pack_gsf_q_frags(sd["gs.conv3D.weight"], DEV)       # MFMA partial-sum kernel
    out = ops.gate_shift(t(xin).to(dtype).to(DEV), B, T, Fd, Fp, dev(s), dev(sh),
                         dev(sd["gs.conv3D.weight"].reshape(Fd, 27).T), dev(sd["gs.conv3D.bias"]), **kw)
    if mode == "gsf":      # the 3-launch form (separate fusion-weight kernel) gives the same bits as the fused apply
        out3 = ops.gate_shift(t(xin).to(dtype).to(DEV), B, T, Fd, Fp, dev(s), dev(sh),
                              dev(sd["gs.conv3D.weight"].reshape(Fd, 27).T), dev(sd["gs.conv3D.bias"]),
                              separate_weight=True, **kw)
        assert torch.equal(out3, out)
    if mode == "gsf" and dtype == torch.bfloat16:
        # the launch that leaves the slice in source channel order (the engine folds the interleave into conv1's columns):
        # the module's output channel co is its channel gs_source_order(F)[co], bit for bit; pad columns stay copies of x
        src = ops.gate_shift(t(xin).to(dtype).to(DEV), B, T, Fd, Fp, dev(s), dev(sh),
                             dev(sd["gs.conv3D.weight"].reshape(Fd, 27).T), dev(sd["gs.conv3D.bias"]), src_order=True, **kw)
        idx = torch.tensor(ops.gs_source_order(Fd) + list(range(Fd, Fp)), device=DEV)
        assert torch.equal(src[:, idx], out)
    out = out.float().cpu().view(B * T, h, w, Fp)
    ref = t(g["y"]).permute(0, 2, 3, 1)
    tol = 2e-5 if dtype == torch.float32 else 3e-2
    assert max_abs(out[..., :Fd], ref) < tol * max(1.0, float(ref.abs().max()))
    if Fp > Fd:   # padding columns are copies of x
        assert max_abs(out[..., Fd:], t(xin)[..., Fd:Fp].to(dtype).float()) == 0.0


# ----------------------------------------------------------------------------- SGP pieces
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_layernorm_pool_golden(ops, dtype):
    meta, g = load_golden("misc_ops")
    sd = module_state("ln", "ln", meta["seed"], C=48)
    x = act(meta["seed"], "ln:x", (2, 48, 25)).transpose(0, 2, 1)          # NTC
    dev = lambda a: t(np.ascontiguousarray(a, dtype=np.float32)).to(DEV)   # noqa: E731
    y = ops.layernorm(dev(x).to(dtype), dev(sd["ln.weight"].reshape(-1)), dev(sd["ln.bias"].reshape(-1)))
    ref = t(g["ln_y"]).permute(0, 2, 1)
    assert max_abs(y.float().cpu(), ref) < (2e-5 if dtype == torch.float32 else 3e-2)
    for (L, o) in [(25, 13), (125, 63), (100, 50), (13, 7)]:
        xp = act(meta["seed"], f"pool{L}:x", (2, 16, L)).transpose(0, 2, 1)
        yp = ops.maxpool(dev(xp).to(dtype), o)
        refp = t(g[f"pool_{L}_{o}"]).permute(0, 2, 1)
        if dtype == torch.float32:
            assert torch.equal(yp.cpu(), refp)
        else:
            assert torch.equal(yp.cpu(), refp.to(torch.bfloat16))


def _run_steps(steps):
    for s_ in steps:
        s_.fn()


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("name", ["sgp_block_c32_t25", "sgp_block_c368_t100", "sgp_block_c48_t13"])
def test_sgp_block_golden(ops, name, dtype):
    from tdeed_amd.engine import SgpBuilder, pack_sgp_block, _Pool
    meta, g = load_golden(name)
    B, C, T = meta["B"], meta["C"], meta["T"]
    sd = module_state("sgp_block", "blk", meta["seed"], **meta)
    x = t(act(meta["seed"], name + ":x", (B, C, T)).transpose(0, 2, 1).copy()).to(dtype).to(DEV)
    steps, keep = [], {}
    sb = SgpBuilder(_Pool(DEV), steps, keep, set(), B, dtype)
    out = sb.block(x, T, pack_sgp_block(sd, "blk", C, dtype, DEV), "blk")
    _run_steps(steps)
    ref = t(g["y"]).permute(0, 2, 1)
    tol = 1e-4 if dtype == torch.float32 else 4e-2
    assert max_abs(out.float().cpu(), ref) < tol * max(1.0, float(ref.abs().max()))


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("name", ["sgp_mixer_c32_t25", "sgp_mixer_c368_t100"])
def test_sgp_mixer_golden(ops, name, dtype):
    from tdeed_amd.engine import SgpBuilder, pack_sgp_mixer, _Pool
    meta, g = load_golden(name)
    B, C, Th, Tl = meta["B"], meta["C"], meta["T_hi"], meta["T_lo"]
    sd = module_state("sgp_mixer", "mix", meta["seed"], **meta)
    z = t(act(meta["seed"], name + ":z", (B, C, Th)).transpose(0, 2, 1).copy()).to(dtype).to(DEV)
    x = t(act(meta["seed"], name + ":x", (B, C, Tl)).transpose(0, 2, 1).copy()).to(dtype).to(DEV)
    steps, keep = [], {}
    sb = SgpBuilder(_Pool(DEV), steps, keep, set(), B, dtype)
    out = sb.mixer(x, Tl, z, Th, pack_sgp_mixer(sd, "mix", C, dtype, DEV), "mix")
    _run_steps(steps)
    ref = t(g["y"]).permute(0, 2, 1)
    tol = 1e-4 if dtype == torch.float32 else 4e-2
    assert max_abs(out.float().cpu(), ref) < tol * max(1.0, float(ref.abs().max()))


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("name", ["pyramid_c32_l25_n2", "pyramid_c64_l100_n3", "pyramid_c48_l250_n2"])
def test_pyramid_golden(ops, name, dtype):
    from tdeed_amd.engine import SgpBuilder, pack_sgp_block, pack_sgp_mixer, _Pool
    meta, g = load_golden(name)
    B, C, L, n = meta["B"], meta["C"], meta["L"], meta["n"]
    sd = module_state("pyramid", "_temp_fine", meta["seed"], **meta)
    x = t(act(meta["seed"], name + ":x", (B, L, C))).to(dtype).to(DEV)
    sgp = [pack_sgp_block(sd, f"_temp_fine._sgp.{i}", C, dtype, DEV) for i in range(2 * n + 1)]
    mix = [pack_sgp_mixer(sd, f"_temp_fine._sgpMixer.{i}", C, dtype, DEV) for i in range(n)]
    steps, keep = [], {}
    sb = SgpBuilder(_Pool(DEV), steps, keep, set(), B, dtype)
    out = sb.pyramid(x, L, n, sgp, mix)
    _run_steps(steps)
    ref = t(g["y"])
    tol = 2e-4 if dtype == torch.float32 else 6e-2
    assert max_abs(out.float().cpu(), ref) < tol * max(1.0, float(ref.abs().max()))


# ----------------------------------------------------------------------------- heads / loss / post-proc
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_heads(ops, dtype):
    rows, C, n_out = 37, 368, 6
    x = rnd(61, "x", (rows, C)).to(dtype)
    w, b = rnd(62, "w", (n_out, C), 0.05), rnd(63, "b", (n_out,), 0.1)
    out = ops.heads(x.to(DEV), w.to(DEV), b.to(DEV))
    assert rel_err(out, x.float() @ w.T + b) < 1e-5


def test_loss_and_process_prediction_golden(ops):
    meta, g = load_golden("loss_postproc")
    B, T, K1, seed = meta["B"], meta["T"], meta["K1"], meta["seed"]
    logits = act(seed, "logits", (B, T, K1), 2.0)
    displ = act(seed, "displ", (B, T), 1.5)
    lab, labD = synth.labels(seed, B, T, K1 - 1, 2, fg_frac=0.3)
    head = t(np.concatenate([logits, displ[..., None]], axis=-1).reshape(B * T, K1 + 1)).to(DEV)
    w = torch.tensor([1.0] + [5.0] * (K1 - 1), device=DEV)
    out = ops.loss(head, K1, w, hard=t(lab.reshape(-1)).to(DEV), displ_col=K1,
                   labelD=t(labD.reshape(-1).astype(np.float32)).to(DEV)).cpu()
    assert abs(float(out[1]) - float(g["ce_hard"])) < 1e-5
    assert abs(float(out[2]) - float(g["mse"])) < 1e-5
    assert abs(float(out[0]) - float(g["ce_hard"]) - float(g["mse"])) < 1e-5
    out = ops.loss(head, K1, w, soft=t(g["soft_labels"].reshape(B * T, K1)).to(DEV)).cpu()
    assert abs(float(out[1]) - float(g["ce_soft"])) < 1e-5
    cls, scores = ops.process_prediction(head, B, T, K1, K1)
    assert max_abs(scores.cpu(), g["process_prediction"]) < 1e-6
    assert np.array_equal(cls.cpu().numpy(), g["process_prediction"].argmax(-1))
    head2 = t(np.concatenate([logits, g["d_half"][..., None]], axis=-1).reshape(B * T, K1 + 1)).to(DEV)
    _, scores = ops.process_prediction(head2, B, T, K1, K1)       # .5 displacements: round-half-even
    assert max_abs(scores.cpu(), g["process_prediction_half"]) < 1e-6
    from tdeed_amd import modules
    s2 = modules.process_double_head(t(logits).to(DEV), t(displ).to(DEV), num_classes=3)
    assert max_abs(s2.cpu(), g["process_double_head"]) < 1e-6


# ----------------------------------------------------------------------------- training path, first pieces
@pytest.mark.parametrize("soft", [False, True])
def test_loss_bwd_matches_autograd(ops, soft):
    from oracle import tdeed_oracle as O
    B, T, K1, seed = 3, 20, 5, 8
    logits = t(act(seed, "logits", (B, T, K1), 2.0)).requires_grad_(True)
    displ = t(act(seed, "displ", (B, T), 1.5)).requires_grad_(True)
    lab, labD = synth.labels(seed, B, T, K1 - 1, 2, fg_frac=0.3)
    _, g = load_golden("loss_postproc")
    label = t(g["soft_labels"]) if soft else t(lab)
    O.loss_fn(logits, label, displ, t(labD)).backward()
    head = torch.cat([logits.detach(), displ.detach()[..., None]], -1).reshape(B * T, K1 + 1).contiguous().to(DEV)
    w = torch.tensor([1.0] + [5.0] * (K1 - 1), device=DEV)
    kw = dict(soft=label.reshape(B * T, K1).contiguous().to(DEV)) if soft else dict(hard=label.reshape(-1).to(DEV))
    d = ops.loss_bwd(head, K1, w, displ_col=K1, labelD=t(labD.reshape(-1).astype(np.float32)).to(DEV), **kw).cpu()
    assert max_abs(d[:, :K1], logits.grad.reshape(B * T, K1)) < 1e-6
    assert max_abs(d[:, K1], displ.grad.reshape(-1)) < 1e-6


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_heads_bwd_matches_autograd(ops, dtype):
    rows, C, n_out = 213, 368, 6
    x = rnd(131, "x", (rows, C)).to(dtype)
    w = rnd(132, "w", (n_out, C), 0.05)
    dout = rnd(133, "dout", (rows, n_out))
    xr = x.float().requires_grad_(True)
    wr = w.clone().requires_grad_(True)
    br = torch.zeros(n_out, requires_grad=True)
    (F.linear(xr, wr, br) * dout).sum().backward()
    dx, dw, db = ops.heads_bwd(dout.to(DEV), x.to(DEV), w.to(DEV))
    assert rel_err(dx.float(), xr.grad) < (1e-5 if dtype == torch.float32 else 1e-2)
    assert rel_err(dw, wr.grad) < 1e-5 and rel_err(db, br.grad) < 1e-5


def test_adamw_step_matches_torch_optim(ops):
    n = 100003                      # not a multiple of 4: exercises the scalar tail
    p0 = rnd(141, "p", (n,))
    opt_p = p0.clone().requires_grad_(True)
    opt = torch.optim.AdamW([opt_p], lr=8e-4)           # reference defaults: betas (0.9,0.999), eps 1e-8, wd 0.01
    p = p0.clone().to(DEV)
    m, v = torch.zeros(n, device=DEV), torch.zeros(n, device=DEV)
    for step in range(1, 4):
        g = rnd(150 + step, "g", (n,))
        opt_p.grad = g.clone()
        opt.step()
        ops.adamw_step(p, g.to(DEV), m, v, step, 8e-4)
        assert max_abs(p.cpu(), opt_p.detach()) < 2e-6
    p2 = p0.clone().to(DEV)       # data-parallel mean folded into the update: g/world
    ops.adamw_step(p2, (2.0 * rnd(151, "g", (n,))).to(DEV), torch.zeros(n, device=DEV), torch.zeros(n, device=DEV), 1,
                   8e-4, grad_scale=0.5)
    p3 = p0.clone().to(DEV)
    ops.adamw_step(p3, rnd(151, "g", (n,)).to(DEV), torch.zeros(n, device=DEV), torch.zeros(n, device=DEV), 1, 8e-4)
    assert max_abs(p2.cpu(), p3.cpu()) < 1e-7


def test_fill_u8_hash_matches_host(ops):
    for shape in [(2, 3, 3, 8, 8), (1, 5, 3, 7, 9)]:
        dev = ops.fill_u8_hash(shape, 1000).cpu().numpy()
        assert np.array_equal(dev, synth.uint8_clip(1000, shape))


def test_errors_are_loud(ops):
    from tdeed_amd._lib import HipCallError
    with pytest.raises(HipCallError):
        ops.gemm(torch.zeros(4, 12, device=DEV), torch.zeros(8, 12, device=DEV))      # K % 8 != 0
    with pytest.raises(RuntimeError):
        ops.gemm(torch.zeros(8, 8), torch.zeros(8, 8))                                # CPU tensors
