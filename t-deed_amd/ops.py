"""torch-tensor front ends of the C-ABI entry points (one function per ``tdeed_*_fwd``).

Tensors must live on the GPU ("cuda" is the ROCm device string); activations are
channels-last (NHWC images, NTC sequences).  Nothing here computes on the host: each
function validates shapes, allocates the output with torch (device memory plumbing) and
launches the HIP kernel on the current stream.
"""
import torch

from . import _lib
from ._lib import call, ptr, stream_ptr, dtype_code, ACT_NONE, ACT_RELU, ACT_GELU  # noqa: F401


def _chk(t, name, dtype=None):
    if t is None:
        return
    if not t.is_cuda:
        raise RuntimeError(f"{name} must be a GPU tensor (tdeed_amd has no CPU path)")
    if not t.is_contiguous():
        raise RuntimeError(f"{name} must be contiguous")
    if dtype is not None and t.dtype != dtype:
        raise TypeError(f"{name}: expected {dtype}, got {t.dtype}")


def _flip_args(flip, N):
    """flip: bool (all frames) or a uint8/bool tensor of N per-frame flags -> (int flag, mask tensor | None)."""
    if isinstance(flip, torch.Tensor):
        m = flip.to(torch.uint8).contiguous()
        if m.numel() != N or not m.is_cuda:
            raise ValueError(f"flip mask must hold one flag per frame on the GPU ({m.numel()} vs {N})")
        return 0, m
    return int(bool(flip)), None


def stem(frames_u8, w, scale, shift, act_dtype, crop=None, flip=False, out=None, relu=True):
    """frames (N,3,H,W) uint8 (or fp32 holding 0..255: mixup batches) -> (N,Ho,Wo,32).  crop = (top,left,h,w) or None.
    flip: bool for all frames, or a (N,) uint8 tensor of per-frame flags."""
    _chk(frames_u8, "frames", torch.float32 if frames_u8.dtype == torch.float32 else torch.uint8)
    N, _, H, W = frames_u8.shape
    top, left, ch, cw = crop if crop is not None else (0, 0, H, W)
    Ho, Wo = (ch + 1) // 2, (cw + 1) // 2
    if out is None:
        out = torch.empty((N, Ho, Wo, 32), dtype=act_dtype, device=frames_u8.device)
    fl, fmask = _flip_args(flip, N)
    call("tdeed_stem_fwd", ptr(frames_u8), int(frames_u8.dtype == torch.float32), N, H, W, top, left, ch, cw, fl,
         ptr(fmask), ptr(w), ptr(scale), ptr(shift),
         ptr(out), int(relu), dtype_code(act_dtype), stream_ptr())
    return out


def stem_mfma_parts(ch, cw):
    return _lib.load().tdeed_stem_mfma_parts(ch, cw)


def stem_mfma(frames, wfrag, crop=None, flip=False):
    """Training stem on the MFMA pipe (bf16): frames (N,3,H,W) uint8 or fp32 0..255 -> (z (N,Ho,Wo,32) raw conv output,
    colpart (N*parts, 2, 32) fp32 per-workgroup column sums / sums of squares of z)."""
    _chk(frames, "frames", torch.float32 if frames.dtype == torch.float32 else torch.uint8)
    _chk(wfrag, "wfrag", torch.float32)
    N, _, H, W = frames.shape
    top, left, ch, cw = crop if crop is not None else (0, 0, H, W)
    parts = stem_mfma_parts(ch, cw)
    if parts <= 0:
        raise RuntimeError(f"stem_mfma: a {cw}-pixel row band does not fit LDS")
    z = torch.empty((N, (ch + 1) // 2, (cw + 1) // 2, 32), dtype=torch.bfloat16, device=frames.device)
    colpart = torch.empty((N * parts, 2, 32), dtype=torch.float32, device=frames.device)
    fl, fmask = _flip_args(flip, N)
    call("tdeed_stem_mfma_fwd", ptr(frames), int(frames.dtype == torch.float32), N, H, W, top, left, ch, cw, fl, ptr(fmask),
         ptr(wfrag), ptr(z), ptr(colpart), stream_ptr())
    return z, colpart


def s1_front_parts(ch, cw, C1):
    return _lib.load().tdeed_s1_front_parts(ch, cw, C1)


def s1_front(frames_u8, fw, crop=None, flip=False, y2=None, shortcut=None, pooled=None):
    """Fused pre-proc + stem + s1.b1.{conv1, conv2, downsample} (bf16).  fw: engine.pack_front_weights(...).
    frames (N,3,H,W) uint8 -> y2 (N,Ho,Wo,C1), shortcut (N,Ho,Wo,C1), pooled (N,parts,C1) fp32 sums."""
    _chk(frames_u8, "frames", torch.uint8)
    N, _, H, W = frames_u8.shape
    top, left, ch, cw = crop if crop is not None else (0, 0, H, W)
    Hs, Ws = (ch + 1) // 2, (cw + 1) // 2
    Ho, Wo = (Hs + 1) // 2, (Ws + 1) // 2
    C1 = fw.C1
    dev = frames_u8.device
    if y2 is None:
        y2 = torch.empty((N, Ho, Wo, C1), dtype=torch.bfloat16, device=dev)
    if shortcut is None:
        shortcut = torch.empty((N, Ho, Wo, C1), dtype=torch.bfloat16, device=dev)
    if pooled is None:
        pooled = torch.empty((N, s1_front_parts(ch, cw, C1), C1), dtype=torch.float32, device=dev)
    call("tdeed_s1_front_fwd", ptr(frames_u8), N, H, W, top, left, ch, cw, int(flip), ptr(fw.stem_wf), ptr(fw.stem_sc),
         ptr(fw.stem_sh), C1, ptr(fw.w1f), ptr(fw.sc1), ptr(fw.sh1), ptr(fw.wdf), ptr(fw.scd), ptr(fw.shd),
         ptr(fw.w2f), ptr(fw.sc2), ptr(fw.sh2), ptr(y2), ptr(shortcut), ptr(pooled), stream_ptr())
    return y2, shortcut, pooled


def gemm_colpart_rows(M):
    """rows of the per-tile column-statistics buffer of gemm(colpart=...): one per 128-row tile"""
    return (M + 127) // 128


def _out2(out2, A):
    if out2 is None:
        return None, 0, 0
    _chk(out2, "out2", A.dtype)
    return ptr(out2), out2.shape[-1], out2.shape[-1]


def gemm(A, W, scale=None, shift=None, act=ACT_NONE, residual=None, a_scale=None, a_scale_rows=0,
         A0=None, k0=0, gather=None, out=None, M=None, lda=None, ldc=None, colpart=None, out2=None, out2_pre=False):
    """C = act((A' @ W^T) * scale + shift + residual).  A (M,K) / W (N,K) same dtype.
    gather = (stride, hi, wi, ho, wo) for the stride-2 1x1 shortcut.  out2 (.., n2): also receives columns [0, n2);
    out2_pre: out2 gets them before residual / activation and C keeps only the residual there."""
    _chk(A, "A"); _chk(W, "W", A.dtype)
    K = W.shape[1]
    N = W.shape[0]
    if M is None:
        M = A.numel() // A.shape[-1]
        if gather is not None:
            s, hi, wi, ho, wo = gather
            M = (M // (hi * wi)) * ho * wo
    lda = A.shape[-1] if lda is None else lda
    if out is None:
        out = torch.empty((M, N), dtype=A.dtype, device=A.device)
    ldc = N if ldc is None else ldc
    g = gather if gather is not None else (1, 0, 0, 0, 0)
    call("tdeed_gemm_fwd", ptr(A), lda, ptr(A0), (A0.shape[-1] if A0 is not None else 0), k0,
         ptr(a_scale), a_scale_rows, M, K, N, ptr(W), W.shape[1], ptr(scale), ptr(shift),
         ptr(residual), (residual.shape[-1] if residual is not None else 0), act, ptr(out), ldc,
         g[0], g[1], g[2], g[3], g[4], ptr(colpart), *_out2(out2, A), int(bool(out2_pre)), dtype_code(A.dtype), stream_ptr())
    return out


def c1_gconv_fits(Hi, Wi, Cin, C, stride):
    return _lib.load().tdeed_c1_gconv_fits(Hi, Wi, Cin, C, stride) != 0


def c1_gconv_slab_tiles(Hi, Wi, C, stride):
    return _lib.load().tdeed_c1_gconv_slab_tiles(Hi, Wi, C, stride)


def c1_gconv(x, w1f, s1, h1, wfrag, scale, shift, gw, stride, C, G=None, out=None, pooled=None):
    """conv1 + grouped 3x3 of one bottleneck in one launch (tdeed_c1_gconv_fwd): x (N,Hi,Wi,Cin) bf16 -> y2 (N,Ho,Wo,C),
    pooled (N, parts, C) squeeze partial sums.  G (N*Hi*Wi, Fp): gate-shift output spliced into conv1's operand."""
    _chk(x, "x", torch.bfloat16); _chk(G, "G", torch.bfloat16)
    N, Hi, Wi, Cin = x.shape
    Ho, Wo = (Hi - 1) // stride + 1, (Wi - 1) // stride + 1
    if out is None:
        out = torch.empty((N, Ho, Wo, C), dtype=x.dtype, device=x.device)
    if pooled is None:
        pooled = torch.empty((N, gconv3x3_parts(Hi, Wi, C, stride, x.dtype), C), dtype=torch.float32, device=x.device)
    call("tdeed_c1_gconv_fwd", ptr(x), ptr(G), (G.shape[-1] if G is not None else 0), N, Hi, Wi, Cin, C, gw, stride, ptr(w1f),
         ptr(s1), ptr(h1), ptr(wfrag), ptr(scale), ptr(shift), ptr(out), ptr(pooled), stream_ptr())
    return out, pooled


def bneck_fits(h, w, C, R):
    return _lib.load().tdeed_bneck_fits(h, w, C, R) != 0


def bneck(x, w1f, s1, h1, w2f, s2, h2, se_w1f, se_b1, se_w2f, se_b2, R, w3f, s3, h3, G=None, out=None, out2=None,
          w2_tap_major=True):
    """Whole stride-1 bottleneck in one launch (tdeed_bneck_fwd): x (N,h,w,C) bf16 -> (N,h,w,C); G (N*h*w, Fp): gate-shift
    output spliced into conv1's operand; out2 (N*h*w, n2): compact copy of the first n2 output channels.
    w2f: engine.pack_gconv_frags(..., tap_major=w2_tap_major) -- tap-major k-slots are the conflict-free order of this launch
    (group width 8: bit-identical to the chain either way); False: the order tdeed_gconv3x3_fwd reads (16-wide groups)."""
    _chk(x, "x", torch.bfloat16); _chk(G, "G", torch.bfloat16); _chk(out2, "out2", torch.bfloat16)
    N, h, w, C = x.shape
    if out is None:
        out = torch.empty_like(x)
    call("tdeed_bneck_fwd", ptr(x), ptr(G), (G.shape[-1] if G is not None else 0), N, h, w, C, ptr(w1f), ptr(s1),
         ptr(h1), ptr(w2f), ptr(s2), ptr(h2), ptr(se_w1f), ptr(se_b1), ptr(se_w2f), ptr(se_b2), R, ptr(w3f), ptr(s3), ptr(h3),
         ptr(out), ptr(out2), (out2.shape[-1] if out2 is not None else 0), int(bool(w2_tap_major)), stream_ptr())
    return out


def bneck_gs(x, gx, gate, ysum, xsum, cw1, cb1, cw2, cb2, T, F, Fp, w1f, s1, h1, w2f, s2, h2, se_w1f, se_b1, se_w2f, se_b2, R,
             w3f, s3, h3, out=None, out2=None, w2_tap_major=True, qtail=None):
    """The one-launch bottleneck behind a gate-shift-fuse site with the site's blend inside its frame load
    (tdeed_bneck_gs_fwd): gx (N,h,w,ldx >= Fp) the slice's source, gate / ysum / xsum from gate_shift(gates_only=True).
    == bneck(x, G=gate_shift(gx, ..., src_order=True)), bit for bit.
    qtail = (wpf, bn, F, Q): also the tap maps Q (N,h,w,6) of the NEXT block's site (fold F, wpf = engine.pack_gsf_p_frags,
    bn = gsq_bn_table) -- what gate_shift's first launch computes from `out`, in another fp32 summation order; that site then
    runs gate_shift(q_given=True)."""
    _chk(x, "x", torch.bfloat16); _chk(gx, "gx", torch.bfloat16); _chk(out2, "out2", torch.bfloat16)
    N, h, w, C = x.shape
    if out is None:
        out = torch.empty_like(x)
    call("tdeed_bneck_gs_fwd", ptr(x), ptr(gx), gx.shape[-1], ptr(gate), ptr(ysum), ptr(xsum), ptr(cw1), ptr(cb1), ptr(cw2),
         ptr(cb2), T, F, Fp, N, h, w, C, ptr(w1f), ptr(s1), ptr(h1), ptr(w2f), ptr(s2), ptr(h2), ptr(se_w1f), ptr(se_b1),
         ptr(se_w2f), ptr(se_b2), R, ptr(w3f), ptr(s3), ptr(h3), ptr(out), ptr(out2),
         (out2.shape[-1] if out2 is not None else 0), int(bool(w2_tap_major)),
         *((ptr(qtail[0]), ptr(qtail[1]), int(qtail[2]), ptr(qtail[3])) if qtail is not None else (None, None, 0, None)),
         stream_ptr())
    return out


def bneck_qtail_fits(h, w, C, F):
    return _lib.load().tdeed_bneck_qtail_fits(h, w, C, F) != 0


def gsq_bn_table(bn_scale, bn_shift):
    """[2][8 * ceil(F / 8)] fp32: the site's folded BatchNorm3d scale | shift, zeros behind channel F (tdeed_bneck_gs_fwd's q_bn)."""
    F = bn_scale.numel()
    t = torch.zeros((2, (F + 7) // 8 * 8), dtype=torch.float32, device=bn_scale.device)
    t[0, :F] = bn_scale
    t[1, :F] = bn_shift
    return t


def gemm_ws_fits_mode(K, N, act_dtype):
    """0: no; 1: weights fit LDS (preferred kernel for narrow layers); 2: weights streamed from L2 (wide layers)."""
    return _lib.load().tdeed_gemm_ws_fits(K, N, dtype_code(act_dtype))


def gemm_ws_fits(K, N, act_dtype):
    return gemm_ws_fits_mode(K, N, act_dtype) != 0


def gemm_ws(A, Wfrag, K, N, scale=None, shift=None, act=ACT_NONE, residual=None, a_scale=None, a_scale_rows=0,
            A0=None, k0=0, gather=None, out=None, M=None, lda=None, ldc=None, out2=None):
    """Weight-stationary streaming form of gemm(); Wfrag from engine.pack_ws_weights."""
    _chk(A, "A")
    if M is None:
        M = A.numel() // A.shape[-1]
        if gather is not None:
            s, hi, wi, ho, wo = gather
            M = (M // (hi * wi)) * ho * wo
    lda = A.shape[-1] if lda is None else lda
    if out is None:
        out = torch.empty((M, N), dtype=A.dtype, device=A.device)
    ldc = N if ldc is None else ldc
    g = gather if gather is not None else (1, 0, 0, 0, 0)
    call("tdeed_gemm_ws_fwd", ptr(A), lda, ptr(A0), (A0.shape[-1] if A0 is not None else 0), k0,
         ptr(a_scale), a_scale_rows, M, K, N, ptr(Wfrag), ptr(scale), ptr(shift),
         ptr(residual), (residual.shape[-1] if residual is not None else 0), act, ptr(out), ldc,
         g[0], g[1], g[2], g[3], g[4], *_out2(out2, A), dtype_code(A.dtype), stream_ptr())
    return out


def gemm_rs_fits(M, K, N):
    return _lib.load().tdeed_gemm_rs_fits(M, K, N) != 0


def gemm_rs(A, Wfrag, K, N, scale=None, shift=None, act=ACT_NONE, residual=None, a_scale=None, a_scale_rows=0, A0=None, k0=0,
            out=None, M=None, out2=None):
    """Register-stationary form of gemm_ws() for K = N = 320 (tdeed_gemm_rs_fwd); Wfrag from engine.pack_ws_weights."""
    _chk(A, "A", torch.bfloat16)
    if M is None:
        M = A.numel() // A.shape[-1]
    if out is None:
        out = torch.empty((M, N), dtype=A.dtype, device=A.device)
    call("tdeed_gemm_rs_fwd", ptr(A), A.shape[-1], ptr(A0), (A0.shape[-1] if A0 is not None else 0), k0, ptr(a_scale),
         a_scale_rows, M, K, N, ptr(Wfrag), ptr(scale), ptr(shift), ptr(residual),
         (residual.shape[-1] if residual is not None else 0), act, ptr(out), N, *_out2(out2, A), stream_ptr())
    return out


def gemm_rs_stats(A, Wfrag, K, N, M=None, A0=None, k0=0):
    """Training forward of a K = N = 320 layer on the register-stationary kernel: raw output + the (sums, sums of squares,
    row stride, rows) partials of its BatchNorm statistics (tdeed_gemm_rs_stats_fwd), like gemm(colpart=...)."""
    _chk(A, "A", torch.bfloat16)
    if M is None:
        M = A.numel() // A.shape[-1]
    out = torch.empty((M, N), dtype=A.dtype, device=A.device)
    P = _lib.load().tdeed_gemm_rs_grid(M)
    cp = torch.empty((P, 2, N), dtype=torch.float32, device=A.device)
    call("tdeed_gemm_rs_stats_fwd", ptr(A), A.shape[-1], ptr(A0), (A0.shape[-1] if A0 is not None else 0), k0, M, K, N,
         ptr(Wfrag), ptr(out), N, ptr(cp), stream_ptr())
    flat = cp.view(-1)
    return out, (flat, flat[N:], 2 * N, P)


def gemm_splitk_splits(K):
    return _lib.load().tdeed_gemm_splitk_splits(K)


def gemm_splitk_workspace(M, K, N, device):
    """fp32 scratch for gemm_splitk: one [M][N] partial per K chunk."""
    return torch.empty((_lib.load().tdeed_gemm_splitk_splits(K), M, N), dtype=torch.float32, device=device)


def gemm_splitk(A, W, scale, shift, act=ACT_NONE, residual=None, out=None, M=None, workspace=None):
    """Split-K contraction for short sequences (bf16): act((A . W^T) * scale + shift + residual)."""
    _chk(A, "A", torch.bfloat16)
    _chk(W, "W", torch.bfloat16)
    N, K = W.shape
    if M is None:
        M = A.numel() // K
    if out is None:
        out = torch.empty(tuple(A.shape[:-1]) + (N,), dtype=A.dtype, device=A.device)
    if workspace is None:
        workspace = gemm_splitk_workspace(M, K, N, A.device)
    call("tdeed_gemm_splitk_fwd", ptr(A), K, M, K, N, ptr(W), K, ptr(scale), ptr(shift), ptr(residual), N, act,
         ptr(out), N, ptr(workspace), stream_ptr())
    return out


def se_gate_mfma_fits(C, R):
    return _lib.load().tdeed_se_gate_mfma_fits(C, R) != 0


def se_gate_mfma(pooled, inv_cnt, w1f, b1, w2f, b2, R, out=None):
    """SE excitation on the MFMA pipe; pooled (N, parts, C) fp32 partial sums -> gate (N, C) fp32."""
    N, parts, C = pooled.shape
    if out is None:
        out = torch.empty((N, C), dtype=torch.float32, device=pooled.device)
    call("tdeed_se_gate_mfma_fwd", ptr(pooled), parts, float(inv_cnt), N, C, R, ptr(w1f), ptr(b1), ptr(w2f), ptr(b2),
         ptr(out), stream_ptr())
    return out


def se_gate_bf16(pooled, inv_cnt, w1p, b1, w2p, b2, R, out=None):
    """SE excitation with bf16 packed weights (engine.pack_se_bf16): pooled (N,parts,C) sums -> gate (N,C)."""
    N, parts, C = pooled.shape
    if out is None:
        out = torch.empty((N, C), dtype=torch.float32, device=pooled.device)
    call("tdeed_se_gate_bf16_fwd", ptr(pooled), parts, float(inv_cnt), N, C, R, ptr(w1p), ptr(b1), ptr(w2p), ptr(b2),
         ptr(out), stream_ptr())
    return out


def gconv3x3_mfma_fits(Hi, Wi, C, stride):
    return _lib.load().tdeed_gconv3x3_mfma_fits(Hi, Wi, C, stride) != 0


def gconv3x3_parts(Hi, Wi, C, stride, act_dtype):
    return _lib.load().tdeed_gconv3x3_parts(Hi, Wi, C, stride, dtype_code(act_dtype))


def gconv3x3(x, w_packed, scale, shift, gw, stride, wfrag=None, out=None, pooled=None, relu=True, pooled_sq=None,
             in_affine=None):
    """x (N,Hi,Wi,C) -> y (N,Ho,Wo,C), pooled (N,parts,C) fp32 partial sums over pixels.
    w_packed: fp32 [G][9][gw][gw] (VALU path); wfrag: bf16 MFMA fragments (bf16 path).
    in_affine = (a, b) fp32 [C] (bf16 MFMA path): x is a raw conv output, relu(a*x + b) is applied while it is staged."""
    _chk(x, "x")
    N, Hi, Wi, C = x.shape
    Ho, Wo = (Hi - 1) // stride + 1, (Wi - 1) // stride + 1
    if out is None:
        out = torch.empty((N, Ho, Wo, C), dtype=x.dtype, device=x.device)
    parts = gconv3x3_parts(Hi, Wi, C, stride, x.dtype) if wfrag is not None else 1
    if pooled is None:
        pooled = torch.empty((N, parts, C), dtype=torch.float32, device=x.device)
    call("tdeed_gconv3x3_fwd", ptr(x), N, Hi, Wi, C, gw, stride, ptr(w_packed), ptr(wfrag), ptr(scale), ptr(shift),
         ptr(out), ptr(pooled), ptr(pooled_sq), ptr(in_affine[0] if in_affine else None),
         ptr(in_affine[1] if in_affine else None), int(relu), dtype_code(x.dtype), stream_ptr())
    return out, pooled


def se_gate(pooled, inv_cnt, w1t, b1, w2t, b2, out=None):
    """pooled (N,parts,C) partial sums -> gate (N,C).  w1t (C,R), w2t (R,C)."""
    N, parts, C = pooled.shape
    R = w1t.shape[1]
    if out is None:
        out = torch.empty((N, C), dtype=torch.float32, device=pooled.device)
    call("tdeed_se_gate_fwd", ptr(pooled), parts, float(inv_cnt), N, C, R, ptr(w1t), ptr(b1), ptr(w2t), ptr(b2),
         ptr(out), stream_ptr())
    return out


def gate_shift(x, B, T, F, Fp, bn_scale, bn_shift, wq, b3d, cw1=None, cb1=None, cw2=None, cb2=None,
               bufs=None, wqf=None, separate_weight=False, src_order=False, gates_only=False, q_given=False):
    """x (B*T,h,w,C) -> (B*T*h*w, Fp): gated/shifted/fused first F channels (+ pad copy).
    GSM when cw1 is None.  bufs: optional dict of preallocated gate/ysum/xsum/fw/out.
    src_order (GSF, bf16): the output stays in source channel order, out[:, gs_source_order(F)] is the module's output.
    q_given: bufs["q"] already holds the site's tap maps (bneck_gs's qtail): the first launch is skipped.
    gates_only: stop behind the gate launches and return (gate, ysum, xsum) -- the caller's next launch does the blend (bneck_gs)."""
    _chk(x, "x")
    N, h, w, C = x.shape
    dev = x.device
    bufs = bufs or {}
    gate = bufs.get("gate")
    if gate is None:
        gate = torch.empty((N, h, w, 2), dtype=torch.float32, device=dev)
    ysum = bufs.get("ysum")
    if ysum is None:
        ysum = torch.empty((N, F), dtype=torch.float32, device=dev)
    xsum = bufs.get("xsum")
    if xsum is None:
        xsum = torch.empty((N, F), dtype=torch.float32, device=dev)
    out = bufs.get("out")
    if out is None and not gates_only:
        out = torch.empty((N * h * w, Fp), dtype=x.dtype, device=dev)
    q = bufs.get("q")
    if q is None:
        q = torch.empty((N, h, w, 6), dtype=torch.float32, device=dev)
    dc = dtype_code(x.dtype)
    if q_given:
        if x.dtype != torch.bfloat16 or bufs.get("q") is None:
            raise ValueError("gate_shift: q_given needs bufs['q'] and bf16")
        call("tdeed_gsf_gate_sums_fwd", ptr(x), B, T, h, w, C, F, ptr(b3d), ptr(q), ptr(gate), ptr(ysum), ptr(xsum), stream_ptr())
    else:
        call("tdeed_gsf_gate_fwd", ptr(x), B, T, h, w, C, F, ptr(bn_scale), ptr(bn_shift), ptr(wq), ptr(wqf), ptr(b3d),
             ptr(q), ptr(gate), ptr(ysum), ptr(xsum), dc, stream_ptr())
    if gates_only:
        return gate, ysum, xsum
    if src_order:
        if cw1 is None or separate_weight or x.dtype != torch.bfloat16:
            raise ValueError("gate_shift: src_order is the fused bf16 GSF launch only")
        call("tdeed_gsf_blend_src_fwd", ptr(x), ptr(gate), ptr(ysum), ptr(xsum), ptr(cw1), ptr(cb1), ptr(cw2), ptr(cb2),
             B, T, h, w, C, F, Fp, ptr(out), dc, stream_ptr())
        return out
    if cw1 is not None and not separate_weight:
        call("tdeed_gsf_apply_fused_fwd", ptr(x), ptr(gate), ptr(ysum), ptr(xsum), ptr(cw1), ptr(cb1), ptr(cw2), ptr(cb2),
             B, T, h, w, C, F, Fp, ptr(out), dc, stream_ptr())
        return out
    fw = None
    if cw1 is not None:
        fw = bufs.get("fw")
        if fw is None:
            fw = torch.empty((B, F, T), dtype=torch.float32, device=dev)
        call("tdeed_gsf_weight_fwd", ptr(ysum), ptr(xsum), B, T, F, h * w, ptr(cw1), ptr(cb1), ptr(cw2), ptr(cb2),
             ptr(fw), stream_ptr())
    call("tdeed_gsf_apply_fwd", ptr(x), ptr(gate), ptr(fw), B, T, h, w, C, F, Fp, ptr(out), dc, stream_ptr())
    return out


def gs_source_order(F):
    """Source channel of every output channel of the gate-shift module's interleave (impl/gsf.py:88-91: inside each half,
    c = i*(F/4)+j -> 2j+i): module_out[:, co] = src_order_out[:, gs_source_order(F)[co]]."""
    Fh, Fq = F // 2, F // 4
    return [g * Fh + (col & 1) * Fq + (col >> 1) for g in range(2) for col in range(Fh)]


def avgpool_posenc(x, B, T, temp_enc, out=None, rowstat=None):
    """rowstat: optional fp32 (B*T, 2) output, LayerNorm mean / rstd over C of every feature row."""
    N, h, w, C = x.shape
    if out is None:
        out = torch.empty((B, T, C), dtype=x.dtype, device=x.device)
    call("tdeed_avgpool_posenc_fwd", ptr(x), B, T, h * w, C, ptr(temp_enc), ptr(out), ptr(rowstat), dtype_code(x.dtype),
         dtype_code(out.dtype), stream_ptr())
    return out


def layernorm(x, w, b, eps=1e-5, out=None, ldy=None, rows=None, C=None, ldx=None):
    C = x.shape[-1] if C is None else C
    rows = x.numel() // x.shape[-1] if rows is None else rows
    if out is None:
        out = torch.empty_like(x)
    call("tdeed_layernorm_fwd", ptr(x), (C if ldx is None else ldx), rows, C, ptr(w), ptr(b), eps, ptr(out),
         (C if ldy is None else ldy), dtype_code(x.dtype), stream_ptr())
    return out


def sgp_branch(o, x, ks, up, dw, db, out=None):
    B, T, C = x.shape
    if out is None:
        out = torch.empty_like(x)
    call("tdeed_sgp_branch_fwd", ptr(o), ptr(x), B, T, C, ks, up, ptr(dw), ptr(db), ptr(out), dtype_code(x.dtype),
         stream_ptr())
    return out


def mixer_branch(xn, cat, T_hi, ks, up, dw1, db1, dw2, db2):
    B, T_lo, C = xn.shape
    call("tdeed_mixer_branch_fwd", ptr(xn), B, T_hi, T_lo, C, ks, up, ptr(dw1), ptr(db1), ptr(dw2), ptr(db2),
         ptr(cat), dtype_code(xn.dtype), stream_ptr())
    return cat


def _rs_parts(rowstat):
    """rowstat (rows, 2) = (mean, rstd) -> 0; (n, rows, 2) = partial (sum, sum of squares) per column tile -> n"""
    return 0 if rowstat is None or rowstat.dim() == 2 else int(rowstat.shape[0])


def sgp_front(x, ks, up, ln_w, ln_b, dw, db, eps=1e-5, out=None, chsum=None, rowstat=None, out16=None):
    """SGPBlock front half with the LayerNorm computed in-kernel: y = x + LN(x) + fc*phi + (convw+convkw)*psi.
    chsum: optional fp32 (B, C, 2) output, per-channel sum / sum of squares over T of y (for sgp_gemm_gn_gelu's GroupNorm);
    rowstat: optional fp32 (B*T, 2) input, LayerNorm statistics of every row of x: (mean, rstd) as avgpool_posenc leaves them, or (n, B*T, 2) partial (sum, sum of
    squares) per column tile as sgp_gemm_residual does."""
    B, T, C = x.shape
    if out is None:
        out = torch.empty_like(x)
    call("tdeed_sgp_front_fwd", ptr(x), B, T, C, ks, up, ptr(ln_w), ptr(ln_b), eps, ptr(dw), ptr(db), ptr(out),
         ptr(chsum), ptr(rowstat), _rs_parts(rowstat), ptr(out16), dtype_code(x.dtype), stream_ptr())
    return out


def mixer_front(z, xlo, cat, ks, up, ln1_w, ln1_b, ln2_w, ln2_b, dw1, db1, dw2, db2, eps=1e-5, rowstat_z=None,
                rowstat_x=None):
    """SGPMixer front half with both LayerNorms in-kernel: fills all six slabs of cat (B, T_hi, 6C)."""
    B, T_hi, C = z.shape
    T_lo = xlo.shape[1]
    call("tdeed_mixer_front_fwd", ptr(z), ptr(xlo), B, T_hi, T_lo, C, ks, up, ptr(ln1_w), ptr(ln1_b), ptr(ln2_w),
         ptr(ln2_b), eps, ptr(dw1), ptr(db1), ptr(dw2), ptr(db2), ptr(cat), ptr(rowstat_z), _rs_parts(rowstat_z),
         ptr(rowstat_x), _rs_parts(rowstat_x), dtype_code(z.dtype), dtype_code(cat.dtype), stream_ptr())
    return cat


def groupnorm(x, G, w, b, eps=1e-5, out=None):
    B, T, C = x.shape
    if out is None:
        out = torch.empty_like(x)
    call("tdeed_groupnorm_fwd", ptr(x), B, T, C, G, ptr(w), ptr(b), eps, ptr(out), dtype_code(x.dtype), stream_ptr())
    return out


def maxpool(x, T_out, out=None):
    B, T_in, C = x.shape
    if out is None:
        out = torch.empty((B, T_out, C), dtype=x.dtype, device=x.device)
    call("tdeed_maxpool_fwd", ptr(x), B, T_in, T_out, C, ptr(out), dtype_code(x.dtype), stream_ptr())
    return out


def maxpool_rowstat(x, T_out, out=None, rowstat=None, eps=1e-5):
    """AdaptiveMaxPool1d(T_out) along T of (B,T,C) + the LayerNorm (mean, rstd) of every pooled row in rowstat (B*T_out, 2)"""
    B, T_in, C = x.shape
    if out is None:
        out = torch.empty((B, T_out, C), dtype=x.dtype, device=x.device)
    if rowstat is None:
        rowstat = torch.empty((B * T_out, 2), dtype=torch.float32, device=x.device)
    call("tdeed_maxpool_rowstat_fwd", ptr(x), B, T_in, T_out, C, ptr(out), ptr(rowstat), eps, dtype_code(x.dtype), stream_ptr())
    return out, rowstat


def heads(x, w, b, out=None):
    rows = x.numel() // x.shape[-1]
    C = x.shape[-1]
    n_out = w.shape[0]
    if out is None:
        out = torch.empty((rows, n_out), dtype=torch.float32, device=x.device)
    call("tdeed_heads_fwd", ptr(x), rows, C, ptr(w), ptr(b), n_out, ptr(out), dtype_code(x.dtype), stream_ptr())
    return out


def _chk_labels(head_out, hard, soft, labelD, K):
    """Host-side argument checks of the loss entry points (dtype / contiguity / shape; no device sync).  Label VALUES are
    checked by the kernels: a label outside [0, K) is never used as an index and turns the loss into NaN."""
    _chk(head_out, "head_out", torch.float32)
    rows = head_out.shape[0]
    if hard is not None:
        _chk(hard, "labels", torch.int64)
        if hard.numel() != rows:
            raise ValueError(f"labels: {hard.numel()} entries for {rows} rows")
    if soft is not None:
        _chk(soft, "soft labels", torch.float32)
        if soft.numel() != rows * K:
            raise ValueError(f"soft labels: {tuple(soft.shape)} for {rows} rows x {K} classes")
    if labelD is not None:
        _chk(labelD, "labelD", torch.float32)
        if labelD.numel() != rows:
            raise ValueError(f"labelD: {labelD.numel()} entries for {rows} rows")
    if hard is None and soft is None:
        raise ValueError("loss needs hard or soft labels")


def loss(head_out, K1, cls_w, hard=None, soft=None, displ_col=-1, labelD=None, out=None):
    rows, ld = head_out.shape
    _chk_labels(head_out, hard, soft, labelD, K1)
    _chk(cls_w, "cls_w", torch.float32)
    if out is None:
        out = torch.empty(3, dtype=torch.float32, device=head_out.device)
    call("tdeed_loss_fwd", ptr(head_out), rows, ld, K1, ptr(hard), ptr(soft), ptr(cls_w), displ_col, ptr(labelD),
         ptr(out), stream_ptr())
    return out


def loss_bwd(head_out, K1, cls_w, hard=None, soft=None, displ_col=-1, labelD=None, grad_scale=1.0):
    """d(CE+MSE)/d(head_out) (rows, ld) fp32."""
    rows, ld = head_out.shape
    _chk_labels(head_out, hard, soft, labelD, K1)
    dhead = torch.empty_like(head_out)
    call("tdeed_loss_bwd", ptr(head_out), rows, ld, K1, ptr(hard), ptr(soft), ptr(cls_w), displ_col, ptr(labelD),
         float(grad_scale), ptr(dhead), stream_ptr())
    return dhead


def loss2(head_out, B, T, K1a, K1b, dataset, hard, cls_w, displ_col=-1, labelD=None, want_grad=False, grad_scale=1.0,
          soft=None):
    """Double-head (joint dataset) loss of model.py:278-306 -> (out[3], dhead | None).  hard: int64 labels over the
    concatenated heads, or soft: (B*T, K1a+K1b) fp32 label distributions (mixup)."""
    ld = head_out.shape[-1]
    _chk_labels(head_out, hard, soft, labelD, K1a + K1b)
    _chk(dataset, "dataset", torch.int64)
    if dataset.numel() != B or cls_w.numel() < max(K1a, K1b):
        raise ValueError("loss2: dataset needs one id per clip, cls_w max(K1a, K1b) weights")
    out = torch.empty(3, dtype=torch.float32, device=head_out.device)
    dhead = torch.empty_like(head_out) if want_grad else None
    call("tdeed_loss2", ptr(head_out), B, T, ld, K1a, K1b, ptr(dataset), ptr(hard), ptr(soft), ptr(cls_w), displ_col,
         ptr(labelD), float(grad_scale), ptr(out), ptr(dhead), stream_ptr())
    return out, dhead


def loss2_soft(head_out, B, T, K1a, K1b, dataset, soft, cls_w, displ_col=-1, labelD=None, grad_scale=1.0):
    return loss2(head_out, B, T, K1a, K1b, dataset, None, cls_w, displ_col, labelD, True, grad_scale, soft=soft)


def heads_bwd(dout, x, w, need_dx=True):
    """FCLayers backward: returns (dx | None, dw (n_out,C) fp32, db (n_out,) fp32)."""
    rows, n_out = dout.shape
    C = x.shape[-1]
    ws = torch.empty(_lib.load().tdeed_heads_bwd_workspace(rows, C, n_out), dtype=torch.uint8, device=x.device)
    dx = torch.empty_like(x) if need_dx else None
    dw = torch.empty((n_out, C), dtype=torch.float32, device=x.device)
    db = torch.empty((n_out,), dtype=torch.float32, device=x.device)
    call("tdeed_heads_bwd", ptr(dout), ptr(x), rows, C, ptr(w), n_out, ptr(dx), ptr(dw), ptr(db), ptr(ws),
         dtype_code(x.dtype), stream_ptr())
    return dx, dw, db


def adamw_step(param, grad, exp_avg, exp_avg_sq, step, lr, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.01,
               grad_scale=1.0):
    """In-place fused AdamW on flat fp32 buffers (torch.optim.AdamW semantics)."""
    for t_ in (param, grad, exp_avg, exp_avg_sq):
        _chk(t_, "adamw buffer", torch.float32)
    call("tdeed_adamw_step", ptr(param), ptr(grad), ptr(exp_avg), ptr(exp_avg_sq), param.numel(), float(lr),
         float(betas[0]), float(betas[1]), float(eps), float(weight_decay), int(step), float(grad_scale), stream_ptr())
    return param


def process_prediction(head_out, B, T, K1, displ_col):
    ld = head_out.shape[-1]
    scores = torch.empty((B, T, K1), dtype=torch.float32, device=head_out.device)
    cls = torch.empty((B, T), dtype=torch.int64, device=head_out.device)
    call("tdeed_process_prediction", ptr(head_out), B, T, ld, K1, displ_col, ptr(scores), ptr(cls), stream_ptr())
    return cls, scores


def cast_bf16(src_f32):
    out = torch.empty(src_f32.shape, dtype=torch.bfloat16, device=src_f32.device)
    call("tdeed_cast_f32_to_bf16", ptr(src_f32), ptr(out), src_f32.numel(), stream_ptr())
    return out


def fill_u8_hash(shape, seed, device="cuda"):
    """Device-side twin of tdeed_amd.synth.uint8_clip (bit-identical bytes)."""
    from .synth import fnv1a64
    n = 1
    for s in shape:
        n *= int(s)
    out = torch.empty(((n + 7) // 8) * 8, dtype=torch.uint8, device=device)
    base = ((seed * 0x9E3779B97F4A7C15) ^ fnv1a64("clip")) & 0xFFFFFFFFFFFFFFFF
    call("tdeed_fill_u8_hash", ptr(out), n, base, stream_ptr())
    return out[:n].view(*shape)


# ---------------------------------------------------------------------------------------------- SGP contractions (sgp_gemm.hip)
def sgp_gemm_ksteps(K):
    return int(_lib.load().tdeed_sgp_gemm_ksteps(K))


def sgp_gemm_form(mode, B, T, N, K):
    """(MT, NT) of the tile the launcher wants for this contraction: 16 MT rows of one clip x 64 NT features per workgroup.
    mode 0 / 3: GroupNorm + fc1 + GELU on bf16 / fp32 rows, 1: fc2 + residual, 2: concat_fc + GELU"""
    f = int(_lib.load().tdeed_sgp_gemm_form(mode, B, T, N, K))
    return f >> 4, f & 15


def sgp_gemm_tiles(T, N, form):
    """(row tiles per clip, column tiles) of a form: the leading dimensions of chs_out / rowstat_part"""
    L = _lib.load()
    return int(L.tdeed_sgp_gemm_row_tiles(T, form[0])), int(L.tdeed_sgp_gemm_col_tiles(N, form[1]))


def sgp_gemm_gn_gelu(y, chsum, gn_w, gn_b, Wp, bias, N, out=None, form=None, G=16, eps=1e-5):
    """H (B,T,N) bf16 = GELU(GroupNorm(y) @ W^T + b).  y (B,T,K) bf16 | fp32; chsum fp32 (parts,B,K,2) or (B,K,2): per-channel
    (sum, sum of squares) of y over each clip's rows; Wp: engine.pack_mfma_frags(W, ks_mult=12)."""
    B, T, K = y.shape
    parts = 1 if chsum.dim() == 3 else chsum.shape[0]
    form = form or sgp_gemm_form(0 if y.dtype == torch.bfloat16 else 3, B, T, N, K)
    if out is None:
        out = torch.empty((B, T, N), dtype=torch.bfloat16, device=y.device)
    call("tdeed_sgp_gemm_gn_gelu", ptr(y), B, T, K, ptr(chsum), parts, ptr(gn_w), ptr(gn_b), G, eps, ptr(Wp), ptr(bias), N,
         ptr(out), form[0] * 16 + form[1], dtype_code(y.dtype), stream_ptr())
    return out


def sgp_gemm_residual(H, Wp, bias, resid, out=None, rowstat_part=None, pooled=None, rowstat_pool_part=None, form=None):
    """out (B,T,N) = resid + H @ W^T + b (H bf16 (B,T,K); out / resid / pooled bf16 | fp32).  rowstat_part fp32 (nct, B*T, 2):
    (sum, sum of squares) of every stored row over each column tile; pooled (B,T/2,N): AdaptiveMaxPool1d(T/2) of out and its
    rowstat_pool_part (nct, B*T/2, 2).  -> (out, form)"""
    B, T, K = H.shape
    N = resid.shape[-1]
    form = form or sgp_gemm_form(1, B, T, N, K)
    if out is None:
        out = torch.empty_like(resid)
    call("tdeed_sgp_gemm_residual", ptr(H), B, T, K, ptr(Wp), ptr(bias), N, ptr(resid), ptr(out), ptr(rowstat_part),
         ptr(pooled), ptr(rowstat_pool_part), 0 if pooled is None else pooled.shape[1], form[0] * 16 + form[1],
         dtype_code(out.dtype), stream_ptr())
    return out


def sgp_gemm_gelu_chsum(A, Wp, bias, N, out, chs_out, form=None, out16=None):
    """out (B,T,N) bf16 | fp32 = GELU(A @ W^T + b), A bf16 (B,T,K); chs_out fp32 (NJ,B,N,2): per-channel (sum, sum of squares)
    of the stored rows per row tile"""
    B, T, K = A.shape
    form = form or sgp_gemm_form(2, B, T, N, K)
    call("tdeed_sgp_gemm_gelu_chsum", ptr(A), B, T, K, ptr(Wp), ptr(bias), N, ptr(out), ptr(chs_out), ptr(out16),
         form[0] * 16 + form[1], dtype_code(out.dtype), stream_ptr())
    return out
