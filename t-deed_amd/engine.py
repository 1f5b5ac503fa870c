"""Launch plan of the T-DEED inference forward on one MI355X.

``ForwardEngine`` turns a reference-format state_dict into packed device weights (BN folded
into per-channel scale/shift epilogues, depthwise weights packed per channel, dense weights in
the activation dtype) and, per input geometry, a static list of kernel launches over
pre-allocated buffers.  The list is captured once into a HIP graph and replayed: no tracing
compiler, no per-op host work in the steady state.

Mirrors ``TDEEDModel.Impl.forward(x, inference=True)`` (/root/reference/model/model.py:105-149).
"""
import math
import os
from types import SimpleNamespace

import numpy as np
import torch

from . import ops, _lib
from .streams import new_stream
from .regnet_spec import regnet_spec, sgp_up_size, pyramid_lengths

BN_EPS = 1e-5


def _np(v):
    if isinstance(v, torch.Tensor):
        return v.detach().cpu().numpy()
    return np.asarray(v)


class _Pool:
    """Free-list of device byte buffers so that dead activations are recycled (keeps the working
    set small enough to live in the 256 MiB Infinity Cache between producer and consumer)."""

    def __init__(self, device):
        self.device = device
        self.free = []
        self.all = []

    def take(self, shape, dtype):
        n = int(np.prod(shape)) * torch.empty((), dtype=dtype).element_size()
        n = (n + 255) // 256 * 256
        best = None
        for i, b in enumerate(self.free):
            if b.numel() >= n and (best is None or b.numel() < self.free[best].numel()):
                best = i
        if best is not None and self.free[best].numel() <= 2 * n + (1 << 20):
            raw = self.free.pop(best)
        else:
            raw = torch.empty(n, dtype=torch.uint8, device=self.device)
            self.all.append(raw)
        t = raw[:int(np.prod(shape)) * torch.empty((), dtype=dtype).element_size()].view(dtype).view(*shape)
        t._td_raw = raw
        return t

    def give(self, t):
        self.free.append(t._td_raw)

    def total_bytes(self):
        return sum(b.numel() for b in self.all)


def _f32(a, device):
    return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(device)


def _dense(a, act_dtype, device):
    return _f32(a, device).to(act_dtype).contiguous()


def _dwpack(sd, pre, names, C, device):
    """per-channel depthwise weights [C][sum of taps] and biases [len(names)][C]"""
    ws = [_np(sd[f"{pre}.{n}.weight"]).reshape(C, -1) for n in names]
    bs = [_np(sd[f"{pre}.{n}.bias"]).reshape(C) for n in names]
    return _f32(np.concatenate(ws, axis=1), device), _f32(np.stack(bs, axis=0), device)


SGP_GEMM = os.environ.get("TDEED_SGP_GEMM", "1") == "1"             # the SGP contractions on sgp_gemm.hip (round 5)
# its residual stream (block / mixer inputs and outputs, the stash) in fp32 under a bf16 trunk, like the reference's autocast
SGP_F32_STREAM = os.environ.get("TDEED_SGP_F32_STREAM", "1") == "1"
# feature dimensions above this get a bf16 copy of the fc1 operand beside the fp32 rows (a wide fc1 launch is bound by the
# bytes per CU; narrow ones are latency bound and take the fp32 rows directly)
SGP_BF16_OPERAND_MIN_C = int(os.environ.get("TDEED_SGP_BF16_OPERAND_MIN_C", "384"))


def sgp_stream_dtype(act_dtype, device):
    """type of the temporal stage's residual stream (feat, block / mixer outputs, the stash) for a trunk in act_dtype"""
    if SGP_GEMM and SGP_F32_STREAM and act_dtype == torch.bfloat16 and str(device) != "cpu":
        return torch.float32
    return act_dtype


def _pack_mlp(sd, pre, C, o, act_dtype, device):
    o.w_fc1 = _dense(_np(sd[pre + ".mlp.0.weight"]).reshape(4 * C, C), act_dtype, device)
    o.b_fc1 = _f32(_np(sd[pre + ".mlp.0.bias"]), device)
    o.w_fc2 = _dense(_np(sd[pre + ".mlp.2.weight"]).reshape(C, 4 * C), act_dtype, device)
    o.b_fc2 = _f32(_np(sd[pre + ".mlp.2.bias"]), device)
    o.gn_w, o.gn_b = _f32(_np(sd[pre + ".gn.weight"]), device), _f32(_np(sd[pre + ".gn.bias"]), device)
    o.w1g = o.w2g = None
    if act_dtype == torch.bfloat16 and str(device) != "cpu" and C % 16 == 0 and SGP_GEMM:
        # sgp_gemm.hip: plain MFMA fragments of the whole weight, k-steps padded to its chunk ring
        o.w1g = pack_mfma_frags(_np(sd[pre + ".mlp.0.weight"]).reshape(4 * C, C), device, ks_mult=12)
        o.w2g = pack_mfma_frags(_np(sd[pre + ".mlp.2.weight"]).reshape(C, 4 * C), device, ks_mult=12)


def pack_ws_weights(W, act_dtype, device):
    """[N][K] dense weight -> fragments for gemm_ws_kernel: [NT][KS][64][epc] with NT = 2*ceil(N/32),
    KS = ceil(K/(4*epc)); MFMA row n of tile 2t+h holds logical channel 32t + 8(n//4) + 4h + n%4 so that a
    lane's accumulators of a tile pair are 8 consecutive output channels."""
    W = _np(W).astype(np.float32)
    N, K = W.shape
    epc = 8 if act_dtype == torch.bfloat16 else 4
    KS = (K + 4 * epc - 1) // (4 * epc)
    NT = (N + 31) // 32 * 2
    Wp = np.zeros((NT * 16, KS * 4 * epc), np.float32)
    n = np.arange(16)
    for nt in range(NT):
        t, h = divmod(nt, 2)
        L = 32 * t + 8 * (n // 4) + 4 * h + (n % 4)
        ok = L < N
        Wp[nt * 16 + n[ok], :K] = W[L[ok]]
    fr = Wp.reshape(NT, 16, KS, 4, epc).transpose(0, 2, 3, 1, 4)          # [NT][KS][q][n][epc]
    fr = np.ascontiguousarray(fr).reshape(NT, KS, 64, epc)
    return torch.from_numpy(fr).to(device).to(act_dtype).contiguous()


def pack_se_bf16(fc1_w, fc2_w, device):
    """SE weights for the bf16 excitation kernel: fc1.weight [R][C][1][1] -> bf16 [C][ceil8(R)] (transposed, zero padded);
    fc2.weight [C][R][1][1] -> bf16 [R][C] (transposed)."""
    w1 = _np(fc1_w).astype(np.float32)
    w1 = w1.reshape(w1.shape[0], -1)
    w2 = _np(fc2_w).astype(np.float32)
    w2 = w2.reshape(w2.shape[0], -1)
    R, C = w1.shape
    R8 = (R + 7) // 8 * 8
    p1 = np.zeros((C, R8), np.float32)
    p1[:, :R] = w1.T
    bf = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(device).to(torch.bfloat16).contiguous()   # noqa: E731
    return dict(se_w1p=bf(p1), se_w2p=bf(w2.T))


def pack_mfma_frags(W, device, rows=None, ks_mult=1):
    """A dense [N][K] weight as MFMA A-operand fragments [ceil(N/16)][ceil(K/32)][64][8] (bf16, zero padded; lane l holds row
    l&15, k = 8*(l>>4)+j of the 16 x 32 tile).  rows: pad N up to this many rows (whole channel slabs); ks_mult: pad the
    k-steps to a multiple of this (sgp_gemm: whole super-iterations of its chunk ring, 12)."""
    W = _np(W).astype(np.float32)
    W = W.reshape(W.shape[0], -1)
    N, K = W.shape
    NT, KS = (max(N, rows or 0) + 15) // 16, (K + 31) // 32
    KS = (KS + ks_mult - 1) // ks_mult * ks_mult
    Wp = np.zeros((NT * 16, KS * 32), np.float32)
    Wp[:N, :K] = W
    fr = Wp.reshape(NT, 16, KS, 4, 8).transpose(0, 2, 3, 1, 4)
    return torch.from_numpy(np.ascontiguousarray(fr).reshape(NT, KS, 64, 8)).to(device).to(torch.bfloat16).contiguous()


def pack_se_mfma(fc1_w, fc2_w, device):
    """SE weights as MFMA A-operand fragments: fc1.weight [R][C] -> [ceil(R/16)][ceil(C/32)][64][8],
    fc2.weight [C][R] -> [ceil(C/16)][ceil(R/32)][64][8] (pack_mfma_frags)."""
    return dict(w1f=pack_mfma_frags(fc1_w, device), w2f=pack_mfma_frags(fc2_w, device))


GS_SLICE = True
# gate-shift-fuse slice left in source channel order, the module's interleave folded into conv1's weight columns
GS_SRC_ORDER = os.environ.get("TDEED_GS_SRC_ORDER", "1") == "1"
WS_NARROW_ONLY = True
WS_WIDE_MIN_ROWS = 250000     # 0: never the sliced form
RS_MIN_ROWS = int(os.environ.get("TDEED_RS_MIN_ROWS", "60000"))                # 0: never the register-stationary kernel


class DenseW:
    """A dense [N][K] weight in the layout the chosen contraction kernel wants."""

    def __init__(self, W, act_dtype, device, gated=False):
        """gated: the layer is a conv3 (SE gate on its operand + residual)."""
        W = _np(W)
        self.N, self.K = W.shape
        # mode 1: the whole W sits in LDS; mode 2: W does not fit, equal column slices over blockIdx.y (activations re-read per
        # slice, 8 waves per workgroup).
        mode = ops.gemm_ws_fits_mode(self.K, self.N, act_dtype) if str(device) != "cpu" else 0
        self.ws = mode == 1
        # The weight-stationary kernel was built for the narrow RegNetY-200MF layers (24 .. 152 channels), where it wins by
        # 20-35 %; on the 64- and 128-wide layers of the 800MF trunk the tiled kernel is the faster one
        # (tools/bench_ws_vs_gemm.py, us: 64x64 conv1 135 vs 152, conv3 190 vs 221; 128x128 conv3 101 vs 123; conv1 a tie)
        if self.ws and WS_NARROW_ONLY and ((self.K == 64 and self.N == 64) or (gated and self.K >= 64 and self.K % 64 == 0)):
            self.ws = False
        self.w = pack_ws_weights(W, act_dtype, device) if self.ws else _dense(W, act_dtype, device)
        # wide layers over MANY rows (the 320-wide s3 layers of RegNetY-800MF at 3 * 10^5 rows): the sliced form reads the
        # activations once per slice instead of once per 64-column tile -- tools/bench_ws_wide.py, M = 313600, K = N = 320: 134
        # vs 180 us.  Below that the operands of the micro-benchmark sit in the Infinity Cache and in the forward, where two
        # sub-batches share the chip, a kernel that takes whole CUs (150 KB of LDS) pushes the other stream's launches out:
        # 800MF B = 16 (156 800 rows per sub-batch) 1437 vs 1460 clips/s with it, cfg2 3525 vs 3560 -- hence the threshold
        self.wide = (not self.ws) and mode == 2 and WS_WIDE_MIN_ROWS > 0
        self.w_wide = pack_ws_weights(W, act_dtype, device) if self.wide else None
        self.kernel = "gemm_ws" if self.ws else "gemm"

    def _use_wide(self, M):
        return self.wide and M is not None and M >= WS_WIDE_MIN_ROWS

    def _use_rs(self, M, kw):
        """register-stationary kernel (K = N = 320, W in the registers of a 10-wave workgroup) -- conv1-shaped calls and, since
        the kernel re-reads its BatchNorm fold from LDS per 16-row tile instead of holding it across the MFMA loops (no spills
        left at its 168-register budget), conv3 with the SE re-scale and the residual too (tools/bench_ws_wide.py at
        M = 313 600: plain 114 vs 181 us tiled, conv3 172 vs 247; the a_scale tile must span at most two frames)"""
        return (RS_MIN_ROWS > 0 and self.wide and M is not None and M >= RS_MIN_ROWS
                and (kw.get("a_scale") is None or kw.get("a_scale_rows", 0) >= 64)
                and kw.get("gather") is None and kw.get("lda") is None and kw.get("ldc") is None
                and ops.gemm_rs_fits(M, self.K, self.N))

    def kern(self, M):
        """kernel family that serves this layer at M rows"""
        return "gemm_ws" if (self.ws or self._use_wide(M)) else "gemm"

    def run(self, A, scale, shift, act, **kw):
        if self.ws:
            return ops.gemm_ws(A, self.w, self.K, self.N, scale, shift, act, **kw)
        if act in (ops.ACT_NONE, ops.ACT_RELU) and self._use_rs(kw.get("M"), kw):
            return ops.gemm_rs(A, self.w_wide, self.K, self.N, scale, shift, act,
                               **{k: v for k, v in kw.items() if k in ("A0", "k0", "out", "M", "out2", "residual", "a_scale",
                                                                       "a_scale_rows")})
        if self._use_wide(kw.get("M")):
            return ops.gemm_ws(A, self.w_wide, self.K, self.N, scale, shift, act, **kw)
        return ops.gemm(A, self.w, scale, shift, act, **kw)


def pack_front_weights(stem_w, stem_sc, stem_sh, w1, sc1, sh1, wd, scd, shd, w2, gw, sc2, sh2, device):
    """Weight fragments of s1_front_kernel (front.hip).  stem_w [32][3][3][3]; w1/wd [C1][32]; w2 [C1][gw][3][3].
    Stem k-slot s = 4ks+q -> (ky = s>>1, half = s&1), element j -> (kx = 2half + j//4, c = j%4 (3 = pad));
    conv1/downsample k-slot q element j -> stem channel 4q+j (j<4) / 16+4q+j-4: the order in which the stem's
    MFMA accumulators hand the 32 channels over."""
    stem_w, w1, wd = _np(stem_w).astype(np.float32), _np(w1).astype(np.float32), _np(wd).astype(np.float32)
    C1 = w1.shape[0]
    sw = _stem_frags_np(stem_w)
    nt = (C1 + 15) // 16

    def kperm(W):
        fr = np.zeros((nt, 64, 8), np.float32)
        for t in range(nt):
            for q in range(4):
                for j in range(8):
                    chn = 4 * q + j if j < 4 else 16 + 4 * q + j - 4
                    for n in range(16):
                        if t * 16 + n < C1:
                            fr[t, q * 16 + n, j] = W[t * 16 + n, chn]
        return fr
    bf = lambda a: torch.from_numpy(a).to(device).to(torch.bfloat16).contiguous()      # noqa: E731
    f32 = lambda a: _f32(_np(a), device)                                              # noqa: E731
    return SimpleNamespace(C1=C1, stem_wf=bf(sw), stem_sc=f32(stem_sc), stem_sh=f32(stem_sh), w1f=bf(kperm(w1)),
                           sc1=f32(sc1), sh1=f32(sh1), wdf=bf(kperm(wd)), scd=f32(scd), shd=f32(shd),
                           w2f=pack_gconv_frags(w2, gw, device), sc2=f32(sc2), sh2=f32(sh2))


def _stem_frags_np(stem_w):
    """stem conv weight [32][3][3][3] -> MFMA A fragments [2 channel tiles][2 k-steps][64 lanes][8] (front.hip: k-slot
    s = 4ks+q -> (ky = s>>1, half = s&1), element j -> (kx = 2half + j//4, c = j%4, 3 = pad))."""
    sw = np.zeros((2, 2, 64, 8), stem_w.dtype)
    for t in range(2):
        for ks in range(2):
            for q in range(4):
                s_ = 4 * ks + q
                if s_ >= 6:
                    continue
                ky, half = s_ >> 1, s_ & 1
                for j in range(8):
                    kx, c = 2 * half + j // 4, j % 4
                    if kx > 2 or c > 2:
                        continue
                    sw[t, ks, q * 16:(q + 1) * 16, j] = stem_w[t * 16:(t + 1) * 16, c, ky, kx]
    return sw


_STEM_IDX = {}


def stem_frags_on_device(w):
    """_features.stem.conv.weight (32,3,3,3) fp32 on the device -> the fragments of _stem_frags_np (kept in fp32: the training
    stem splits them into bf16 head + tail itself) by one gather through a cached index map (the training step re-packs the
    updated weight without a host round trip)."""
    key = str(w.device)
    if key not in _STEM_IDX:
        ids = (np.arange(32 * 27, dtype=np.float32) + 1).reshape(32, 3, 3, 3)              # exact in fp32
        _STEM_IDX[key] = torch.from_numpy(_stem_frags_np(ids).astype(np.int64)).to(w.device)
    ext = torch.cat([torch.zeros(1, dtype=w.dtype, device=w.device), w.reshape(-1)])
    return ext[_STEM_IDX[key]].contiguous()


_GSFQ_IDX = {}


def gs_source_order_columns(w1, F):
    """conv1 weight (cout, cin) of a gate-shift-fuse site -> the same weight for a slice left in SOURCE channel order:
    out[:, ci] = w1[:, co] for the output channel co that source channel ci is interleaved to (impl/gsf.py:88-91);
    columns >= F unchanged."""
    w = np.array(w1, copy=True)
    src = ops.gs_source_order(F)                  # src[co] = ci
    w[:, src] = w1[:, :F]
    return w


def gsf_q_frags_on_device(w3d):
    """conv3D.weight (2,F/2,3,3,3) fp32 on the device -> the bf16 MFMA fragments of pack_gsf_q_frags, by one gather
    through a cached index map (a training step re-packs the updated weight without a host round trip)."""
    Fh = w3d.shape[1]
    key = (Fh, str(w3d.device))
    if key not in _GSFQ_IDX:
        ids = (np.arange(2 * Fh * 27, dtype=np.float32) + 1).reshape(2, Fh, 3, 3, 3)       # exact in fp32
        _GSFQ_IDX[key] = torch.from_numpy(_gsf_q_frags_np(ids).astype(np.int64)).to(w3d.device)
    from . import repack as R
    ext = torch.cat([torch.zeros(1, dtype=w3d.dtype, device=w3d.device), w3d.reshape(-1)])
    return R.to_bf16(ext[_GSFQ_IDX[key]]).contiguous()


def _gsf_q_frags_np(w3d):
    Fh = w3d.shape[1]
    F = 2 * Fh
    nch = (F + 7) // 8
    KS = (9 * nch + 3) // 4
    fr = np.zeros((KS, 64, 8), np.float32)
    for ks in range(KS):
        for q in range(4):
            s_ = 4 * ks + q
            tap, ck = divmod(s_, nch)
            if tap >= 9:
                continue
            dy, dx = divmod(tap, 3)
            for n in range(6):
                jt, g = divmod(n, 2)
                for e in range(8):
                    c = ck * 8 + e
                    if c < F and c // Fh == g:
                        fr[ks, q * 16 + n, e] = w3d[g, c - g * Fh, jt, dy, dx]
    return fr


def pack_gsf_q_frags(w3d, device):
    """conv3D.weight [2][F/2][3][3][3] -> bf16 MFMA A fragments [KS][64][8] for gsf_q_mfma_kernel:
    row n = jg = 2*j_t + g (rows 6..15 zero); k-slot s = 4ks+q = tap*nch + chunk, element e = channel 8*chunk+e,
    non-zero only for channels of gate group g."""
    return torch.from_numpy(_gsf_q_frags_np(_np(w3d).astype(np.float32))).to(device).to(torch.bfloat16).contiguous()


def _gsf_p_frags_np(w3d):
    Fh = w3d.shape[1]
    F = 2 * Fh
    nch = (F + 7) // 8
    KSc = (nch + 3) // 4
    fr = np.zeros((4, KSc, 64, 8), np.float32)
    for rt in range(4):
        for n in range(16):
            r = rt * 16 + n
            if r >= 54:
                continue
            tap, jg = divmod(r, 6)
            dy, dx = divmod(tap, 3)
            jt, g = divmod(jg, 2)
            for ks in range(KSc):
                for q in range(4):
                    for e in range(8):
                        c = (4 * ks + q) * 8 + e
                        if c < F and c // Fh == g:
                            fr[rt, ks, q * 16 + n, e] = w3d[g, c - g * Fh, jt, dy, dx]
    return fr


def pack_gsf_p_frags(w3d, device):
    """conv3D.weight [2][F/2][3][3][3] -> bf16 MFMA A fragments [4][ceil(nch/4)][64][8] for the tap-map tail of
    tdeed_bneck_gs_fwd: row r = tap*6 + jg (jg = 2*j_t + g as in pack_gsf_q_frags; rows 54..63 zero), k = channel,
    non-zero only for channels of gate group g -- the 3x3x3 conv as ONE 1x1 contraction to per-tap sums."""
    return torch.from_numpy(_gsf_p_frags_np(_np(w3d).astype(np.float32))).to(device).to(torch.bfloat16).contiguous()


def pack_gconv_frags(w, gw, device, tap_major=False):
    """Conv2d.weight [C][gw][3][3] -> bf16 MFMA A-operand fragments [ceil4(C/16)][5][64][8] for
    gconv3x3_mfma_kernel: unit u = output channels [16u,16u+16); lane l holds Wt[n=l&15][k=8(l>>4)+j];
    k-slot s = 4*ks + (l>>4) = half*9 + tap; for gw=8 'half' selects which of the unit's two groups
    the 8 input channels belong to (block-diagonal), for gw=16 which half of the group's 16 inputs.
    tap_major (tdeed_bneck_fwd): s = 2*tap + half -- the two k-slots of a ds_read_b128 lane group then differ by 16 bytes
    at the SAME tap pixel, which is conflict-free at the one-launch bottleneck's even row stride (bneck.hip)."""
    return torch.from_numpy(_gconv_frags_np(_np(w).astype(np.float32), gw, tap_major)).to(device).to(torch.bfloat16).contiguous()


def _gconv_frags_np(w, gw, tap_major=False):
    C = w.shape[0]
    nu = (C + 15) // 16
    nu4 = (nu + 3) // 4 * 4
    fr = np.zeros((nu4, 5, 64, 8), np.float32)
    for u in range(nu):
        for ks in range(5):
            for q in range(4):
                s_ = 4 * ks + q
                if s_ >= 18:
                    continue
                half, tap = (s_ & 1, s_ >> 1) if tap_major else divmod(s_, 9)
                ky, kx = divmod(tap, 3)
                for n in range(16):
                    co = u * 16 + n
                    if co >= C:
                        continue
                    lane = q * 16 + n
                    if gw == 16:
                        fr[u, ks, lane, :] = w[co, half * 8:half * 8 + 8, ky, kx]
                    elif n // 8 == half:
                        fr[u, ks, lane, :] = w[co, :, ky, kx]
    return fr


_GCONV_IDX = {}


def gconv_frag_index(C, gw, device):
    """Index map of pack_gconv_frags: frags = cat([0, w.reshape(-1)])[idx], so a training step can re-pack the MFMA
    fragments of an updated weight on the device (one gather) instead of through numpy."""
    key = (C, gw, str(device))
    if key not in _GCONV_IDX:
        ids = (np.arange(C * gw * 9, dtype=np.float32) + 1).reshape(C, gw, 3, 3)     # exact in fp32 (< 2^24)
        fr = _gconv_frags_np(ids, gw)
        _GCONV_IDX[key] = torch.from_numpy(fr.astype(np.int64)).to(device)
    return _GCONV_IDX[key]


def gconv_frags_on_device(w, gw):
    """Conv2d.weight (C,gw,3,3) fp32 on the device -> bf16 MFMA fragments (same layout as pack_gconv_frags)."""
    idx = gconv_frag_index(w.shape[0], gw, w.device)
    from . import repack as R
    ext = torch.cat([torch.zeros(1, dtype=w.dtype, device=w.device), w.reshape(-1)])
    return R.to_bf16(ext[idx]).contiguous()


def pack_sgp_block(sd, pre, C, act_dtype, device):
    """SGPBlock parameters (modules.py:91-145) in kernel layout."""
    o = SimpleNamespace(C=C)
    o.ln_w = _f32(_np(sd[pre + ".ln.weight"]).reshape(C), device)
    o.ln_b = _f32(_np(sd[pre + ".ln.bias"]).reshape(C), device)
    o.dw, o.db = _dwpack(sd, pre, ["psi", "convw", "convkw", "fc", "global_fc"], C, device)
    o.ks = _np(sd[pre + ".psi.weight"]).shape[-1]
    o.up = _np(sd[pre + ".convkw.weight"]).shape[-1]
    _pack_mlp(sd, pre, C, o, act_dtype, device)
    return o


def pack_sgp_mixer(sd, pre, C, act_dtype, device):
    """SGPMixer parameters (modules.py:192-254) in kernel layout."""
    o = SimpleNamespace(C=C)
    for n in ("ln1", "ln2"):
        setattr(o, n + "_w", _f32(_np(sd[f"{pre}.{n}.weight"]).reshape(C), device))
        setattr(o, n + "_b", _f32(_np(sd[f"{pre}.{n}.bias"]).reshape(C), device))
    o.dw1, o.db1 = _dwpack(sd, pre, ["psi1", "convw1", "convkw1", "fc1", "global_fc1"], C, device)
    o.dw2, o.db2 = _dwpack(sd, pre, ["psi2", "convw2", "convkw2", "fc2", "global_fc2"], C, device)
    o.ks = _np(sd[pre + ".psi1.weight"]).shape[-1]
    o.up = _np(sd[pre + ".convkw1.weight"]).shape[-1]
    o.w_cat = _dense(_np(sd[pre + ".concat_fc.weight"]).reshape(C, 6 * C), act_dtype, device)
    o.b_cat = _f32(_np(sd[pre + ".concat_fc.bias"]), device)
    o.wcg = (pack_mfma_frags(_np(sd[pre + ".concat_fc.weight"]).reshape(C, 6 * C), device, ks_mult=12)
             if act_dtype == torch.bfloat16 and str(device) != "cpu" and C % 16 == 0 and SGP_GEMM else None)
    _pack_mlp(sd, pre, C, o, act_dtype, device)
    return o


def _se(pooled, inv_cnt, bw, gate):
    """SE excitation: bf16 packed weights in throughput mode, fp32 weights in parity mode."""
    if bw.se_mf is not None:
        return ops.se_gate_mfma(pooled, inv_cnt, bw.se_mf.w1f, bw.se_b1, bw.se_mf.w2f, bw.se_b2, bw.spec.se_rd, out=gate)
    if bw.se_bf is not None:
        return ops.se_gate_bf16(pooled, inv_cnt, bw.se_bf.se_w1p, bw.se_b1, bw.se_bf.se_w2p, bw.se_b2, bw.spec.se_rd, out=gate)
    return ops.se_gate(pooled, inv_cnt, bw.se_w1t, bw.se_b1, bw.se_w2t, bw.se_b2, out=gate)


BNECK_ONE_LAUNCH = os.environ.get("TDEED_BNECK", "1") == "1"
# the gate-shift-fuse blend inside the one-launch bottleneck's frame load (tdeed_bneck_gs_fwd): the site's third launch and the
# round trip of its output slice are gone
BNECK_BLEND = True
# ... and the tap maps of the NEXT block's site in the same launch's tail (tdeed_bneck_gs_fwd's Q): that site's first launch is gone
BNECK_QTAIL = True
C1_GCONV = os.environ.get("TDEED_C1_GCONV", "1") == "1"           # conv1 (+ downsample) computed inside the grouped conv's launch
C1_GCONV_MAX_CIN = 160


def _bneck_fused(bw, h, w, out_is_slice):
    """True when the whole block runs as ONE launch (tdeed_bneck_fwd): bf16, stride 1, identity shortcut, a map small enough
    for a workgroup's frames to stay in LDS (s3.b2-b4 and s4.b2-b7 of RegNetY-200MF), contiguous output."""
    blk = bw.spec
    return bool(BNECK_ONE_LAUNCH and getattr(bw, "fused", None) is not None and bw.se_mf is not None and blk.stride == 1
                and not blk.has_downsample and blk.cin == blk.cout and not out_is_slice
                and ops.bneck_fits(h, w, blk.cout, blk.se_rd))


class Step:
    """One kernel launch of a plan with its algorithmic cost (what a perfect kernel must move / compute)."""
    __slots__ = ("name", "kernel", "fn", "bytes", "flops")

    def __init__(self, name, kernel, fn, nbytes=0, flops=0):
        self.name, self.kernel, self.fn, self.bytes, self.flops = name, kernel, fn, int(nbytes), int(flops)


def _esz(dt):
    return 2 if dt == torch.bfloat16 else 4


def gemm_cost(M, K, N, es, residual=False, extra=0):
    return (M * K + N * K + M * N * (2 if residual else 1)) * es + extra, 2 * M * K * N


class SgpBuilder:
    """Appends the launches of SGP blocks / mixers / the whole encoder-decoder to a step list."""

    def __init__(self, pool, steps, keep, taps, B, act_dtype):
        self.pool, self.steps, self.keep, self.taps, self.B, self.dt = pool, steps, keep, taps, B, act_dtype
        self.splitk_rows = 4096
        # fused launches (sgp_fused.hip): LayerNorm inside the branch kernels (both dtypes), GroupNorm + fc1 + GELU + fc2 +
        # residual in one MFMA launch (bf16).  TDEED_SGP_FUSED=0 restores the launch-per-op chain (A/B measurements).
        self.fused = os.environ.get("TDEED_SGP_FUSED", "1") == "1"
        # round 5: the contractions on sgp_gemm.hip (full K per workgroup, no fp32 partials, no fold launches); `adt` is the
        # type of the stage's residual stream -- fp32 under a bf16 trunk unless TDEED_SGP_F32_STREAM=0
        self.gemm = SGP_GEMM and act_dtype == torch.bfloat16
        # (the residual stream's type is the type of what the stage is handed: sgp_stream_dtype() for the model's plans)

    def _mlp_gemm(self, name, y, o, Tn, chsum, pool_to=None, y16=None):
        """out = y + mlp(GN(y)) as two sgp_gemm launches; leaves the output's partial row sums on it (`_td_rowstat`) and, when
        the following AdaptiveMaxPool1d halves the length, the pooled rows in `self.last_pooled`."""
        pool, steps, B, C = self.pool, self.steps, self.B, o.C
        R, adt = B * Tn, y.dtype
        es = _esz(adt)
        ya = y if y16 is None else y16           # fc1's operand: the rows themselves, or their bf16 copy (wide models, fp32 stream)
        f1 = ops.sgp_gemm_form(0 if ya.dtype == torch.bfloat16 else 3, B, Tn, 4 * C, C)
        f2 = ops.sgp_gemm_form(1, B, Tn, C, 4 * C)
        nct = ops.sgp_gemm_tiles(Tn, C, f2)[1]
        H = pool.take((B, Tn, 4 * C), torch.bfloat16)
        outb = pool.take((B, Tn, C), adt)
        rsp = torch.empty((nct, R, 2), dtype=torch.float32, device=y.device)          # lives as long as the plan (tiny)
        pooled = rpp = None
        if pool_to is not None and 2 * pool_to == Tn:
            pooled = pool.take((B, pool_to, C), adt)
            rpp = torch.empty((nct, B * pool_to, 2), dtype=torch.float32, device=y.device)
            pooled._td_rowstat = rpp
        self.last_pooled = pooled
        steps.append(Step(name + ".fc1", "sgp_gemm", lambda: ops.sgp_gemm_gn_gelu(ya, chsum, o.gn_w, o.gn_b, o.w1g, o.b_fc1, 4 * C,
                                                                                out=H, form=f1),
                          R * C * _esz(ya.dtype) + 4 * C * C * 2 + R * 4 * C * 2, 2 * R * 4 * C * C))
        steps.append(Step(name + ".fc2", "sgp_gemm", lambda: ops.sgp_gemm_residual(H, o.w2g, o.b_fc2, y, out=outb, rowstat_part=rsp,
                                                                                 pooled=pooled, rowstat_pool_part=rpp, form=f2),
                          R * 4 * C * 2 + 4 * C * C * 2 + 2 * R * C * es + (0 if pooled is None else B * pool_to * C * es),
                          2 * R * 4 * C * C))
        pool.give(H)
        outb._td_rowstat = rsp
        return outb

    def dense(self, name, A, Wt, bias, act, out, R, residual=None):
        """One Conv1d(k=1) of the mlp / concat_fc.  Short sequences (bf16, <= splitk_rows rows) go through the split-K
        kernel: a tiled contraction would be ~40 workgroups each walking K in 20+ dependent round trips."""
        N, K = Wt.shape
        es = _esz(self.dt)
        if (self.dt == torch.bfloat16 and R <= self.splitk_rows and str(A.device) != "cpu"
                and ops.gemm_splitk_splits(K) >= 4):      # K = 4C / 6C; for K = C the tiled kernel is faster (11 vs 15 us)
            ws = self.pool.take((ops.gemm_splitk_splits(K), R, N), torch.float32)
            self.steps.append(Step(name, "gemm_splitk", lambda: ops.gemm_splitk(A, Wt, None, bias, act, residual=residual,
                                                                                 out=out, M=R, workspace=ws),
                                   *gemm_cost(R, K, N, es, residual is not None)))
            self.pool.give(ws)
        else:
            self.steps.append(Step(name, "gemm", lambda: ops.gemm(A, Wt, None, bias, act, residual=residual, out=out, M=R),
                                   *gemm_cost(R, K, N, es, residual is not None)))

    def _mlp(self, name, y, o, outb, Tn):
        """out = y + mlp(GN(y)) as groupnorm + two contractions (fp32 mode and TDEED_SGP_GEMM=0; the bf16 path is _mlp_gemm)."""
        pool, steps, B, C, dt = self.pool, self.steps, self.B, o.C, self.dt
        es, R = _esz(dt), B * Tn
        self.last_pooled = None
        gn = pool.take((B, Tn, C), dt)
        hid = pool.take((B, Tn, 4 * C), dt)
        steps.append(Step(name + ".gn", "groupnorm", lambda: ops.groupnorm(y, 16, o.gn_w, o.gn_b, out=gn), 2 * R * C * es))
        self.dense(name + ".fc1", gn, o.w_fc1, o.b_fc1, ops.ACT_GELU, hid, R)
        self.dense(name + ".fc2", hid, o.w_fc2, o.b_fc2, ops.ACT_NONE, outb, R, residual=y)
        pool.give(gn)
        pool.give(hid)

    def block(self, xin, Tn, o, name, pool_to=None):
        """One SGPBlock.  pool_to: length of the AdaptiveMaxPool1d that follows (encoder half); when the MLP launch can
        carry it, `self.last_pooled` holds the pooled tensor afterwards (else None: the caller adds a max-pool launch)."""
        pool, steps, B, C, dt = self.pool, self.steps, self.B, o.C, self.dt
        self.last_pooled = None
        if self.gemm and self.fused and str(xin.device) != "cpu" and getattr(o, "w1g", None) is not None:
            adt = xin.dtype
            es, R = _esz(adt), B * Tn
            wl = 2 * o.ks + o.up + 2
            y = pool.take((B, Tn, C), adt)
            y16 = pool.take((B, Tn, C), torch.bfloat16) if (adt == torch.float32 and C > SGP_BF16_OPERAND_MIN_C) else None
            chs = pool.take((B, C, 2), torch.float32)
            rs_in = getattr(xin, "_td_rowstat", None)
            steps.append(Step(name + ".front", "sgp_front", lambda: ops.sgp_front(xin, o.ks, o.up, o.ln_w, o.ln_b, o.dw, o.db,
                                                                                 out=y, chsum=chs, rowstat=rs_in, out16=y16),
                              2 * R * C * es + C * (wl + 7) * 4 + (0 if y16 is None else R * C * 2), 2 * R * C * (wl + 3)))
            outb = self._mlp_gemm(name, y, o, Tn, chs, pool_to=pool_to, y16=y16)
            pool.give(y)
            if y16 is not None:
                pool.give(y16)
            pool.give(chs)
            if name in self.taps:
                self.keep[name] = outb
            return outb
        if self.fused and str(xin.device) != "cpu":
            y = pool.take((B, Tn, C), dt)
            outb = pool.take((B, Tn, C), dt)
            es, R = _esz(dt), B * Tn
            wl = 2 * o.ks + o.up + 2
            rs_in = getattr(xin, "_td_rowstat", None)          # LayerNorm statistics left by the producer of xin (avgpool_posenc)
            steps.append(Step(name + ".front", "sgp_front", lambda: ops.sgp_front(xin, o.ks, o.up, o.ln_w, o.ln_b, o.dw, o.db,
                                                                                 out=y, rowstat=rs_in),
                              2 * R * C * es + C * (wl + 7) * 4, 2 * R * C * (wl + 3)))
            self._mlp(name, y, o, outb, Tn)
            pool.give(y)
            if name in self.taps:
                self.keep[name] = outb
            return outb
        ln = pool.take((B, Tn, C), dt)
        y = pool.take((B, Tn, C), dt)
        gn = pool.take((B, Tn, C), dt)
        hid = pool.take((B, Tn, 4 * C), dt)
        outb = pool.take((B, Tn, C), dt)
        es, R = _esz(dt), B * Tn
        wl = 2 * o.ks + o.up + 2
        steps.append(Step(name + ".ln", "layernorm", lambda: ops.layernorm(xin, o.ln_w, o.ln_b, out=ln), 2 * R * C * es))
        steps.append(Step(name + ".branch", "sgp_branch", lambda: ops.sgp_branch(ln, xin, o.ks, o.up, o.dw, o.db, out=y),
                          3 * R * C * es + C * (wl + 5) * 4, 2 * R * C * (wl + 3)))
        steps.append(Step(name + ".gn", "groupnorm", lambda: ops.groupnorm(y, 16, o.gn_w, o.gn_b, out=gn), 2 * R * C * es))
        self.dense(name + ".fc1", gn, o.w_fc1, o.b_fc1, ops.ACT_GELU, hid, R)
        self.dense(name + ".fc2", hid, o.w_fc2, o.b_fc2, ops.ACT_NONE, outb, R, residual=y)
        for t_ in (ln, y, gn, hid):
            pool.give(t_)
        if name in self.taps:
            self.keep[name] = outb
        return outb

    def mixer(self, xlo, T_lo, z, T_hi, o, name):
        pool, steps, B, C, dt = self.pool, self.steps, self.B, o.C, self.dt
        if self.gemm and self.fused and str(z.device) != "cpu" and getattr(o, "wcg", None) is not None:
            adt = z.dtype
            if xlo.dtype != adt:
                raise TypeError("SgpBuilder.mixer: z and x_lo must share the residual stream's type")
            es, R, Rl = _esz(adt), B * T_hi, B * T_lo
            wl = 2 * o.ks + o.up + 2
            cat = pool.take((B, T_hi, 6 * C), torch.bfloat16)
            rs_z, rs_x = getattr(z, "_td_rowstat", None), getattr(xlo, "_td_rowstat", None)
            steps.append(Step(name + ".front", "mixer_front",
                              lambda: ops.mixer_front(z, xlo, cat, o.ks, o.up, o.ln1_w, o.ln1_b, o.ln2_w, o.ln2_b, o.dw1,
                                                      o.db1, o.dw2, o.db2, rowstat_z=rs_z, rowstat_x=rs_x),
                              (R + Rl) * C * es + 6 * R * C * 2 + 2 * C * (wl + 7) * 4, 4 * R * C * (wl + 3)))
            fc = ops.sgp_gemm_form(2, B, T_hi, C, 6 * C)
            NJ = ops.sgp_gemm_tiles(T_hi, C, fc)[0]
            mo = pool.take((B, T_hi, C), adt)
            mo16 = pool.take((B, T_hi, C), torch.bfloat16) if (adt == torch.float32 and C > SGP_BF16_OPERAND_MIN_C) else None
            chs = pool.take((NJ, B, C, 2), torch.float32)
            steps.append(Step(name + ".cat", "sgp_gemm", lambda: ops.sgp_gemm_gelu_chsum(cat, o.wcg, o.b_cat, C, mo, chs, form=fc,
                                                                                       out16=mo16),
                              R * 6 * C * 2 + 6 * C * C * 2 + R * C * es + (0 if mo16 is None else R * C * 2), 2 * R * 6 * C * C))
            pool.give(cat)
            outb = self._mlp_gemm(name, mo, o, T_hi, chs, y16=mo16)
            pool.give(mo)
            if mo16 is not None:
                pool.give(mo16)
            pool.give(chs)
            if name in self.taps:
                self.keep[name] = outb
            return outb
        if self.fused and str(z.device) != "cpu":
            cat = pool.take((B, T_hi, 6 * C), dt)
            mo = pool.take((B, T_hi, C), dt)
            outb = pool.take((B, T_hi, C), dt)
            es, R, Rl = _esz(dt), B * T_hi, B * T_lo
            wl = 2 * o.ks + o.up + 2
            rs_z, rs_x = getattr(z, "_td_rowstat", None), getattr(xlo, "_td_rowstat", None)
            steps.append(Step(name + ".front", "mixer_front",
                              lambda: ops.mixer_front(z, xlo, cat, o.ks, o.up, o.ln1_w, o.ln1_b, o.ln2_w, o.ln2_b, o.dw1,
                                                      o.db1, o.dw2, o.db2, rowstat_z=rs_z, rowstat_x=rs_x),
                              (7 * R + Rl) * C * es + 2 * C * (wl + 7) * 4, 4 * R * C * (wl + 3)))
            self.dense(name + ".cat", cat, o.w_cat, o.b_cat, ops.ACT_GELU, mo, R)
            self._mlp(name, mo, o, outb, T_hi)
            pool.give(cat)
            pool.give(mo)
            if name in self.taps:
                self.keep[name] = outb
            return outb
        cat = pool.take((B, T_hi, 6 * C), dt)
        xn = pool.take((B, T_lo, C), dt)
        mo = pool.take((B, T_hi, C), dt)
        gn = pool.take((B, T_hi, C), dt)
        hid = pool.take((B, T_hi, 4 * C), dt)
        outb = pool.take((B, T_hi, C), dt)
        zslab = cat.view(-1)[4 * C:]
        es, R, Rl = _esz(dt), B * T_hi, B * T_lo
        wl = 2 * o.ks + o.up + 2
        steps.append(Step(name + ".ln1", "layernorm", lambda: ops.layernorm(z, o.ln1_w, o.ln1_b, out=zslab, ldy=6 * C,
                                                                            rows=R, C=C), 2 * R * C * es))
        steps.append(Step(name + ".ln2", "layernorm", lambda: ops.layernorm(xlo, o.ln2_w, o.ln2_b, out=xn), 2 * Rl * C * es))
        steps.append(Step(name + ".branch", "mixer_branch",
                          lambda: ops.mixer_branch(xn, cat, T_hi, o.ks, o.up, o.dw1, o.db1, o.dw2, o.db2),
                          (6 * R + Rl) * C * es + 2 * C * (wl + 5) * 4, 4 * R * C * (wl + 3)))
        self.dense(name + ".cat", cat, o.w_cat, o.b_cat, ops.ACT_GELU, mo, R)
        steps.append(Step(name + ".gn", "groupnorm", lambda: ops.groupnorm(mo, 16, o.gn_w, o.gn_b, out=gn), 2 * R * C * es))
        self.dense(name + ".fc1", gn, o.w_fc1, o.b_fc1, ops.ACT_GELU, hid, R)
        self.dense(name + ".fc2", hid, o.w_fc2, o.b_fc2, ops.ACT_NONE, outb, R, residual=mo)
        for t_ in (cat, xn, mo, gn, hid):
            pool.give(t_)
        if name in self.taps:
            self.keep[name] = outb
        return outb

    def pyramid(self, feat, T, n, sgp, mixers, pre="_temp_fine."):
        """EDSGPMIXERLayers.forward (modules.py:69-87) on an NTC tensor."""
        pool, steps, B, dt = self.pool, self.steps, self.B, self.dt
        C = sgp[0].C
        lens = pyramid_lengths(T, n)
        cur = feat
        stash = []
        for i in range(n):
            cur = self.block(cur, lens[i], sgp[i], f"{pre}_sgp.{i}", pool_to=lens[i + 1])
            stash.append(cur)
            pooled = self.last_pooled
            if pooled is None:
                pooled = pool.take((B, lens[i + 1], C), cur.dtype)
                if self.gemm and self.fused and str(cur.device) != "cpu":
                    # lengths that do not halve: a pooling launch that also leaves the pooled rows' LayerNorm statistics
                    prs = torch.empty((B * lens[i + 1], 2), dtype=torch.float32, device=cur.device)
                    pooled._td_rowstat = prs
                    steps.append(Step(f"{pre}pool{i}", "maxpool", lambda a=cur, b=pooled, L=lens[i + 1], r=prs:
                                      ops.maxpool_rowstat(a, L, out=b, rowstat=r), B * (lens[i] + lens[i + 1]) * C * _esz(cur.dtype)))
                else:
                    steps.append(Step(f"{pre}pool{i}", "maxpool", lambda a=cur, b=pooled, L=lens[i + 1]: ops.maxpool(a, L, out=b),
                                      B * (lens[i] + lens[i + 1]) * C * _esz(cur.dtype)))
            cur = pooled
        cur = self.block(cur, lens[n], sgp[n], f"{pre}_sgp.{n}")
        for i in range(n):
            lvl = n - 1 - i
            cur = self.mixer(cur, lens[lvl + 1], stash[lvl], lens[lvl], mixers[lvl], f"{pre}_sgpMixer.{lvl}")
            cur = self.block(cur, lens[lvl], sgp[n + 1 + i], f"{pre}_sgp.{n + 1 + i}")
        return cur


class PackedWeights:
    """Device-side weights in kernel layouts.  ``state``: reference key grammar."""

    def __init__(self, cfg, state, act_dtype, device):
        g = (lambda k: cfg[k]) if isinstance(cfg, dict) else (lambda k: getattr(cfg, k))
        self.arch = g("feature_arch")
        self.spec = regnet_spec(self.arch)
        self.mode = "gsm" if self.arch.endswith("_gsm") else ("gsf" if self.arch.endswith("_gsf") else None)
        self.clip_len = g("clip_len")
        self.n_layers = g("n_layers")
        self.ks = g("sgp_ks")
        self.up = sgp_up_size(self.ks, g("sgp_r"))
        self.radi = g("radi_displacement")
        self.act_dtype = act_dtype
        self.device = device
        sd = {k: _np(v) for k, v in state.items()}
        self.double_head = "_pred_fine._fc1._fc_out.weight" in sd
        f32 = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(device)   # noqa: E731
        dense = lambda a: f32(a).to(act_dtype).contiguous()                                       # noqa: E731

        def bn_fold(pre):
            w, b = sd[pre + ".weight"].astype(np.float64), sd[pre + ".bias"].astype(np.float64)
            m, v = sd[pre + ".running_mean"].astype(np.float64), sd[pre + ".running_var"].astype(np.float64)
            s = w / np.sqrt(v + BN_EPS)
            return f32(s), f32(b - m * s)

        W = SimpleNamespace()
        p = "_features."
        W.stem_w = f32(sd[p + "stem.conv.weight"].reshape(32, 27))
        W.stem_scale, W.stem_shift = bn_fold(p + "stem.bn")
        W.blocks = []
        for blk in self.spec.blocks:
            bp = p + blk.name
            bw = SimpleNamespace(spec=blk)
            c1 = bp + (".conv1.net" if blk.gsf_fold else ".conv1")
            w1_mat = _np(sd[c1 + ".conv.weight"]).reshape(blk.cout, blk.cin)
            # gate-shift-fuse sites of the bf16 engine: the module's channel interleave is folded into conv1's columns (the
            # blend launch leaves its slice in source channel order, ops.gate_shift(src_order=True)): column ci of the
            # packed weight is the column of the output channel that source channel ci is interleaved to
            bw.gs_src = bool(blk.gsf_fold and GS_SRC_ORDER and self.mode == "gsf" and act_dtype == torch.bfloat16
                             and str(device) != "cpu")
            if bw.gs_src:
                w1_mat = gs_source_order_columns(w1_mat, blk.gsf_fold)
            bw.w1 = DenseW(w1_mat, act_dtype, device)
            bw.w1_raw = w1_mat
            bw.wd_raw = _np(sd[bp + ".downsample.conv.weight"]).reshape(blk.cout, blk.cin) if blk.has_downsample else None
            bw.c1g_w1f = None    # conv1 / downsample as MFMA fragments padded to whole channel slabs
                                              # (tdeed_c1_gconv_fwd), packed on first use
            bw.s1, bw.h1 = bn_fold(c1 + ".bn")
            w2 = sd[bp + ".conv2.conv.weight"]                       # [C][gw][3][3]
            G, gw = blk.groups, blk.gw
            w2 = w2.reshape(G, gw, gw, 3, 3).transpose(0, 3, 4, 2, 1)  # [G][ky][kx][in][out]
            bw.w2 = f32(w2.reshape(G, 9, gw, gw))
            bw.s2, bw.h2 = bn_fold(bp + ".conv2.bn")
            bw.w2frag = pack_gconv_frags(sd[bp + ".conv2.conv.weight"], gw, device) if act_dtype == torch.bfloat16 else None
            bw.se_w1t = f32(sd[bp + ".se.fc1.weight"].reshape(blk.se_rd, blk.cout).T)
            bw.se_b1 = f32(sd[bp + ".se.fc1.bias"])
            bw.se_w2t = f32(sd[bp + ".se.fc2.weight"].reshape(blk.cout, blk.se_rd).T)
            bw.se_b2 = f32(sd[bp + ".se.fc2.bias"])
            bw.se_bf = (SimpleNamespace(**pack_se_bf16(sd[bp + ".se.fc1.weight"], sd[bp + ".se.fc2.weight"], device))
                        if act_dtype == torch.bfloat16 and str(device) != "cpu" else None)
            bw.se_mf = (SimpleNamespace(**pack_se_mfma(sd[bp + ".se.fc1.weight"], sd[bp + ".se.fc2.weight"], device))
                        if (bw.se_bf is not None and True
                            and ops.se_gate_mfma_fits(blk.cout, blk.se_rd)) else None)
            bw.w3 = DenseW(sd[bp + ".conv3.conv.weight"].reshape(blk.cout, blk.cout), act_dtype, device, gated=True)
            # MFMA-fragment copies of conv1 / conv3 for the one-launch bottleneck (stride-1 identity blocks up to 384 wide)
            bw.fused = (SimpleNamespace(w1f=pack_mfma_frags(w1_mat, device),
                                        w2f=pack_gconv_frags(sd[bp + ".conv2.conv.weight"], gw, device, tap_major=(gw == 8)),
                                        w2_tap_major=(gw == 8),
                                        w3f=pack_mfma_frags(sd[bp + ".conv3.conv.weight"].reshape(blk.cout, blk.cout), device))
                        if (bw.se_mf is not None and blk.stride == 1 and not blk.has_downsample and blk.cin == blk.cout
                            and blk.cout <= 384) else None)
            bw.s3, bw.h3 = bn_fold(bp + ".conv3.bn")
            if blk.has_downsample:
                bw.wd = DenseW(sd[bp + ".downsample.conv.weight"].reshape(blk.cout, blk.cin), act_dtype, device)
                bw.sd, bw.hd = bn_fold(bp + ".downsample.bn")
            if blk.gsf_fold:
                gp = bp + ".conv1.gs"
                F = blk.gsf_fold
                bw.gs_scale, bw.gs_shift = bn_fold(gp + ".bn")
                w3d = sd[gp + ".conv3D.weight"]                       # [2][F/2][3][3][3]
                bw.gs_wq = f32(w3d.reshape(F, 27).T)                  # [27][F], c = g*F/2 + cl
                bw.gs_b3d = f32(sd[gp + ".conv3D.bias"])
                bw.gs_wqf = pack_gsf_q_frags(w3d, device) if (act_dtype == torch.bfloat16 and str(device) != "cpu") else None
                bw.gs_bnq = ops.gsq_bn_table(bw.gs_scale, bw.gs_shift) if bw.gs_wqf is not None else None
                bw.gs_wpf = pack_gsf_p_frags(w3d, device) if bw.gs_wqf is not None else None
                if self.mode == "gsf":
                    bw.gs_cw1 = f32(sd[gp + ".channel_conv1.weight"].reshape(18))
                    bw.gs_cb1 = f32(sd[gp + ".channel_conv1.bias"])
                    bw.gs_cw2 = f32(sd[gp + ".channel_conv2.weight"].reshape(18))
                    bw.gs_cb2 = f32(sd[gp + ".channel_conv2.bias"])
                else:
                    bw.gs_cw1 = bw.gs_cb1 = bw.gs_cw2 = bw.gs_cb2 = None
            W.blocks.append(bw)
        W.front = None
        b0 = self.spec.blocks[0]
        if act_dtype == torch.bfloat16 and b0.stride == 2 and b0.has_downsample and b0.cout <= 64 and not b0.gsf_fold \
                and str(device) != "cpu":
            bw0, bp0 = W.blocks[0], p + b0.name
            W.front = pack_front_weights(
                sd[p + "stem.conv.weight"], W.stem_scale, W.stem_shift,
                sd[bp0 + ".conv1.conv.weight"].reshape(b0.cout, b0.cin), bw0.s1, bw0.h1,
                sd[bp0 + ".downsample.conv.weight"].reshape(b0.cout, b0.cin), bw0.sd, bw0.hd,
                sd[bp0 + ".conv2.conv.weight"], b0.gw, bw0.s2, bw0.h2, device)
        W.temp_enc = f32(sd["temp_enc"])
        C = self.spec.feat_dim

        W.sgp = [pack_sgp_block(sd, f"_temp_fine._sgp.{i}", C, act_dtype, device) for i in range(2 * self.n_layers + 1)]
        W.mixer = [pack_sgp_mixer(sd, f"_temp_fine._sgpMixer.{i}", C, act_dtype, device) for i in range(self.n_layers)]
        if self.double_head:
            hw = [sd["_pred_fine._fc1._fc_out.weight"], sd["_pred_fine._fc2._fc_out.weight"]]
            hb = [sd["_pred_fine._fc1._fc_out.bias"], sd["_pred_fine._fc2._fc_out.bias"]]
        else:
            hw, hb = [sd["_pred_fine._fc_out.weight"]], [sd["_pred_fine._fc_out.bias"]]
        self.n_cls = int(sum(w.shape[0] for w in hw))
        if self.radi > 0:
            hw.append(sd["_pred_displ._fc_out.weight"])
            hb.append(sd["_pred_displ._fc_out.bias"])
        W.head_w = f32(np.concatenate(hw, axis=0))
        W.head_b = f32(np.concatenate(hb, axis=0))
        self.n_out = W.head_w.shape[0]
        self.displ_col = self.n_cls if self.radi > 0 else -1
        self.W = W


_DEAD_GRAPHS = []          # graph executables of dropped engines, destroyed by _drain_dead_graphs()
_CAPTURING = 0


def _drain_dead_graphs():
    """Destroy the graph executables of engines that were garbage-collected (safe point: no capture in progress; the device
    is synchronised first so that none of them is still executing)."""
    if not _DEAD_GRAPHS or _CAPTURING:
        return
    torch.cuda.synchronize()
    while _DEAD_GRAPHS:
        _lib.call("tdeed_graph_destroy", _DEAD_GRAPHS.pop())


class ForwardEngine:
    def __init__(self, cfg, state, act_dtype=torch.bfloat16, device="cuda", use_graph=True, fuse_front=True, n_split=2):
        if not torch.cuda.is_available():
            raise RuntimeError("tdeed_amd.ForwardEngine needs an MI355X (no CPU path)")
        _lib.load()
        self.cfg = cfg
        g = (lambda k: cfg[k]) if isinstance(cfg, dict) else (lambda k: getattr(cfg, k))
        self.crop_dim = g("crop_dim")
        self.pw = PackedWeights(cfg, state, act_dtype, device)
        self.act_dtype = act_dtype
        self.device = device
        self.use_graph = use_graph
        self.fuse_front = fuse_front
        self.n_split = int(os.environ.get("TDEED_SPLIT", n_split))
        # the temporal stage (SGP encoder-decoder + heads) of a split batch runs ONCE over all clips behind the join of the
        # sub-batch trunks: its launches are latency bound and their cost does not depend on the row count at these sizes
        self.merge_tail = True
        # where the sub-batch pipelines join inside the trunk (index into the block list; None = behind the last block)
        self.join_at = None
        self._plans = {}

    # ------------------------------------------------------------------ plan construction
    def _blocks(self, pool, steps, keep, taps, B, x, h, w, blocks, x_kept, out_last=None):
        """Appends the launches of a run of bottlenecks for B clips (N = B*T frames) to `steps`; x (N,h,w,Cin) is the input
        map (owned by `pool` unless x_kept).  out_last: where the last block writes its output (a slice of a buffer shared
        with the plan that continues the trunk) instead of a pool buffer.  Returns (x, h, w, x_kept)."""
        pw, Wt = self.pw, self.pw.W
        T = pw.clip_len
        N = B * T
        dt = self.act_dtype
        es = _esz(dt)
        xs = None           # compact copy of the first Fp channels of x, written by the producer of x (see below)
        q_carry = None      # tap maps of this block's gate-shift site, made in the tail of the bottleneck launch in front of it
        for bi, bw in enumerate(blocks):
            blk = bw.spec
            M = N * h * w
            one_launch = _bneck_fused(bw, h, w, out_last is not None and bw is blocks[-1])
            # conv1 inside the grouped conv's launch (the y1 map never exists): narrow block inputs, bf16
            # (block inputs up to 160 channels = 5 k-steps of conv1 fragments per wave; s4.b1 of RegNetY-200MF, 152 -> 368: 4072 vs
            # 4002 clips/s with it on the same box)
            c1g = bool(C1_GCONV and not one_launch and dt == torch.bfloat16 and bw.w2frag is not None and blk.cin <= C1_GCONV_MAX_CIN
                       and ops.c1_gconv_fits(h, w, blk.cin, blk.cout, blk.stride))
            if c1g and bw.c1g_w1f is None:
                rows_ = 16 * ops.c1_gconv_slab_tiles(h, w, blk.cout, blk.stride)
                bw.c1g_w1f = pack_mfma_frags(bw.w1_raw, self.device, rows=rows_)
            # conv1 (optionally behind the gate-shift splice)
            y1 = None if (one_launch or c1g) else pool.take((N, h, w, blk.cout), dt)
            if blk.gsf_fold:
                F = blk.gsf_fold
                Fp = (F + 7) // 8 * 8
                # the gate-shift launches read only channels [0, Fp): from the compact slice the previous block's conv3 wrote
                # beside its output when there is one (a slice of the channels-last map drags whole cache lines)
                xg = xs if (xs is not None and xs.shape[-1] == Fp) else x
                q_given = q_carry is not None
                gb = dict(gate=pool.take((N, h, w, 2), torch.float32),
                          q=(q_carry if q_given else pool.take((N, h, w, 6), torch.float32)),
                          ysum=pool.take((N, F), torch.float32),
                          xsum=pool.take((N, F), torch.float32))
                q_carry = None
                # the blend itself runs inside the one-launch bottleneck's frame load when it can (tdeed_bneck_gs_fwd)
                blend_in = bool(one_launch and BNECK_BLEND and bw.gs_src and bw.gs_cw1 is not None and h * w >= 14
                                and 2 * Fp <= blk.cin and ("_features." + blk.name + ".gs_out") not in taps)
                if not blend_in:
                    gb["out"] = pool.take((M, Fp), dt)
                if bw.gs_cw1 is not None and not blend_in:
                    gb["fw"] = pool.take((B, F, T), torch.float32)
                steps.append(Step(blk.name + ".gate_shift", "gate_shift", lambda x=xg, bw=bw, gb=gb, F=F, Fp=Fp, go=blend_in, qg=q_given: ops.gate_shift(
                    x, B, T, F, Fp, bw.gs_scale, bw.gs_shift, bw.gs_wq, bw.gs_b3d, bw.gs_cw1, bw.gs_cb1,
                    bw.gs_cw2, bw.gs_cb2, bufs=gb, wqf=bw.gs_wqf, src_order=bw.gs_src, gates_only=go, q_given=qg),
                    M * ((1 if blend_in else 2) * F + (0 if blend_in else Fp)) * es + M * 16, 2 * M * F * 27))
                if not (one_launch or c1g):
                    steps.append(Step(blk.name + ".conv1", bw.w1.kern(M), lambda x=x, bw=bw, gb=gb, Fp=Fp, y1=y1, M=M: bw.w1.run(
                        x, bw.s1, bw.h1, ops.ACT_RELU, A0=gb["out"], k0=Fp, out=y1, M=M),
                        *gemm_cost(M, blk.cin, blk.cout, es)))
                if blk.name and ("_features." + blk.name + ".gs_out") in taps:
                    if bw.gs_src:
                        raise ValueError("the gs_out tap is in module channel order: build the engine with TDEED_GS_SRC_ORDER=0")
                    keep["_features." + blk.name + ".gs_out"] = gb["out"]
                gs_bufs = list(gb.values()) + ([xs] if xs is not None else [])
            else:
                blend_in = False
                if not (one_launch or c1g):
                    steps.append(Step(blk.name + ".conv1", bw.w1.kern(M), lambda x=x, bw=bw, y1=y1, M=M: bw.w1.run(
                        x, bw.s1, bw.h1, ops.ACT_RELU, out=y1, M=M), *gemm_cost(M, blk.cin, blk.cout, es)))
                gs_bufs = []
            if one_launch:
                # conv1 (+ splice) -> conv2 -> SE -> conv3 + shortcut in one launch: only x and the output cross HBM
                out = pool.take((N, h, w, blk.cout), dt)
                nxt = blocks[bi + 1].spec if bi + 1 < len(blocks) else None
                xs_next = (pool.take((N, h, w, (nxt.gsf_fold + 7) // 8 * 8), dt)
                           if (nxt is not None and nxt.gsf_fold and GS_SLICE) else None)
                G = gb["out"] if (blk.gsf_fold and not blend_in) else None
                o2 = xs_next.view(-1, xs_next.shape[-1]) if xs_next is not None else None
                if blend_in:
                    # the next site's tap maps in this launch's tail (its input slice is this block's output rows)
                    qt = None
                    nbw = blocks[bi + 1] if bi + 1 < len(blocks) else None
                    if (BNECK_QTAIL and nbw is not None and nxt.gsf_fold and getattr(nbw, "gs_wqf", None) is not None
                            and xs_next is not None and ops.bneck_qtail_fits(h, w, blk.cout, nxt.gsf_fold)):
                        q_carry = pool.take((N, h, w, 6), torch.float32)
                        qt = (nbw.gs_wpf, nbw.gs_bnq, nxt.gsf_fold, q_carry)
                    run = lambda x=x, xg=xg, bw=bw, gb=gb, out=out, o2=o2, F=F, Fp=Fp, qt=qt: ops.bneck_gs(           # noqa: E731
                        x, xg, gb["gate"], gb["ysum"], gb["xsum"], bw.gs_cw1, bw.gs_cb1, bw.gs_cw2, bw.gs_cb2, T, F, Fp,
                        bw.fused.w1f, bw.s1, bw.h1, bw.fused.w2f, bw.s2, bw.h2, bw.se_mf.w1f, bw.se_b1, bw.se_mf.w2f, bw.se_b2,
                        bw.spec.se_rd, bw.fused.w3f, bw.s3, bw.h3, out=out, out2=o2, w2_tap_major=bw.fused.w2_tap_major, qtail=qt)
                else:
                    run = lambda x=x, bw=bw, G=G, out=out, o2=o2: ops.bneck(                                    # noqa: E731
                        x, bw.fused.w1f, bw.s1, bw.h1, bw.fused.w2f, bw.s2, bw.h2, bw.se_mf.w1f, bw.se_b1, bw.se_mf.w2f, bw.se_b2,
                        bw.spec.se_rd, bw.fused.w3f, bw.s3, bw.h3, G=G, out=out, out2=o2, w2_tap_major=bw.fused.w2_tap_major)
                steps.append(Step(blk.name + ".bneck", "bneck", run,
                                  # x in, out; with the blend: each slice piece's temporal neighbour and the gate maps; the tail's Q
                                  (2 * M * blk.cout + (M * Fp if blend_in else 0)) * es + (8 * M if blend_in else 0)
                                  + (24 * M if (blend_in and qt is not None) else 0)
                                  + (2 * blk.cout * blk.cout + blk.cout * blk.gw * 9) * es,
                                  2 * M * blk.cout * (2 * blk.cout + blk.gw * 9)))
                for t_ in gs_bufs:
                    pool.give(t_)
                xs = xs_next
                if not x_kept and hasattr(x, "_td_raw"):
                    pool.give(x)
                tapname = "_features." + blk.name
                x_kept = tapname in taps
                if x_kept:
                    keep[tapname] = out
                x = out
                continue
            s = blk.stride
            h2, w2 = (h - 1) // s + 1, (w - 1) // s + 1
            M2 = N * h2 * w2
            y2 = pool.take((N, h2, w2, blk.cout), dt)
            parts = ops.gconv3x3_parts(h, w, blk.cout, s, dt) if bw.w2frag is not None else 1
            pooled = pool.take((N, parts, blk.cout), torch.float32)
            gate = pool.take((N, blk.cout), torch.float32)
            if c1g:
                G = gb["out"] if blk.gsf_fold else None
                steps.append(Step(blk.name + ".conv1_conv2", "c1_gconv", lambda x=x, bw=bw, blk=blk, G=G, y2=y2, pooled=pooled: ops.c1_gconv(
                    x, bw.c1g_w1f, bw.s1, bw.h1, bw.w2frag, bw.s2, bw.h2, blk.gw, blk.stride, blk.cout, G=G, out=y2, pooled=pooled),
                    (M * blk.cin + M2 * blk.cout) * es + blk.cout * (blk.cin + blk.gw * 9) * es,
                    2 * M * blk.cin * blk.cout + 2 * M2 * blk.cout * blk.gw * 9))
            else:
                steps.append(Step(blk.name + ".conv2", "gconv3x3", lambda y1=y1, bw=bw, blk=blk, y2=y2, pooled=pooled: ops.gconv3x3(
                    y1, bw.w2, bw.s2, bw.h2, blk.gw, blk.stride, wfrag=bw.w2frag, out=y2, pooled=pooled),
                    (M + M2) * blk.cout * es + blk.cout * blk.gw * 9 * 4, 2 * M2 * blk.cout * blk.gw * 9))
            steps.append(Step(blk.name + ".se", "se_gate", lambda pooled=pooled, bw=bw, gate=gate, ic=1.0 / (h2 * w2): _se(pooled, ic, bw, gate),
                2 * N * blk.cout * 4 + 2 * blk.cout * blk.se_rd * 4, 4 * N * blk.cout * blk.se_rd))
            if blk.has_downsample:
                sc = pool.take((N, h2, w2, blk.cout), dt)
                gather = (s, h, w, h2, w2) if s > 1 else None
                steps.append(Step(blk.name + ".downsample", bw.wd.kern(M2), lambda x=x, bw=bw, sc=sc, gather=gather, M2=M2: bw.wd.run(
                    x, bw.sd, bw.hd, ops.ACT_NONE, gather=gather, out=sc, M=M2),
                    *gemm_cost(M2, blk.cin, blk.cout, es)))
            else:
                sc = x
            out = (out_last if (out_last is not None and bw is blocks[-1]) else pool.take((N, h2, w2, blk.cout), dt))
            nxt = blocks[bi + 1].spec if bi + 1 < len(blocks) else None
            xs_next = None
            if nxt is not None and nxt.gsf_fold and GS_SLICE:
                xs_next = pool.take((N, h2, w2, (nxt.gsf_fold + 7) // 8 * 8), dt)
            steps.append(Step(blk.name + ".conv3", bw.w3.kern(M2), lambda y2=y2, bw=bw, gate=gate, sc=sc, out=out, M2=M2, hw2=h2 * w2, xs_next=xs_next: bw.w3.run(
                y2, bw.s3, bw.h3, ops.ACT_RELU, residual=sc, a_scale=gate, a_scale_rows=hw2, out=out, M=M2, out2=xs_next),
                *gemm_cost(M2, blk.cout, blk.cout, es, True)))
            # liveness: everything but `out` (and the next block's slice) dies here
            for t_ in ([y1] if y1 is not None else []) + [y2, pooled, gate] + gs_bufs + ([sc] if blk.has_downsample else []):
                pool.give(t_)
            xs = xs_next
            if not x_kept and hasattr(x, "_td_raw"):
                pool.give(x)
            tapname = "_features." + blk.name
            x_kept = tapname in taps
            if x_kept:
                keep[tapname] = out
            x, h, w = out, h2, w2
        return x, h, w, x_kept

    def _build_tail(self, B, feat, head_out, trunk_in=None, start=None):
        """The launches behind the sub-batch join, over all B clips: optionally the rest of the trunk (blocks[start:] on the
        map `trunk_in` = (x, h, w) that the sub-batch plans wrote) + avg-pool, then SGP encoder-decoder + heads."""
        pw, Wt = self.pw, self.pw.W
        T, C, dt = pw.clip_len, pw.spec.feat_dim, self.act_dtype
        pool, steps, keep = _Pool(self.device), [], {}
        N = B * T
        if trunk_in is not None:
            x, h, w = trunk_in
            x, h, w, _ = self._blocks(pool, steps, keep, set(), B, x, h, w, list(Wt.blocks[start:]), True)
            frs = torch.empty((N, 2), dtype=torch.float32, device=self.device)
            feat._td_rowstat = frs
            steps.append(Step("avgpool", "avgpool_posenc", lambda x=x: ops.avgpool_posenc(x, B, T, Wt.temp_enc, out=feat,
                                                                                       rowstat=frs),
                              (N * h * w + N) * C * _esz(dt)))
        sb = SgpBuilder(pool, steps, keep, set(), B, dt)
        cur = sb.pyramid(feat, T, pw.n_layers, Wt.sgp, Wt.mixer)
        steps.append(Step("heads", "heads", lambda cur=cur: ops.heads(cur, Wt.head_w, Wt.head_b, out=head_out),
                          N * C * _esz(dt) + N * pw.n_out * 4, 2 * N * C * pw.n_out))
        return SimpleNamespace(steps=steps, pool_bytes=pool.total_bytes(), sgp_out=cur)

    def _build(self, B, H, W, flip, taps, head_out=None, feat_out=None, stop_at=None, trunk_out=None, feat_rs=None,
               frames_dtype=torch.uint8):
        """flip: bool (all frames) or a uint8 device tensor (B*T,) of per-frame flags (the augmented eval path)."""
        pw, Wt = self.pw, self.pw.W
        T = pw.clip_len
        N = B * T
        dt = self.act_dtype
        dev = self.device
        pool = _Pool(dev)
        steps = []
        keep = {}
        crop = None
        ch, cw = H, W
        if self.crop_dim is not None and self.crop_dim > 0 and (self.crop_dim != H or self.crop_dim != W):
            ch = cw = self.crop_dim
            crop = (int(round((H - ch) / 2.0)), int(round((W - cw) / 2.0)), ch, cw)
        frames = torch.empty((N, 3, H, W), dtype=frames_dtype, device=dev)
        Ho, Wo = (ch + 1) // 2, (cw + 1) // 2
        es = _esz(dt)
        blocks = list(Wt.blocks)
        fused_front = (Wt.front is not None and "_features.stem" not in taps and self.fuse_front
                       and frames_dtype == torch.uint8 and not isinstance(flip, torch.Tensor)
                       and ops.s1_front_parts(ch, cw, Wt.blocks[0].spec.cout) > 0)
        if fused_front:
            bw = blocks.pop(0)
            blk = bw.spec
            h2, w2 = (Ho + 1) // 2, (Wo + 1) // 2
            parts = ops.s1_front_parts(ch, cw, blk.cout)
            y2 = pool.take((N, h2, w2, blk.cout), dt)
            sc = pool.take((N, h2, w2, blk.cout), dt)
            pooled = pool.take((N, parts, blk.cout), torch.float32)
            gate = pool.take((N, blk.cout), torch.float32)
            out = pool.take((N, h2, w2, blk.cout), dt)
            M2 = N * h2 * w2
            steps.append(Step("s1_front", "s1_front", lambda y2=y2, sc=sc, pooled=pooled: ops.s1_front(
                frames, Wt.front, crop, flip, y2=y2, shortcut=sc, pooled=pooled),
                              N * 3 * ch * cw + 2 * M2 * blk.cout * es,
                              2 * N * Ho * Wo * 32 * (27 + 2 * blk.cout) // 1 + 2 * M2 * blk.cout * blk.gw * 9))
            steps.append(Step(blk.name + ".se", "se_gate", lambda bw=bw, pooled=pooled, gate=gate, ic=1.0 / (h2 * w2): _se(pooled, ic, bw, gate),
                2 * N * blk.cout * 4 + 2 * blk.cout * blk.se_rd * 4, 4 * N * blk.cout * blk.se_rd))
            steps.append(Step(blk.name + ".conv3", bw.w3.kern(M2), lambda bw=bw, y2=y2, sc=sc, gate=gate, out=out, M2=M2, hw2=h2 * w2: bw.w3.run(
                y2, bw.s3, bw.h3, ops.ACT_RELU, residual=sc, a_scale=gate, a_scale_rows=hw2, out=out, M=M2),
                *gemm_cost(M2, blk.cout, blk.cout, es, True)))
            for t_ in (y2, sc, pooled, gate):
                pool.give(t_)
            x, h, w = out, h2, w2
            x_kept = ("_features." + blk.name) in taps
            if x_kept:
                keep["_features." + blk.name] = out
        else:
            x = pool.take((N, Ho, Wo, 32), dt)
        if not fused_front:
            steps.append(Step("stem", "stem", lambda x=x: ops.stem(frames, Wt.stem_w, Wt.stem_scale, Wt.stem_shift, dt, crop,
                                                                   flip, out=x),
                              N * 3 * ch * cw + N * Ho * Wo * 32 * es, 2 * N * Ho * Wo * 32 * 27))
        if not fused_front:
            h, w = Ho, Wo
            x_kept = "_features.stem" in taps
            if x_kept:
                keep["_features.stem"] = x
        if stop_at is not None:
            blocks = blocks[:max(0, stop_at - (1 if fused_front else 0))]
        if trunk_out is not None and not blocks:
            raise ValueError("join_at must leave at least one un-fused bottleneck in the sub-batch plans")
        x, h, w, x_kept = self._blocks(pool, steps, keep, taps, B, x, h, w, blocks, x_kept, out_last=trunk_out)
        if stop_at is not None:          # trunk head only: the rest of the trunk runs once for all sub-batches (plan.tail)
            return SimpleNamespace(frames=frames, steps=steps, keep=keep, head_out=None, pool_bytes=pool.total_bytes(), B=B, T=T,
                                   h=h, w=w)
        C = pw.spec.feat_dim
        feat = pool.take((B, T, C), sgp_stream_dtype(dt, dev)) if feat_out is None else feat_out
        # LayerNorm statistics of the feature rows for the first SGP block's front kernel (the caller's slice of the shared
        # buffer when the temporal stage runs once for all sub-batches)
        frs = feat_rs if feat_rs is not None else torch.empty((N, 2), dtype=torch.float32, device=dev)
        feat._td_rowstat = frs
        steps.append(Step("avgpool", "avgpool_posenc", lambda x=x, feat=feat, frs=frs: ops.avgpool_posenc(
            x, B, T, Wt.temp_enc, out=feat, rowstat=frs), (N * h * w + N) * C * es))
        keep["feat"] = feat
        if not x_kept:
            pool.give(x)
        if feat_out is not None:         # trunk only: the temporal stage runs once for all sub-batches (plan.tail)
            return SimpleNamespace(frames=frames, steps=steps, keep=keep, head_out=None, pool_bytes=pool.total_bytes(), B=B, T=T)

        # ---------------- SGP encoder-decoder
        sb = SgpBuilder(pool, steps, keep, taps, B, dt)
        cur = sb.pyramid(feat, T, pw.n_layers, Wt.sgp, Wt.mixer)
        keep["sgp_out"] = cur
        if head_out is None:
            head_out = torch.empty((N, pw.n_out), dtype=torch.float32, device=dev)
        steps.append(Step("heads", "heads", lambda cur=cur: ops.heads(cur, Wt.head_w, Wt.head_b, out=head_out),
                          N * C * es + N * pw.n_out * 4, 2 * N * C * pw.n_out))
        return SimpleNamespace(frames=frames, steps=steps, keep=keep, head_out=head_out,
                               pool_bytes=pool.total_bytes(), B=B, T=T)

    def plan(self, B, H, W, flip=False, taps=(), slot=0):
        """Launch plan for a batch geometry.  With n_split > 1 (and no taps) the batch is cut into n_split
        sub-batches of whole clips, each with its own buffers and its own HIP stream: two half-batch
        pipelines in flight keep the CUs busy while the other one sits in a latency-bound small launch."""
        # slot: independent buffer sets / graphs of the same geometry, so that consecutive batches can be in flight together
        key = (B, H, W, bool(flip), tuple(sorted(taps)), slot)
        if key in self._plans:
            return self._plans[key]
        ns = self.n_split if (not taps and self.n_split > 1 and B % self.n_split == 0 and B >= self.n_split) else 1
        if ns == 1:
            sub = self._build(B, H, W, bool(flip), set(taps))
            plan = SimpleNamespace(subs=[sub], streams=[None], head_out=sub.head_out, keep=sub.keep, graph=None, tail=None,
                                   steps=sub.steps, pool_bytes=sub.pool_bytes, B=B, T=sub.T)
        else:
            Bs = B // ns
            T = self.pw.clip_len
            head_out = torch.empty((B * T, self.pw.n_out), dtype=torch.float32, device=self.device)
            tail = None
            if self.merge_tail:
                feat = torch.empty((B, T, self.pw.spec.feat_dim), dtype=sgp_stream_dtype(self.act_dtype, self.device),
                                   device=self.device)
                k = self.join_at
                if k is not None and 0 < k < len(self.pw.W.blocks):
                    # the sub-batches split only the bandwidth-bound head of the trunk (blocks [0, k)); the latency-bound
                    # small maps behind it run once for the whole batch, like the temporal stage
                    blk = self.pw.W.blocks[k - 1].spec
                    ch = self.crop_dim if (self.crop_dim and self.crop_dim > 0) else H
                    cw = self.crop_dim if (self.crop_dim and self.crop_dim > 0) else W
                    hh, ww = (ch + 1) // 2, (cw + 1) // 2
                    for b_ in self.pw.W.blocks[:k]:
                        hh, ww = (hh - 1) // b_.spec.stride + 1, (ww - 1) // b_.spec.stride + 1
                    shared = torch.empty((B * T, hh, ww, blk.cout), dtype=self.act_dtype, device=self.device)
                    subs = [self._build(Bs, H, W, bool(flip), set(), stop_at=k,
                                        trunk_out=shared[i * Bs * T:(i + 1) * Bs * T]) for i in range(ns)]
                    tail = self._build_tail(B, feat, head_out, trunk_in=(shared, hh, ww), start=k)
                else:
                    frs = torch.empty((B * T, 2), dtype=torch.float32, device=self.device)
                    feat._td_rowstat = frs
                    subs = [self._build(Bs, H, W, bool(flip), set(), feat_out=feat[i * Bs:(i + 1) * Bs],
                                        feat_rs=frs[i * Bs * T:(i + 1) * Bs * T]) for i in range(ns)]
                    tail = self._build_tail(B, feat, head_out)
            else:
                subs = [self._build(Bs, H, W, bool(flip), set(), head_out=head_out[i * Bs * T:(i + 1) * Bs * T])
                        for i in range(ns)]
            forks = []
            for _ in range(ns - 1):
                forks.append(new_stream(self.device, avoid=forks))
            plan = SimpleNamespace(subs=subs, streams=[None] + forks,
                                   head_out=head_out, keep=subs[0].keep, graph=None, tail=tail,
                                   steps=[st for sb in subs for st in sb.steps] + (tail.steps if tail else []),
                                   pool_bytes=sum(sb.pool_bytes for sb in subs) + (tail.pool_bytes if tail else 0), B=B, T=T)
        self._plans[key] = plan
        return plan

    def set_frames(self, plan, frames_u8):
        """Copy a (B,T,3,H,W) uint8 batch into the plan's input buffers (one per sub-batch)."""
        B, T = frames_u8.shape[:2]
        Bs = B // len(plan.subs)
        for i, sb in enumerate(plan.subs):
            sb.frames.copy_(frames_u8[i * Bs:(i + 1) * Bs].reshape(Bs * T, *frames_u8.shape[2:]), non_blocking=True)

    # ------------------------------------------------------------------ execution
    def _launch_all(self, plan, main):
        """Issue every sub-batch's launches: sub-batch 0 on `main`, the others forked onto their own streams."""
        if len(plan.subs) == 1:
            for s_ in plan.subs[0].steps:
                s_.fn()
            return
        fork = torch.cuda.Event()
        fork.record(main)
        joins = []
        for sb, st in zip(plan.subs, plan.streams):
            if st is None or st.cuda_stream == main.cuda_stream:   # (the pool handed the launching stream out again)
                for s_ in sb.steps:
                    s_.fn()
            else:
                st.wait_event(fork)
                with torch.cuda.stream(st):
                    for s_ in sb.steps:
                        s_.fn()
                    ev = torch.cuda.Event()
                    ev.record(st)
                    joins.append(ev)
        for ev in joins:
            main.wait_event(ev)
        if plan.tail is not None:
            for s_ in plan.tail.steps:
                s_.fn()

    def _capture(self, st, fn):
        """Capture what fn() launches on stream `st` into a graph executable (fn has run once already: modules loaded,
        arguments validated)."""
        import ctypes
        import gc
        # no cyclic garbage collection between begin and end of the capture: the finaliser of an engine that an earlier
        # caller dropped would otherwise run here at a random allocation and call into the HIP runtime (graph
        # destruction, frees) from the capturing thread -- the graph launched afterwards then crashed the host
        global _CAPTURING
        gc_was = gc.isenabled()
        gc.disable()
        _CAPTURING += 1
        _lib.call("tdeed_graph_begin", st.cuda_stream)
        try:
            fn()
        finally:
            h = ctypes.c_void_p()
            try:
                _lib.call("tdeed_graph_end", st.cuda_stream, ctypes.byref(h))
            finally:
                _CAPTURING -= 1
                if gc_was:
                    gc.enable()
        return h

    def run_plan(self, plan):
        """Launch the plan on the current stream (eager) or replay its HIP graph(s)."""
        st = torch.cuda.current_stream()
        if not self.use_graph:
            self._launch_all(plan, st)
            return
        if st.cuda_stream == 0:
            raise RuntimeError("graph replay needs a non-default stream: wrap the call in torch.cuda.stream(s)")
        if plan.graph is None:
            _drain_dead_graphs()
            self._launch_all(plan, st)         # warm-up launch (module load, validates arguments)
            st.synchronize()
            plan.graph = self._capture(st, lambda: self._launch_all(plan, st))   # forked streams join through the fork event
        _lib.call("tdeed_graph_launch", plan.graph, st.cuda_stream)

    def forward(self, frames_u8, augment_inference=False, taps=(), slot=0):
        """frames: uint8 (B,T,3,H,W) on the GPU.  Returns head_out fp32 (B*T, n_cls [+1]) (a view of the plan's buffer:
        consume it on the launching stream before the same slot runs again)."""
        if frames_u8.dtype != torch.uint8:
            raise TypeError("frames must be uint8 (0..255); callers holding floats convert with .to(torch.uint8)")
        B, T, Cc, H, W = frames_u8.shape
        if T != self.pw.clip_len:
            raise ValueError(f"clip length {T} != clip_len {self.pw.clip_len} (gate-shift needs exact clips)")
        plan = self.plan(B, H, W, augment_inference, taps, slot=slot)
        self.set_frames(plan, frames_u8)
        self.run_plan(plan)
        return plan.head_out, plan

    def forward_augmented(self, frames, flip_frames=None):
        """Eval-mode forward (running-statistics BatchNorm) of frames that the caller has already cropped / augmented:
        frames (B,T,3,h,w) uint8 or fp32 0..255 on the GPU, flip_frames: optional uint8 (B*T,) per-frame h-flip flags.
        What `Impl.forward(inference=False)` runs under .eval() (model/model.py:105-129 with the BatchNorms in eval mode).
        Eager launches of a cached plan (fp32 frames and per-frame flips go through the stand-alone stem kernel)."""
        if frames.dtype not in (torch.uint8, torch.float32):
            raise TypeError("frames must be uint8 or float32 (0..255)")
        B, T, Cc, H, W = frames.shape
        if T != self.pw.clip_len:
            raise ValueError(f"clip length {T} != clip_len {self.pw.clip_len} (gate-shift needs exact clips)")
        key = ("aug", B, H, W, frames.dtype, flip_frames is not None)
        plan = self._plans.get(key)
        if plan is None:
            fbuf = torch.zeros((B * T,), dtype=torch.uint8, device=self.device) if flip_frames is not None else False
            crop_dim, self.crop_dim = self.crop_dim, None           # the caller's window: no centre crop on top of it
            try:
                sub = self._build(B, H, W, fbuf, set(), frames_dtype=frames.dtype)
            finally:
                self.crop_dim = crop_dim
            plan = SimpleNamespace(sub=sub, flip_buf=fbuf if flip_frames is not None else None, graph=None)
            self._plans[key] = plan
        plan.sub.frames.copy_(frames.reshape(B * T, Cc, H, W), non_blocking=True)
        if flip_frames is not None:
            plan.flip_buf.copy_(flip_frames.to(torch.uint8), non_blocking=True)
        for s_ in plan.sub.steps:
            s_.fn()
        return plan.sub.head_out, plan

    def __del__(self):
        # graph executables are not destroyed from a finaliser (it may run at any allocation, e.g. inside another engine's
        # capture, and the graph may still be executing): they are parked and destroyed at the next plan build, after a
        # device synchronisation
        try:
            for p in self._plans.values():
                if p.graph is not None:
                    _DEAD_GRAPHS.append(p.graph)
                    p.graph = None
        except Exception:
            pass
