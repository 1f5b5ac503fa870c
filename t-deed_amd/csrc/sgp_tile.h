// Shared LDS-tile helpers of the SGP depthwise-branch kernels (forward: sgp.hip, backward: sgp_bwd.hip).
#pragma once
#include "common.h"

// =========================================================================== depthwise branch helpers
// A block owns CH=16 channels of one clip: tile[(T + 2*halo)][16] fp32 in LDS with zero halo rows,
// per-channel weights transposed to wl[tap][16].  Thread (tl = tid>>4, c = tid&15).
#define SGP_CH 16

// load rows [0,T) x 16 channels (row stride ld) into tile rows [halo, halo+T); zero the halos.
template <typename T>
__device__ __forceinline__ void load_tile(const T* __restrict__ src, long ld, int T_len, int c0, int C,
                                          float* tile, int halo) {
  constexpr int EPC = Chunk<T>::N;
  constexpr int CPR = SGP_CH / EPC;
  for (int i = threadIdx.x; i < 2 * halo * SGP_CH; i += 256) {
    int r = i / SGP_CH, c = i - r * SGP_CH;
    int row = r < halo ? r : (T_len + r);
    tile[row * SGP_CH + c] = 0.f;
  }
  for (int i = threadIdx.x; i < T_len * CPR; i += 256) {
    int t = i / CPR, ck = i - t * CPR;
    float v[EPC];
    if (c0 + ck * EPC < C) {
      Chunk<T>::load(src + (long)t * ld + c0 + ck * EPC, v);
    } else {
#pragma unroll
      for (int e = 0; e < EPC; ++e) v[e] = 0.f;
    }
#pragma unroll
    for (int e = 0; e < EPC; ++e) tile[(halo + t) * SGP_CH + ck * EPC + e] = v[e];
  }
}

// store a [T][16] fp32 LDS tile to dst rows (stride ld), optionally adding `add` (same geometry as dst)
template <typename T>
__device__ __forceinline__ void store_tile(const float* res, T* __restrict__ dst, long ld, int t_begin, int t_end,
                                           int c0, int C, const T* __restrict__ add, long ld_add) {
  constexpr int EPC = Chunk<T>::N;
  constexpr int CPR = SGP_CH / EPC;
  for (int i = threadIdx.x; i < (t_end - t_begin) * CPR; i += 256) {
    int t = t_begin + i / CPR, ck = i % CPR;
    if (c0 + ck * EPC >= C) continue;
    float v[EPC];
#pragma unroll
    for (int e = 0; e < EPC; ++e) v[e] = res[t * SGP_CH + ck * EPC + e];
    if (add) {
      float a[EPC];
      Chunk<T>::load(add + (long)t * ld_add + c0 + ck * EPC, a);
#pragma unroll
      for (int e = 0; e < EPC; ++e) v[e] += a[e];
    }
    Chunk<T>::store(dst + (long)t * ld + c0 + ck * EPC, v);
  }
}

// per-channel weights: dw[c][psi ks | convw ks | convkw up | fc | gfc] -> wl[tap][16]
__device__ __forceinline__ void load_dw(const float* __restrict__ dw, int wlen, int c0, int C, float* wl) {
  for (int i = threadIdx.x; i < wlen * SGP_CH; i += 256) {
    int c = i / wlen, k = i - c * wlen;
    wl[k * SGP_CH + c] = (c0 + c < C) ? dw[(long)(c0 + c) * wlen + k] : 0.f;
  }
}

// ---- two-phase staging: every global load of a kernel is ISSUED (branch-free, clamped addresses) before anything
// is written to LDS, so tiles, weights and biases share one memory round trip instead of one each.
constexpr int SGP_TI = 4;       // tile items per lane: T * (16 / EPC) <= 1024
constexpr int SGP_WI = 5;       // weight items per lane: wlen * 16 <= 1280

template <typename T>
__device__ __forceinline__ void tile_issue(const T* __restrict__ src, long ld, int T_len, int c0, int C,
                                           float (&v)[SGP_TI][Chunk<T>::N]) {
  constexpr int EPC = Chunk<T>::N;
  constexpr int CPR = SGP_CH / EPC;
  const int n = T_len * CPR;
#pragma unroll
  for (int u = 0; u < SGP_TI; ++u) {
    const int i = min((int)threadIdx.x + u * 256, n - 1);
    const int t = i / CPR, ck = i - t * CPR;
    const int cc = min(c0 + ck * EPC, C - EPC);                 // channel chunks past C are zeroed at commit
    Chunk<T>::load(src + (long)t * ld + cc, v[u]);
  }
}

template <typename T>
__device__ __forceinline__ void tile_commit(const float (&v)[SGP_TI][Chunk<T>::N], int T_len, int c0, int C,
                                            float* tile, int halo, float* raw = nullptr) {
  constexpr int EPC = Chunk<T>::N;
  constexpr int CPR = SGP_CH / EPC;
  const int n = T_len * CPR;
  for (int i = threadIdx.x; i < 2 * halo * SGP_CH; i += 256) {
    int r = i / SGP_CH, c = i - r * SGP_CH;
    int row = r < halo ? r : (T_len + r);
    tile[row * SGP_CH + c] = 0.f;
  }
#pragma unroll
  for (int u = 0; u < SGP_TI; ++u) {
    const int i = threadIdx.x + u * 256;
    if (i < n) {
      const int t = i / CPR, ck = i - t * CPR;
      const bool ok = c0 + ck * EPC < C;
#pragma unroll
      for (int e = 0; e < EPC; ++e) tile[(halo + t) * SGP_CH + ck * EPC + e] = ok ? v[u][e] : 0.f;
      if (raw) {
#pragma unroll
        for (int e = 0; e < EPC; ++e) raw[t * SGP_CH + ck * EPC + e] = ok ? v[u][e] : 0.f;
      }
    }
  }
}

// mean over T of the tile per channel from the per-row-lane partial sums red[tl][c] (16 lanes): every thread folds the 16
// partials of its channel itself, in a fixed order (no designated threads, no second barrier)
__device__ __forceinline__ float tile_mean_fold(const float* red, int c, int T_len) {
  float a = 0.f;
#pragma unroll
  for (int i = 0; i < 16; ++i) a += red[i * SGP_CH + c];
  return a / (float)T_len;
}

__device__ __forceinline__ void dw_issue(const float* __restrict__ dw, int wlen, int c0, int C, float (&w)[SGP_WI]) {
  const int n = wlen * SGP_CH;
#pragma unroll
  for (int u = 0; u < SGP_WI; ++u) {
    const int i = min((int)threadIdx.x + u * 256, n - 1);
    const int c = i / wlen, k = i - c * wlen;
    w[u] = dw[(long)min(c0 + c, C - 1) * wlen + k];
  }
}

__device__ __forceinline__ void dw_commit(const float (&w)[SGP_WI], int wlen, int c0, int C, float* wl) {
  const int n = wlen * SGP_CH;
#pragma unroll
  for (int u = 0; u < SGP_WI; ++u) {
    const int i = threadIdx.x + u * 256;
    if (i < n) {
      const int c = i / wlen, k = i - c * wlen;
      wl[k * SGP_CH + c] = (c0 + c < C) ? w[u] : 0.f;
    }
  }
}

struct Bias5 { float psi, cw, ckw, fc, g; };
__device__ __forceinline__ Bias5 bias_issue(const float* __restrict__ bias5, long bstride, int cglob, int C) {
  const int cc = min(cglob, C - 1);
  Bias5 b;
  b.psi = bias5[cc]; b.cw = bias5[bstride + cc]; b.ckw = bias5[2 * bstride + cc];
  b.fc = bias5[3 * bstride + cc]; b.g = bias5[4 * bstride + cc];
  return b;
}

// mean over T of the tile per channel; result broadcast through red[16]
__device__ __forceinline__ void tile_mean(const float* tile, int T_len, int halo, float* red /*[17][16]*/) {
  const int c = threadIdx.x & 15, tl = threadIdx.x >> 4;
  float s = 0.f;
  for (int t = tl; t < T_len; t += 16) s += tile[(halo + t) * SGP_CH + c];
  red[tl * SGP_CH + c] = s;
  __syncthreads();
  if (threadIdx.x < SGP_CH) {
    float a = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) a += red[i * SGP_CH + threadIdx.x];
    red[16 * SGP_CH + threadIdx.x] = a / (float)T_len;
  }
  __syncthreads();
}

struct BranchOut { float conv_gate; float inst; };   // (convw+convkw)*psi ,  fc*phi

__device__ __forceinline__ BranchOut branch_eval(const float* tile, const float* wl, const Bias5& bb, bool cok,
                                                 int t, int c, int halo, int ks, int up, float mean_c) {
  // bb: biases of psi, convw, convkw, fc, gfc for this lane's channel
  const float b_psi = cok ? bb.psi : 0.f, b_cw = cok ? bb.cw : 0.f, b_ckw = cok ? bb.ckw : 0.f,
              b_fc = cok ? bb.fc : 0.f, b_g = cok ? bb.g : 0.f;
  const float* col = tile + (halo + t) * SGP_CH + c;
  float psi = b_psi, cw = b_cw, ckw = b_ckw;
  const int hk = ks >> 1, hu = up >> 1;
  for (int k = 0; k < ks; ++k) {
    const float v = col[(k - hk) * SGP_CH];
    psi = fmaf(wl[k * SGP_CH + c], v, psi);
    cw = fmaf(wl[(ks + k) * SGP_CH + c], v, cw);
  }
  for (int k = 0; k < up; ++k) ckw = fmaf(wl[(2 * ks + k) * SGP_CH + c], col[(k - hu) * SGP_CH], ckw);
  const float o = col[0];
  const float fc = fmaf(wl[(2 * ks + up) * SGP_CH + c], o, b_fc);
  const float phi = fmaxf(fmaf(wl[(2 * ks + up + 1) * SGP_CH + c], mean_c, b_g), 0.f);
  BranchOut r;
  r.conv_gate = (cw + ckw) * psi;
  r.inst = fc * phi;
  return r;
}



// The same branches for a RUN of consecutive rows per thread, fully unrolled for a compile-time (KS, UP): thread (c, g)
// takes rows [r*RUN, r*RUN + RUN) for r = g, g + 16, ...; its channel's taps live in registers and the UP + RUN - 1 window
// values of a run are read from the tile ONCE (branch_eval re-reads 2 x (2 KS + UP) LDS words per output: the front
// kernels were bound by those reads).  Same arithmetic order per output as branch_eval: bit-identical results.
template <int KS, int UP, typename F>
__device__ __forceinline__ void branch_runs(const float* tile, const float* wl, const Bias5& bb, bool cok, int c, int g,
                                            int T_len, int halo, float mean_c, F&& emit) {
  constexpr int RUN = 7, HK = KS >> 1, HU = UP >> 1, WN = UP + RUN - 1;
  float wpsi[KS], wcw[KS], wckw[UP];
#pragma unroll
  for (int k = 0; k < KS; ++k) {
    wpsi[k] = wl[k * SGP_CH + c];
    wcw[k] = wl[(KS + k) * SGP_CH + c];
  }
#pragma unroll
  for (int k = 0; k < UP; ++k) wckw[k] = wl[(2 * KS + k) * SGP_CH + c];
  const float wfc = wl[(2 * KS + UP) * SGP_CH + c], wg = wl[(2 * KS + UP + 1) * SGP_CH + c];
  const float b_psi = cok ? bb.psi : 0.f, b_cw = cok ? bb.cw : 0.f, b_ckw = cok ? bb.ckw : 0.f,
              b_fc = cok ? bb.fc : 0.f, b_g = cok ? bb.g : 0.f;
  const float phi = fmaxf(fmaf(wg, mean_c, b_g), 0.f);
  const int last = T_len + 2 * halo - 1;                           // last row of the tile (halo rows are zero)
  for (int t0 = g * RUN; t0 < T_len; t0 += 16 * RUN) {
    float v[WN];
    // window rows t0 - HU .. t0 + RUN - 1 + HU of the sequence = tile rows halo - HU + t0 + j (halo >= HU)
#pragma unroll
    for (int j = 0; j < WN; ++j) v[j] = tile[min(halo - HU + t0 + j, last) * SGP_CH + c];
#pragma unroll
    for (int i = 0; i < RUN; ++i) {
      if (t0 + i < T_len) {
        float psi = b_psi, cw = b_cw, ckw = b_ckw;
#pragma unroll
        for (int k = 0; k < KS; ++k) {
          const float x = v[i + HU - HK + k];
          psi = fmaf(wpsi[k], x, psi);
          cw = fmaf(wcw[k], x, cw);
        }
#pragma unroll
        for (int k = 0; k < UP; ++k) ckw = fmaf(wckw[k], v[i + k], ckw);
        BranchOut r;
        r.conv_gate = (cw + ckw) * psi;
        r.inst = fmaf(wfc, v[i + HU], b_fc) * phi;
        emit(t0 + i, r, v[i + HU]);
      }
    }
  }
}

// dispatch over the (kernel size, long-kernel size) pairs of the shipped configs (sgp_ks 5/7/9/11 x sgp_r 4; modules.py
// 119-126: up = round((ks + 1) r) made odd); other pairs take the generic loop
#define SGP_BRANCH_DISPATCH(ks, up, CALL_FAST, CALL_GENERIC)            \
  if ((ks) == 7 && (up) == 33) { CALL_FAST(7, 33) }                      \
  else if ((ks) == 9 && (up) == 41) { CALL_FAST(9, 41) }                 \
  else if ((ks) == 11 && (up) == 49) { CALL_FAST(11, 49) }               \
  else if ((ks) == 5 && (up) == 25) { CALL_FAST(5, 25) }                 \
  else { CALL_GENERIC }
