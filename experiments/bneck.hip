// Whole RegNetY bottleneck in one launch, one frame per workgroup (bf16 throughput mode, stride-1 blocks with
// identity shortcut and a small map: s4.b2..b7 of RegNetY-200MF, 7x7 x 368):
//   conv1 1x1 (+gate-shift splice) + BN + ReLU -> conv2 grouped 3x3 + BN + ReLU -> SE squeeze/excite
//   -> conv3 1x1 (on y2 * gate) + BN + residual + ReLU
// The unfused path moves every intermediate through HBM (x, y1, y2 read/written 7 times: 203 MB per
// block-layer at B=8) in 4 launches; here a frame's activations stay in LDS: region A holds x (with the
// gate-shift columns spliced in) and is later overwritten by y2, region B holds y1.  HBM sees x once
// (+ once more, L2-hot, for the residual) and the output: 58 MB.
// Every contraction is v_mfma_f32_16x16x32_bf16 with the weights as the A operand (D[n][pixel]); a wave
// owns output-channel tiles, so each weight fragment is fetched from L2 by exactly one wave of the block
// (pre-packed, 1-KiB coalesced wave loads, next tile's fragments prefetched during the current MFMAs),
// while the activation fragments are conflict-free ds_read_b128 of the LDS rows (stride C*2+16 B).
#include "common.h"

struct BneckP {
  const bf16_t* x; const bf16_t* G; int Fp;       // block input [N][hw][C]; gate-shift splice [N*hw][Fp] (or null)
  const bf16x8* w1f; const float* s1; const float* h1;        // [NT][KS][64], natural row order
  const bf16x8* w2f; const float* s2; const float* h2;        // [NT4][5][64]  (pack_gconv_frags)
  const bf16_t* se_w1p; const float* se_b1; const bf16_t* se_w2p; const float* se_b2; int R;   // [C][R8], [R][C] bf16
  const bf16x8* w3f; const float* s3; const float* h3;        // [NT][KS][64], natural row order
  bf16_t* out;
  int h, w, C, KS, NT;                             // NT = n-tiles of 16 channels (ceil(C/16))
  long long* dbg;                                  // diagnostic: per-block phase time stamps (or null)
};

#define BN_STAMP(i) do { if (p.dbg && threadIdx.x == 0) p.dbg[(long)blockIdx.x * 8 + (i)] = clock64(); } while (0)

// acc[pt] += W_tile (KS fragments in registers) x activation rows in LDS, for up to 4 pixel tiles.
// k-step outer / pixel tile inner: 4 independent accumulators, LDS fragment reads run two k-steps ahead.
template <int KS>
__device__ __forceinline__ void tile_contract(const bf16x8 (&wc)[KS], const unsigned char* act, const int (&prow)[4],
                                              int q, f32x4 (&acc)[4]) {
  bf16x8 cur[4], nxt[4];
  auto ld = [&](int ks, bf16x8 (&dst)[4]) {
    const int kb = (32 * ks + 8 * q) * 2;               // rows are zero padded to KS*32 channels: always in range
#pragma unroll
    for (int pt = 0; pt < 4; ++pt) dst[pt] = *reinterpret_cast<const bf16x8*>(act + prow[pt] + kb);
  };
  ld(0, cur);
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) {
    if (ks + 1 < KS) ld(ks + 1, nxt);
#pragma unroll
    for (int pt = 0; pt < 4; ++pt) acc[pt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wc[ks], cur[pt], acc[pt], 0, 0, 0);
#pragma unroll
    for (int pt = 0; pt < 4; ++pt) cur[pt] = nxt[pt];
  }
}

template <int KS>
__global__ __launch_bounds__(256) void bneck_kernel(const BneckP p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int hw = p.h * p.w;
  const int RS = KS * 64 + 16;                           // activation row stride (bytes): K padded to 32, + 16 B skew
  unsigned char* At = smem;                              // [hw][RS]  x, later y2
  unsigned char* Bt = smem + hw * RS;                    // [hw + 1][RS]  y1; last row = zeros (taps outside the map)
  float* pooled = reinterpret_cast<float*>(Bt + (hw + 1) * RS);     // [C] sums, [R] hidden, [max(C, nsl*R)] gate/scratch
  float* hid = pooled + p.C;
  float* gate = hid + ((p.R + 3) & ~3);
  const int n = blockIdx.x;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int pl = lane & 15, q = lane >> 4;
  const bf16_t* xn = p.x + (long)n * hw * p.C;
  const int npt = (hw + 15) >> 4;                        // pixel tiles (<= 4)

  BN_STAMP(0);
  // ---- P0: x (with the gate-shift columns spliced in) -> region A, 16-byte chunks, coalesced
  {
    const int cpr = p.C >> 3;
    const bf16_t* gn = p.G ? p.G + (long)n * hw * p.Fp : nullptr;
    const int total = hw * cpr;
    for (int i0 = tid; i0 < total; i0 += 256 * 10) {          // up to 10 independent 16-byte loads in flight per thread
      u32x4 v[10];
#pragma unroll
      for (int b = 0; b < 10; ++b) {
        const int i = i0 + b * 256;
        if (i < total) {
          const int px = i / cpr, ck = i - px * cpr;
          const int k = ck * 8;
          const bf16_t* src = (gn && k < p.Fp) ? gn + (long)px * p.Fp + k : xn + (long)px * p.C + k;
          v[b] = *reinterpret_cast<const u32x4*>(src);
        }
      }
#pragma unroll
      for (int b = 0; b < 10; ++b) {
        const int i = i0 + b * 256;
        if (i < total) {
          const int px = i / cpr, ck = i - px * cpr;
          *reinterpret_cast<u32x4*>(At + (long)px * RS + ck * 16) = v[b];
        }
      }
    }
    // channels C .. KS*32 of every row (both regions) and the extra zero row must read as exact zeros:
    // they meet zero weights, but 0 * stale-NaN = NaN
    const int padc = (RS - p.C * 2) >> 4;                // 16-byte pieces behind the C real channels
    for (int i = tid; i < (2 * hw + 1) * padc; i += 256) {
      const int r = i / padc, j = i - r * padc;
      *reinterpret_cast<u32x4*>(smem + (long)r * RS + p.C * 2 + j * 16) = (u32x4){0u, 0u, 0u, 0u};
    }
    for (int i = tid; i < (p.C * 2) >> 4; i += 256)
      *reinterpret_cast<u32x4*>(Bt + (long)hw * RS + i * 16) = (u32x4){0u, 0u, 0u, 0u};
  }
  __syncthreads();

  BN_STAMP(1);
  // per-lane row offsets of this lane's pixel in each pixel tile (rows beyond hw clamp to row 0)
  int prow[4];
  bool pok[4];
#pragma unroll
  for (int pt = 0; pt < 4; ++pt) {
    const int px = pt * 16 + pl;
    pok[pt] = px < hw && pt < npt;
    prow[pt] = (pok[pt] ? px : 0) * RS;
  }

  // ---- P1: conv1: y1 = relu(bn(W1 x'))  (wave = n-tiles wv, wv+4, ...)
  {
    bf16x8 wc[KS], wn[KS], wn2[KS];                  // current tile + the next TWO tiles' fragments in flight
    int T = wv;
    if (T < p.NT) {
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) wc[ks] = p.w1f[((long)T * KS + ks) * 64 + lane];
    }
    if (T + 4 < p.NT) {
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) wn[ks] = p.w1f[((long)(T + 4) * KS + ks) * 64 + lane];
    }
    for (; T < p.NT; T += 4) {
      const int Tn = T + 8;
      if (Tn < p.NT) {
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) wn2[ks] = p.w1f[((long)Tn * KS + ks) * 64 + lane];
      }
      const int ch0 = T * 16 + 4 * q;
      float sc[4], sh[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const bool ok = ch0 + r < p.C;
        sc[r] = ok ? p.s1[ch0 + r] : 0.f;
        sh[r] = ok ? p.h1[ch0 + r] : 0.f;
      }
      f32x4 acc[4];
#pragma unroll
      for (int pt = 0; pt < 4; ++pt) acc[pt] = (f32x4){0.f, 0.f, 0.f, 0.f};
      tile_contract<KS>(wc, At, prow, q, acc);
#pragma unroll
      for (int pt = 0; pt < 4; ++pt) {
        if (pok[pt] && ch0 < p.C) {
          bf16x4 o;
#pragma unroll
          for (int r = 0; r < 4; ++r) o[r] = (bf16_t)fmaxf(acc[pt][r] * sc[r] + sh[r], 0.f);
          *reinterpret_cast<bf16x4*>(Bt + prow[pt] + ch0 * 2) = o;
        }
      }
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) { wc[ks] = wn[ks]; wn[ks] = wn2[ks]; }
    }
  }
  __syncthreads();
  BN_STAMP(2);

  // ---- P2: conv2 grouped 3x3 from y1 (no halo: taps outside the map give a zero fragment) -> y2 in region A
  {
    // per (pixel tile, k-step): byte offset of the tap pixel's row in region B and validity
    int toff[4][5];
#pragma unroll
    for (int pt = 0; pt < 4; ++pt) {
      const int px = pt * 16 + pl;
      const int pc = (px < hw) ? px : 0;
      const int oy = pc / p.w, ox = pc - oy * p.w;
#pragma unroll
      for (int ks = 0; ks < 5; ++ks) {
        const int sidx = 4 * ks + q;
        const int half = sidx / 9, tap = sidx - half * 9;
        const int dy = tap / 3 - 1, dx = tap - (tap / 3) * 3 - 1;
        const int yy = oy + dy, xx = ox + dx;
        const bool ok = sidx < 18 && yy >= 0 && yy < p.h && xx >= 0 && xx < p.w;
        toff[pt][ks] = (ok ? (yy * p.w + xx) : hw) * RS + (sidx < 18 ? half * 16 : 0);   // row hw = zeros
      }
    }
    const int NU = (p.C + 15) >> 4;
    bf16x8 wf[5], wfn[5];
    float sc[4], sh[4], scn[4], shn[4];
    auto fetch = [&](int U, bf16x8 (&wd)[5], float (&sd)[4], float (&hd)[4]) {
#pragma unroll
      for (int ks = 0; ks < 5; ++ks) wd[ks] = p.w2f[((long)U * 5 + ks) * 64 + lane];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int c = U * 16 + 4 * q + r;
        const bool ok = c < p.C;
        sd[r] = ok ? p.s2[c] : 0.f;
        hd[r] = ok ? p.h2[c] : 0.f;
      }
    };
    if (wv < NU) fetch(wv, wf, sc, sh);
    for (int U = wv; U < NU; U += 4) {
      if (U + 4 < NU) fetch(U + 4, wfn, scn, shn);
      const int ch0 = U * 16 + 4 * q;
      float psum[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int pt = 0; pt < 4; ++pt) {
        if (pt >= npt) break;
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < 5; ++ks) {
          const bf16x8 yf = *reinterpret_cast<const bf16x8*>(Bt + toff[pt][ks] + U * 32);
          acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[ks], yf, acc, 0, 0, 0);
        }
        if (pok[pt] && ch0 < p.C) {
          bf16x4 o;
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            o[r] = (bf16_t)fmaxf(acc[r] * sc[r] + sh[r], 0.f);
            psum[r] += (float)o[r];
          }
          *reinterpret_cast<bf16x4*>(At + prow[pt] + ch0 * 2) = o;
        }
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float v = psum[r];
        v += __shfl_xor(v, 1, 64);
        v += __shfl_xor(v, 2, 64);
        v += __shfl_xor(v, 4, 64);
        v += __shfl_xor(v, 8, 64);
        if (pl == 0 && ch0 + r < p.C) pooled[ch0 + r] = v;
      }
#pragma unroll
      for (int ks = 0; ks < 5; ++ks) wf[ks] = wfn[ks];
#pragma unroll
      for (int r = 0; r < 4; ++r) { sc[r] = scn[r]; sh[r] = shn[r]; }
    }
  }
  __syncthreads();
  BN_STAMP(3);

  // ---- P3: SE excitation on this frame, then y2 *= gate in place.  bf16 weights, 16-byte loads (8 outputs each);
  //          every thread issues its whole share of the weight matrix as ONE batch of independent loads.
  {
    const float inv = 1.0f / (float)hw;
    const int R = p.R, C = p.C;
    const int R8 = (R + 7) & ~7;
    float* part = reinterpret_cast<float*>(Bt);          // scratch [nsl][R8] / [nsl2][C] in the (now dead) y1 region
    {
      const int NJ = R8 >> 3;                            // octets of hidden units
      const int nsl = 256 / NJ;                          // C slices
      const int jo = tid % NJ, sl = tid / NJ;
      float a[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
      if (sl < nsl) {
        const int cper = (C + nsl - 1) / nsl;
        const int c0 = sl * cper, c1 = min(C, c0 + cper);
        constexpr int MAXB = 24;
        for (int cb = c0; cb < c1; cb += MAXB) {
          bf16x8 wv8[MAXB];
#pragma unroll
          for (int i = 0; i < MAXB; ++i)
            if (cb + i < c1) wv8[i] = *reinterpret_cast<const bf16x8*>(p.se_w1p + (long)(cb + i) * R8 + jo * 8);
#pragma unroll
          for (int i = 0; i < MAXB; ++i)
            if (cb + i < c1) {
              const float pv = pooled[cb + i];
#pragma unroll
              for (int e = 0; e < 8; ++e) a[e] = fmaf(pv, (float)wv8[i][e], a[e]);
            }
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) part[sl * R8 + jo * 8 + e] = a[e];
      }
      __syncthreads();
      if (tid < R) {
        float v = 0.f;
        for (int s_ = 0; s_ < nsl; ++s_) v += part[s_ * R8 + tid];
        hid[tid] = fmaxf(v * inv + p.se_b1[tid], 0.f);
      }
      __syncthreads();
    }
    {
      const int NC = C >> 3;                             // octets of channels
      const int nsl = 256 / NC > 0 ? 256 / NC : 1;       // slices of the hidden dimension
      const int co = tid % NC, sl = tid / NC;
      float a[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
      const bool act = sl < nsl && tid < NC * nsl;
      if (act) {
        const int jper = (R + nsl - 1) / nsl;
        const int j0 = sl * jper, j1 = min(R, j0 + jper);
        constexpr int MAXB = 24;
        for (int jb = j0; jb < j1; jb += MAXB) {
          bf16x8 wv8[MAXB];
#pragma unroll
          for (int i = 0; i < MAXB; ++i)
            if (jb + i < j1) wv8[i] = *reinterpret_cast<const bf16x8*>(p.se_w2p + (long)(jb + i) * C + co * 8);
#pragma unroll
          for (int i = 0; i < MAXB; ++i)
            if (jb + i < j1) {
              const float hv = hid[jb + i];
#pragma unroll
              for (int e = 0; e < 8; ++e) a[e] = fmaf(hv, (float)wv8[i][e], a[e]);
            }
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) part[sl * C + co * 8 + e] = a[e];
      }
      __syncthreads();
      for (int c = tid; c < C; c += 256) {
        float g = 0.f;
        for (int s_ = 0; s_ < nsl; ++s_) g += part[s_ * C + c];
        gate[c] = sigmoidf_(g + p.se_b2[c]);
      }
      __syncthreads();
    }
    const int cpr = C >> 3;
    for (int i = tid; i < hw * cpr; i += 256) {
      const int px = i / cpr, ck = i - px * cpr;
      bf16x8* ptr8 = reinterpret_cast<bf16x8*>(At + (long)px * RS + ck * 16);
      bf16x8 v = *ptr8;
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = (bf16_t)((float)v[e] * gate[ck * 8 + e]);
      *ptr8 = v;
    }
  }
  __syncthreads();
  BN_STAMP(4);

  // ---- P4: conv3 on the gated y2, + residual x, ReLU
  {
    bf16x8 wc[KS], wn[KS], wn2[KS];                  // current tile + the next TWO tiles' fragments in flight
    int T = wv;
    if (T < p.NT) {
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) wc[ks] = p.w3f[((long)T * KS + ks) * 64 + lane];
    }
    if (T + 4 < p.NT) {
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) wn[ks] = p.w3f[((long)(T + 4) * KS + ks) * 64 + lane];
    }
    for (; T < p.NT; T += 4) {
      const int Tn = T + 8;
      if (Tn < p.NT) {
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) wn2[ks] = p.w3f[((long)Tn * KS + ks) * 64 + lane];
      }
      const int ch0 = T * 16 + 4 * q;
      const bool cok = ch0 < p.C;
      float sc[4], sh[4];
      u32x2 rres[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const bool ok = ch0 + r < p.C;
        sc[r] = ok ? p.s3[ch0 + r] : 0.f;
        sh[r] = ok ? p.h3[ch0 + r] : 0.f;
      }
#pragma unroll
      for (int pt = 0; pt < 4; ++pt) {
        rres[pt] = (u32x2){0u, 0u};
        if (pok[pt] && cok) rres[pt] = *reinterpret_cast<const u32x2*>(xn + (long)(pt * 16 + pl) * p.C + ch0);
      }
      f32x4 acc[4];
#pragma unroll
      for (int pt = 0; pt < 4; ++pt) acc[pt] = (f32x4){0.f, 0.f, 0.f, 0.f};
      tile_contract<KS>(wc, At, prow, q, acc);
#pragma unroll
      for (int pt = 0; pt < 4; ++pt) {
        if (pok[pt] && cok) {
          const float r0 = __uint_as_float(rres[pt][0] << 16), r1 = __uint_as_float(rres[pt][0] & 0xffff0000u);
          const float r2 = __uint_as_float(rres[pt][1] << 16), r3 = __uint_as_float(rres[pt][1] & 0xffff0000u);
          bf16x4 o;
          o[0] = (bf16_t)fmaxf(acc[pt][0] * sc[0] + sh[0] + r0, 0.f);
          o[1] = (bf16_t)fmaxf(acc[pt][1] * sc[1] + sh[1] + r1, 0.f);
          o[2] = (bf16_t)fmaxf(acc[pt][2] * sc[2] + sh[2] + r2, 0.f);
          o[3] = (bf16_t)fmaxf(acc[pt][3] * sc[3] + sh[3] + r3, 0.f);
          *reinterpret_cast<bf16x4*>(p.out + ((long)n * hw + pt * 16 + pl) * p.C + ch0) = o;
        }
      }
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) { wc[ks] = wn[ks]; wn[ks] = wn2[ks]; }
    }
  }
  BN_STAMP(5);
}

static size_t bneck_rs(int C) { return (size_t)((C + 31) / 32) * 64 + 16; }
static size_t bneck_smem(int h, int w, int C, int R) {
  return (2 * (size_t)h * w + 1) * bneck_rs(C) + ((size_t)2 * C + ((R + 3) & ~3)) * sizeof(float);
}
static bool bneck_scratch_ok(int h, int w, int C, int R) {        // SE partial sums live in the y1 region
  const int R8 = (R + 7) & ~7;
  const size_t s1 = (size_t)(256 / (R8 / 8)) * R8, s2 = (size_t)(256 / (C / 8) > 0 ? 256 / (C / 8) : 1) * C;
  return (s1 > s2 ? s1 : s2) * sizeof(float) <= (size_t)h * w * bneck_rs(C);
}

static long long* g_bneck_dbg = nullptr;
extern "C" int tdeed_bneck_set_debug(void* buf) { g_bneck_dbg = (long long*)buf; return TDEED_OK; }

extern "C" int tdeed_bneck_fits(int h, int w, int C, int R) {
  const int KS = (C + 31) / 32;
  if (h * w > 64 || C % 8 != 0 || R > 256 || C > 512) return 0;
  if (!(KS == 5 || KS == 10 || KS == 12)) return 0;
  return (bneck_smem(h, w, C, R) <= 80 * 1024 && bneck_scratch_ok(h, w, C, R)) ? 1 : 0;
}

extern "C" int tdeed_bneck_fwd(const void* x, const void* G, int Fp, int N, int h, int w, int C,
                               const void* w1f, const float* s1, const float* h1, const void* w2f,
                               const float* s2, const float* h2, const void* se_w1p, const float* se_b1,
                               const void* se_w2p, const float* se_b2, int R, const void* w3f, const float* s3,
                               const float* h3, void* out, void* stream) {
  TD_CHECK(x && w1f && s1 && h1 && w2f && s2 && h2 && se_w1p && se_b1 && se_w2p && se_b2 && w3f && s3 && h3 && out,
           "bneck: null pointer");
  TD_CHECK(N > 0 && tdeed_bneck_fits(h, w, C, R), "bneck: geometry h=%d w=%d C=%d R=%d unsupported", h, w, C, R);
  TD_CHECK(!G || (Fp % 8 == 0 && Fp > 0 && Fp <= C), "bneck: bad splice width %d", Fp);
  BneckP p;
  p.x = (const bf16_t*)x; p.G = (const bf16_t*)G; p.Fp = G ? Fp : 0;
  p.w1f = (const bf16x8*)w1f; p.s1 = s1; p.h1 = h1;
  p.w2f = (const bf16x8*)w2f; p.s2 = s2; p.h2 = h2;
  p.se_w1p = (const bf16_t*)se_w1p; p.se_b1 = se_b1; p.se_w2p = (const bf16_t*)se_w2p; p.se_b2 = se_b2; p.R = R;
  p.w3f = (const bf16x8*)w3f; p.s3 = s3; p.h3 = h3;
  p.out = (bf16_t*)out;
  p.h = h; p.w = w; p.C = C; p.KS = (C + 31) / 32; p.NT = (C + 15) / 16;
  p.dbg = g_bneck_dbg;
  const size_t smem = bneck_smem(h, w, C, R);
  hipStream_t st = (hipStream_t)stream;
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute((const void*)bneck_kernel<5>, hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024);
    if (e == hipSuccess) e = hipFuncSetAttribute((const void*)bneck_kernel<10>, hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024);
    if (e == hipSuccess) e = hipFuncSetAttribute((const void*)bneck_kernel<12>, hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024);
    if (e != hipSuccess) { tdeed_set_error("bneck: hipFuncSetAttribute: %s", hipGetErrorString(e)); return TDEED_ERR_RUNTIME; }
    attr_set = true;
  }
  switch (p.KS) {
    case 5: hipLaunchKernelGGL(bneck_kernel<5>, dim3(N), dim3(256), smem, st, p); break;
    case 10: hipLaunchKernelGGL(bneck_kernel<10>, dim3(N), dim3(256), smem, st, p); break;
    case 12: hipLaunchKernelGGL(bneck_kernel<12>, dim3(N), dim3(256), smem, st, p); break;
    default: tdeed_set_error("bneck: KS=%d", p.KS); return TDEED_ERR_ARG;
  }
  TD_LAUNCH_CHECK("bneck");
  return TDEED_OK;
}
