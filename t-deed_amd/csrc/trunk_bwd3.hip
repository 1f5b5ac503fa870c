// conv1 backward of a NARROW training bottleneck in ONE launch (bf16; RegNetY-800MF s1.b1 64 <- 32, s2.b1 128 <- 64,
// s2.b2/b3 128 <- 128 channels; with fa = 0, fb = 1 (no ReLU between the BatchNorm and the conv), no shortcut gradient and no
// sink the same launch is conv3's backward of those blocks: 64 <- 64, 128 <- 128; timm Bottleneck.conv1 = ConvNormAct under autograd, /root/reference/model/model.py:265-324):
//
//   d_y1 (conv2's input gradient), z1 (conv1's raw output)            -- read ONCE
//     -> dz1 = k1 * g + k2 * z1 + k3,  g = d_y1 * [fa z1 + fb > 0]     (BatchNorm + ReLU backward, statistics given)
//     -> dx  = dz1 @ W1  (+ shortcut gradient)  masked by [x > 0], with the column sums the BatchNorm backward of the block in
//             front needs ("gradient sink", trunk_bwd2.hip / tdeed_gemm_dgrad)
//     -> dW1 = dz1^T @ x                                               (per-workgroup partials, folded by the write-out)
//
// The pass-per-op chain writes dz1 (a map at the block's INPUT resolution: 2.6 GB at s1.b1 of cfg3) and reads it twice more
// (input-gradient contraction, weight-gradient contraction); here it lives in LDS for one 64-row tile.  W1 (<= 32 KB) sits in
// LDS, read as MFMA A-operand fragments, the weight gradient (<= 128 x 128) in the accumulators of a persistent workgroup.  Both contractions read the same two LDS tiles: the input gradient with plain 16-byte fragment reads
// (k = output channel), the weight gradient with transposing reads (k = row), like wgrad_tr128_kernel.
#include "common.h"

struct NbwP {
  const bf16_t* dY; const bf16_t* Z;                  // [M][CO]
  const float* fa; const float* fb;                   // ReLU mask of conv1's BatchNorm: fa z + fb > 0
  const float* mean; const float* rstd; const float* w; const float* sums;   // its statistics: sums[0][c] = sum g, sums[1][c] = sum g xhat
  const bf16_t* X;                                    // [M][CI] conv1's input (the ReLU output of the block in front: also the sink's mask)
  const bf16_t* Wt;                                   // conv1's weight transposed, [CI][CO] row-major
  const bf16_t* R; long ldr; int r_hi, r_wi;          // shortcut gradient: [M][CI], or rows for the even pixels of an r_hi x r_wi frame
  bf16_t* dX;                                         // [M][CI]
  int use_mask;
  const bf16_t* bz; const float* bmean; const bf16_t* bzd; const float* bmean_d;   // sink statistics operands, [M][CI]
  float* bpart;                                       // [gridDim.x][3][CI]
  float* wpart;                                       // [gridDim.x][CO][CI]
  long M; float inv_M;
};

// waves WN x WK over (output channels, input channels) of the weight gradient; each wave NTW x KTW accumulator tiles
template <int CO, int CI> struct NbwGeom;
template <> struct NbwGeom<64, 32> { static constexpr int WN = 4, WK = 1; };
template <> struct NbwGeom<64, 64> { static constexpr int WN = 2, WK = 2; };
template <> struct NbwGeom<128, 64> { static constexpr int WN = 2, WK = 2; };
template <> struct NbwGeom<128, 128> { static constexpr int WN = 2, WK = 2; };

// RC (Z == NULL at the C ABI): the raw conv output z is not read but RECOMPUTED from the x tile and the weight that are in LDS
// anyway (z = x @ W^T, CI / 32 MFMA k-steps per 16 x 16 tile, rounded to bf16 like the stored map): a quarter of the bytes
// of a launch (2.6 GB at s1.b1 of cfg3) for CO * CI / 64 MFMAs per 64-row tile.  Not at 128 x 128 (MFMA-bound there).
template <int CO, int CI, bool RC = false>
__global__ __launch_bounds__(256, 2) void narrow_conv1_bwd_kernel(const NbwP p) {
  typedef NbwGeom<CO, CI> G;
  constexpr int WN = G::WN, WK = G::WK;
  constexpr int NTW = CO / WN / 16, KTW = CI / WK / 16;
  constexpr int RSY = CO + 16, RSX = CI + 16;                     // LDS row strides (elements): + 32 bytes
  constexpr int KS = CO / 32;                                     // k-steps of the input-gradient contraction
  // its output goes in PAIRS of 16-channel tiles: a lane holds 8 consecutive input channels of one row (16-byte epilogue
  // accesses, 64 bytes per row and wave instruction -- with 32-byte row segments the same bytes move 1.5x slower,
  // tools/ubench/access_shape.hip); wave w serves pair w % PAIRS on the 16-row tiles w / PAIRS + WPP * j
  constexpr int PAIRS = CI / 32, WPP = 4 / PAIRS, MPW = PAIRS;
  constexpr int CPY = CO / 8, CPX = CI / 8;                       // 16-byte pieces per row
  constexpr int NPY = 64 * CPY / 256, NPX = (64 * CPX + 255) / 256;      // pieces per thread per tile
  __shared__ __attribute__((aligned(16))) bf16_t sY[64 * RSY];    // dz1 tile [row][output channel]
  __shared__ __attribute__((aligned(16))) bf16_t sX[64 * RSX];    // x tile   [row][input channel]
  __shared__ __attribute__((aligned(16))) float kt_[5 * CO];      // k1 | k2 | k3 | fa | fb per output channel
  __shared__ __attribute__((aligned(16))) float bt_[2 * CI];      // sink means per input channel
  __shared__ __attribute__((aligned(16))) bf16_t sW[CI * (CO + 8)];      // Wt [ci][co] (row stride + 16 bytes)
  const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6), pl = lane & 15, q = lane >> 4;
  const int g4 = lane >> 4, q4 = (lane >> 2) & 3, p4 = lane & 3;
  const int wn = wv / WK, wk = wv % WK;

  // ---- per-thread constants of the staging pass: a thread serves ONE 8-channel chunk of dY / Z (CPY divides 256)
  const int cy = (tid % CPY) * 8, ry = tid / CPY;                 // rows ry + (256 / CPY) * j
  // per-channel constants and the weight live in LDS (as registers they cost the second wave per SIMD at 128 channels)
  for (int c = tid; c < CO; c += 256) {
    const float rs = p.rstd[c], mu = p.mean[c], s1 = p.sums[c] * p.inv_M, s2 = p.sums[CO + c] * p.inv_M;
    const float k1 = p.w[c] * rs;
    kt_[c] = k1;
    kt_[CO + c] = -k1 * rs * s2;
    kt_[2 * CO + c] = k1 * (mu * rs * s2 - s1);
    kt_[3 * CO + c] = p.fa[c];
    kt_[4 * CO + c] = p.fb[c];
  }
  for (int c = tid; c < CI; c += 256) {
    bt_[c] = p.bpart ? p.bmean[c] : 0.f;
    bt_[CI + c] = (p.bpart && p.bzd) ? p.bmean_d[c] : 0.f;
  }
  for (int i = tid; i < CI * CO / 8; i += 256) {
    const int r = i / CPY, ck = i - r * CPY;
    *reinterpret_cast<u32x4*>(sW + r * (CO + 8) + ck * 8) = *reinterpret_cast<const u32x4*>(p.Wt + (long)r * CO + ck * 8);
  }
  float ss1[8], ss2[8], ss3[8];
#pragma unroll
  for (int r = 0; r < 8; ++r) ss1[r] = ss2[r] = ss3[r] = 0.f;
  const int pr = wv % PAIRS, mt0 = wv / PAIRS;                    // this wave's channel pair and first 16-row tile
  const int c0 = pr * 32 + 8 * q;                                 // this lane's 8 input channels
  f32x4 acc[NTW][KTW];
#pragma unroll
  for (int nt = 0; nt < NTW; ++nt)
#pragma unroll
    for (int kt = 0; kt < KTW; ++kt) acc[nt][kt] = (f32x4){0.f, 0.f, 0.f, 0.f};

  // RC: A-operand fragments of W (rows = output channels, k = input channels) for the z contraction, gathered once from the
  // transposed copy in LDS: lane (i = pl, k-chunk q) of tile T, k-step ks holds W[T * 16 + pl][32 ks + 8 q + (0..7)]
  constexpr int NZT = CO / 64, KS1 = CI / 32;
  bf16x8 zf[RC ? NZT : 1][RC ? KS1 : 1];
  if constexpr (RC) {
    __syncthreads();
#pragma unroll
    for (int zt = 0; zt < NZT; ++zt)
#pragma unroll
      for (int ks = 0; ks < KS1; ++ks)
#pragma unroll
        for (int e = 0; e < 8; ++e) zf[zt][ks][e] = sW[(32 * ks + 8 * q + e) * (CO + 8) + (wv + 4 * zt) * 16 + pl];
  }
  const long ntiles = (p.M + 63) / 64;
  u32x4 vy[NPY], vz[RC ? 1 : NPY], vx[NPX];
  auto issue = [&](long t) {
    const long m0 = t * 64;
#pragma unroll
    for (int j = 0; j < NPY; ++j) {
      const long row = min(m0 + ry + (256 / CPY) * j, p.M - 1);
      vy[j] = *reinterpret_cast<const u32x4*>(p.dY + row * CO + cy);
      if constexpr (!RC) vz[j] = *reinterpret_cast<const u32x4*>(p.Z + row * CO + cy);
    }
#pragma unroll
    for (int j = 0; j < NPX; ++j) {
      const int i = min(tid + 256 * j, 64 * CPX - 1);
      const long row = min(m0 + i / CPX, p.M - 1);
      vx[j] = *reinterpret_cast<const u32x4*>(p.X + row * CI + (i % CPX) * 8);
    }
  };
  // the input gradient's epilogue operands (shortcut gradient, sink statistics maps: 8 bytes per lane and 16-row tile) of a
  // tile are requested while the tile BEFORE it runs its weight-gradient phase: asked for inside the epilogue itself they cost
  // one exposed memory round trip per 16-row tile (3.1 ms instead of ~2 at s1.b1 of cfg3)
  // (not at 128 x 128: the 48 registers of the prefetch would cost the second workgroup per CU, measured slower)
  constexpr bool PRE = !(CO == 128 && CI == 128);
  u32x4 er[PRE ? MPW : 1], ez[PRE ? MPW : 1], ezd[PRE ? MPW : 1];
  unsigned rmask = 0u;                                            // bit j: the row of tile j has a shortcut-gradient row
  auto res_row = [&](long mc, bool& has_r) -> long {
    has_r = p.R != nullptr;
    if (p.R && p.r_hi > 0) {
      const long per = (long)p.r_hi * p.r_wi;
      const long f = mc / per;
      const int rem = (int)(mc - f * per);
      const int yy = rem / p.r_wi, xx = rem - yy * p.r_wi;
      has_r = !((yy | xx) & 1);
      return has_r ? (f * ((p.r_hi + 1) >> 1) + (yy >> 1)) * ((p.r_wi + 1) >> 1) + (xx >> 1) : 0;
    }
    return mc;
  };
  auto issue_epi = [&](long tq) {
    if constexpr (!PRE) return;
    rmask = 0u;
#pragma unroll
    for (int j = 0; j < (PRE ? MPW : 1); ++j) {
      const long mc = min(tq * 64 + (mt0 + WPP * j) * 16 + pl, p.M - 1);
      bool has_r;
      const long rm = res_row(mc, has_r);
      if (has_r) rmask |= 1u << j;
      if (p.R) er[j] = *reinterpret_cast<const u32x4*>(p.R + rm * p.ldr + c0);
      if (p.bpart) {
        ez[j] = *reinterpret_cast<const u32x4*>(p.bz + mc * CI + c0);
        if (p.bzd) ezd[j] = *reinterpret_cast<const u32x4*>(p.bzd + mc * CI + c0);
      }
    }
  };
  long t = blockIdx.x;
  if (t < ntiles) {
    issue(t);
    issue_epi(t);
  }
  for (; t < ntiles; t += gridDim.x) {
    const long m0 = t * 64;
    __syncthreads();                                              // the previous tile's LDS reads are over
    auto store_x = [&]() {
#pragma unroll
      for (int j = 0; j < NPX; ++j) {
        const int i = tid + 256 * j;
        if (i < 64 * CPX) {
          const int r = i / CPX;
          const bool rok = m0 + r < p.M;
          *reinterpret_cast<u32x4*>(sX + r * RSX + (i % CPX) * 8) = rok ? vx[j] : (u32x4){0u, 0u, 0u, 0u};
        }
      }
    };
    if constexpr (RC) {
      // ---- z = x @ W^T of this tile -> sY (bf16, like the stored map); wave w: output-channel tiles w, w + 4, ...
      store_x();
      __syncthreads();
#pragma unroll
      for (int zt = 0; zt < NZT; ++zt) {
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) {
          f32x4 a = {0.f, 0.f, 0.f, 0.f};
          const bf16_t* xr = sX + (mt * 16 + pl) * RSX + 8 * q;
#pragma unroll
          for (int ks = 0; ks < KS1; ++ks)
            a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(zf[zt][ks], *reinterpret_cast<const bf16x8*>(xr + 32 * ks), a, 0, 0, 0);
          typedef __bf16 bf16x4_t __attribute__((ext_vector_type(4)));
          const bf16x4_t o4 = {(bf16_t)a[0], (bf16_t)a[1], (bf16_t)a[2], (bf16_t)a[3]};
          *reinterpret_cast<bf16x4_t*>(sY + (mt * 16 + pl) * RSY + (wv + 4 * zt) * 16 + 4 * q) = o4;
        }
      }
      __syncthreads();
    }
    // ---- dz1 = k1 g + k2 z + k3 -> sY (rows beyond M: zeros, they then add nothing to either contraction)
#pragma unroll
    for (int j = 0; j < NPY; ++j) {
      const int r = ry + (256 / CPY) * j;
      const bool rok = m0 + r < p.M;
      const bf16x8 d8 = *reinterpret_cast<const bf16x8*>(&vy[j]);
      bf16x8 z8;
      if constexpr (RC) z8 = *reinterpret_cast<const bf16x8*>(sY + r * RSY + cy);      // (this thread overwrites the same chunk)
      else z8 = *reinterpret_cast<const bf16x8*>(&vz[j]);
      bf16x8 o;
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const f32x4 k1 = *reinterpret_cast<const f32x4*>(kt_ + cy + 4 * h), k2 = *reinterpret_cast<const f32x4*>(kt_ + CO + cy + 4 * h);
        const f32x4 k3 = *reinterpret_cast<const f32x4*>(kt_ + 2 * CO + cy + 4 * h);
        const f32x4 ma = *reinterpret_cast<const f32x4*>(kt_ + 3 * CO + cy + 4 * h), mb = *reinterpret_cast<const f32x4*>(kt_ + 4 * CO + cy + 4 * h);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float zv = (float)z8[4 * h + e];
          const float g = fmaf(zv, ma[e], mb[e]) > 0.f ? (float)d8[4 * h + e] : 0.f;
          o[4 * h + e] = rok ? (bf16_t)fmaf(k1[e], g, fmaf(k2[e], zv, k3[e])) : (bf16_t)0.f;
        }
      }
      *reinterpret_cast<bf16x8*>(sY + r * RSY + cy) = o;
    }
    if constexpr (!RC) store_x();
    __syncthreads();
    if (t + gridDim.x < ntiles) issue(t + gridDim.x);             // the next tile travels under this tile's contractions

    // ---- input gradient: dx[m][ci] = sum_co dz1[m][co] * Wt[ci][co]; A = Wt fragments, B = dz1 rows.  The two tiles of a
    // pair take the rows of Wt permuted (A row i of tile h <-> channel 8 (i / 4) + 4 h + i % 4 of the pair) so that
    // accumulator h of lane (pl, q) holds channels 8 q + 4 h + (0..3): 8 consecutive channels per lane
#pragma unroll(PRE ? MPW : 1)
    for (int j = 0; j < MPW; ++j) {
      const int rl = (mt0 + WPP * j) * 16 + pl;
      const long m = m0 + rl;
      const bool mok = m < p.M;
      const long mc = mok ? m : p.M - 1;
      u32x4 r4 = {0u, 0u, 0u, 0u}, z4 = {0u, 0u, 0u, 0u}, zd4 = {0u, 0u, 0u, 0u};
      bool has_r;
      if constexpr (PRE) {
        r4 = er[j]; z4 = ez[j]; zd4 = ezd[j];
        has_r = (rmask >> j) & 1u;
      } else {
        const long rm = res_row(mc, has_r);
        if (p.R) r4 = *reinterpret_cast<const u32x4*>(p.R + rm * p.ldr + c0);
        if (p.bpart) {
          z4 = *reinterpret_cast<const u32x4*>(p.bz + mc * CI + c0);
          if (p.bzd) zd4 = *reinterpret_cast<const u32x4*>(p.bzd + mc * CI + c0);
        }
      }
      f32x4 a0 = {0.f, 0.f, 0.f, 0.f}, a1 = {0.f, 0.f, 0.f, 0.f};
      const bf16_t* br = sY + rl * RSY + 8 * q;
      const bf16_t* wr = sW + (pr * 32 + 8 * (pl >> 2) + (pl & 3)) * (CO + 8) + 8 * q;
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        const bf16x8 b = *reinterpret_cast<const bf16x8*>(br + 32 * ks);
        a0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(*reinterpret_cast<const bf16x8*>(wr + 32 * ks), b, a0, 0, 0, 0);
        a1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(*reinterpret_cast<const bf16x8*>(wr + 4 * (CO + 8) + 32 * ks), b, a1, 0, 0, 0);
      }
      const bf16x8 x8 = *reinterpret_cast<const bf16x8*>(sX + rl * RSX + c0);
      const bf16x8 r8 = *reinterpret_cast<const bf16x8*>(&r4);
      bf16x8 o;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        float v = (e < 4 ? a0[e & 3] : a1[e & 3]) + (has_r ? (float)r8[e] : 0.f);
        if (p.use_mask && !((float)x8[e] > 0.f)) v = 0.f;
        o[e] = (bf16_t)v;
      }
      if (mok) {
        *reinterpret_cast<bf16x8*>(p.dX + m * CI + c0) = o;
        if (p.bpart) {
          const bf16x8 z8 = *reinterpret_cast<const bf16x8*>(&z4), zd8 = *reinterpret_cast<const bf16x8*>(&zd4);
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            const float vr = (float)o[e];
            ss1[e] += vr;
            ss2[e] = fmaf(vr, (float)z8[e] - bt_[c0 + e], ss2[e]);
            if (p.bzd) ss3[e] = fmaf(vr, (float)zd8[e] - bt_[CI + c0 + e], ss3[e]);
          }
        }
      }
    }
    if (t + gridDim.x < ntiles) issue_epi(t + gridDim.x);         // (this tile's epilogue values are consumed)
    // ---- weight gradient: dW[co][ci] += sum_m dz1[m][co] * x[m][ci]: both operands through transposing reads (k = row)
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const int row = ks * 32 + g4 * 8 + q4;
      bf16x8 af[NTW], bfr[KTW];
#pragma unroll
      for (int nt = 0; nt < NTW; ++nt) {
        const bf16_t* a = sY + row * RSY + wn * (NTW * 16) + nt * 16 + p4 * 4;
        af[nt] = td_tr_read8(a, a + 4 * RSY);
      }
#pragma unroll
      for (int kt = 0; kt < KTW; ++kt) {
        const bf16_t* b = sX + row * RSX + wk * (KTW * 16) + kt * 16 + p4 * 4;
        bfr[kt] = td_tr_read8(b, b + 4 * RSX);
      }
#pragma unroll
      for (int nt = 0; nt < NTW; ++nt)
#pragma unroll
        for (int kt = 0; kt < KTW; ++kt) acc[nt][kt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[nt], bfr[kt], acc[nt][kt], 0, 0, 0);
    }
  }
  // ---- partials of this workgroup
  float* wp = p.wpart + (long)blockIdx.x * CO * CI;
#pragma unroll
  for (int nt = 0; nt < NTW; ++nt)
#pragma unroll
    for (int kt = 0; kt < KTW; ++kt) {
      const int k = wk * (KTW * 16) + kt * 16 + pl;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int n = wn * (NTW * 16) + nt * 16 + 4 * g4 + e;
        wp[(long)n * CI + k] = acc[nt][kt][e];
      }
    }
  if (p.bpart) {
    // lanes sharing q (the 16 rows of a tile), then the WPP waves sharing the pair, in a fixed order
    float* bp = p.bpart + (long)blockIdx.x * 3 * CI;
    float* red = reinterpret_cast<float*>(sY);                    // [4 waves][3][32]
    __syncthreads();                                              // (the last tile's LDS reads are over)
#pragma unroll
    for (int e = 0; e < 8; ++e) {
#pragma unroll
      for (int o = 1; o < 16; o <<= 1) {
        ss1[e] += __shfl_xor(ss1[e], o, 64);
        ss2[e] += __shfl_xor(ss2[e], o, 64);
        ss3[e] += __shfl_xor(ss3[e], o, 64);
      }
    }
    if (pl == 0) {
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        red[(wv * 3 + 0) * 32 + 8 * q + e] = ss1[e];
        red[(wv * 3 + 1) * 32 + 8 * q + e] = ss2[e];
        red[(wv * 3 + 2) * 32 + 8 * q + e] = ss3[e];
      }
    }
    __syncthreads();
    for (int i = tid; i < 3 * CI; i += 256) {
      const int which = i / CI, c = i - which * CI;
      const int pp = c >> 5, cl = c & 31;
      float a = 0.f;
#pragma unroll
      for (int k = 0; k < WPP; ++k) a += red[((pp + PAIRS * k) * 3 + which) * 32 + cl];
      bp[i] = a;
    }
  }
}

extern "C" int tdeed_narrow_conv1_bwd_fits(int Co, int Ci) {
  return ((Co == 64 && (Ci == 32 || Ci == 64)) || (Co == 128 && Ci == 64) || (Co == 128 && Ci == 128)) ? 1 : 0;
}
// persistent workgroups; also the row count of wpart / bpart
extern "C" int tdeed_narrow_conv1_bwd_grid(long M, int Co, int Ci) {
  const long tiles = (M + 63) / 64;
  const long cap = (Co == 64) ? 768 : 512;                       // resident workgroups: three per CU at 64 <- 32, two otherwise
  return (int)(tiles < cap ? tiles : cap);
}

// dY, Z [M][Co] bf16 (Z NULL: recomputed as X @ W^T in the launch -- Z must be exactly that product, not at 128 <- 128); fa / fb / mean / rstd / w: conv1's BatchNorm (forward affine, batch statistics, weight); sums fp32 [2][Co]
// = (sum g, sum g * xhat) of its backward (tdeed_bn_bwd_masked_from_parts leaves them); X [M][Ci]; Wt [Ci][Co] (the weight
// transposed); R: shortcut gradient ([M][Ci], or with r_hi > 0 the rows of the even pixels of an r_hi x r_wi frame) or NULL;
// dX [M][Ci]; use_mask: dX *= [X > 0]; bz / bmean (/ bzd / bmean_d) + bpart fp32 [grid][3][Ci]: the gradient sink's statistics
// (NULL: none); wpart fp32 [grid][Co][Ci]: partial weight gradients, grid = tdeed_narrow_conv1_bwd_grid(M, Co, Ci).
extern "C" int tdeed_narrow_conv1_bwd(const void* dY, const void* Z, long M, int Co, int Ci, const float* fa, const float* fb,
                                      const float* mean, const float* rstd, const float* w, const float* sums, const void* X,
                                      const void* Wt, const void* R, long ldr, int r_hi, int r_wi, void* dX, int use_mask,
                                      const void* bz, const float* bmean, const void* bzd, const float* bmean_d, float* bpart,
                                      float* wpart, void* stream) {
  TD_CHECK(dY && fa && fb && mean && rstd && w && sums && X && Wt && dX && wpart, "narrow_conv1_bwd: null pointer");
  TD_CHECK(Z || !(Co == 128 && Ci == 128), "narrow_conv1_bwd: the 128 <- 128 form reads Z (no recompute)");
  TD_CHECK(M > 0 && tdeed_narrow_conv1_bwd_fits(Co, Ci), "narrow_conv1_bwd: %d <- %d channels not served", Co, Ci);
  TD_CHECK(!R || ldr % 4 == 0, "narrow_conv1_bwd: bad residual stride");
  TD_CHECK(r_hi == 0 || (R && r_hi > 0 && r_wi > 0 && M % ((long)r_hi * r_wi) == 0), "narrow_conv1_bwd: bad stride-2 residual geometry");
  TD_CHECK(!bpart || (bz && bmean && (!bzd || bmean_d)), "narrow_conv1_bwd: statistics operands missing");
  NbwP p{};
  p.dY = (const bf16_t*)dY; p.Z = (const bf16_t*)Z; p.fa = fa; p.fb = fb; p.mean = mean; p.rstd = rstd; p.w = w; p.sums = sums;
  p.X = (const bf16_t*)X; p.Wt = (const bf16_t*)Wt; p.R = (const bf16_t*)R; p.ldr = ldr; p.r_hi = r_hi; p.r_wi = r_wi;
  p.dX = (bf16_t*)dX; p.use_mask = use_mask;
  p.bz = (const bf16_t*)bz; p.bmean = bmean; p.bzd = bpart ? (const bf16_t*)bzd : nullptr; p.bmean_d = bmean_d; p.bpart = bpart;
  p.wpart = wpart; p.M = M; p.inv_M = 1.0f / (float)M;
  const int grid = tdeed_narrow_conv1_bwd_grid(M, Co, Ci);
  hipStream_t st = (hipStream_t)stream;
  if (!Z) {
    if (Co == 64 && Ci == 32) hipLaunchKernelGGL((narrow_conv1_bwd_kernel<64, 32, true>), dim3(grid), dim3(256), 0, st, p);
    else if (Co == 64) hipLaunchKernelGGL((narrow_conv1_bwd_kernel<64, 64, true>), dim3(grid), dim3(256), 0, st, p);
    else hipLaunchKernelGGL((narrow_conv1_bwd_kernel<128, 64, true>), dim3(grid), dim3(256), 0, st, p);
  } else if (Co == 64 && Ci == 32) hipLaunchKernelGGL((narrow_conv1_bwd_kernel<64, 32>), dim3(grid), dim3(256), 0, st, p);
  else if (Co == 64) hipLaunchKernelGGL((narrow_conv1_bwd_kernel<64, 64>), dim3(grid), dim3(256), 0, st, p);
  else if (Ci == 64) hipLaunchKernelGGL((narrow_conv1_bwd_kernel<128, 64>), dim3(grid), dim3(256), 0, st, p);
  else hipLaunchKernelGGL((narrow_conv1_bwd_kernel<128, 128>), dim3(grid), dim3(256), 0, st, p);
  TD_LAUNCH_CHECK("narrow_conv1_bwd");
  return TDEED_OK;
}
