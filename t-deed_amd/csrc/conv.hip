// Non-GEMM pieces of the RegNetY trunk: fused pre-proc + stem, grouped 3x3 conv (+BN+ReLU+SE
// squeeze), SE excitation, global average pool + positional encoding.
// All are HBM / VALU bound (group width 8/16 and K=27 are no MFMA shapes); weights are
// block-uniform so they travel through the scalar cache (s_load) and feed v_fma as SGPR operands.
#include "common.h"

// =========================================================================== stem
// 16x16 output pixels per block; the 33x33x3 input patch is normalised once into LDS.
// IN = uint8_t (the normal path: 1 byte/px in HBM) or float (mixup batches: l*frame + (1-l)*frame2 is not integral)
template <typename T, typename IN>
__global__ __launch_bounds__(256) void stem_kernel(const IN* __restrict__ frames, int H, int W,
                                                   int top, int left, int ch, int cw, int flip_all,
                                                   const unsigned char* __restrict__ flip_mask,
                                                   const float* __restrict__ w,
                                                   const float* __restrict__ scale,
                                                   const float* __restrict__ shift, T* __restrict__ out,
                                                   int Ho, int Wo, int relu) {
  __shared__ float tile[3][33][34];
  const int n = blockIdx.z;
  const int oy0 = blockIdx.y * 16, ox0 = blockIdx.x * 16;
  const int iy0 = oy0 * 2 - 1, ix0 = ox0 * 2 - 1;
  const float mean[3] = {0.485f, 0.456f, 0.406f};
  const float stdv[3] = {0.229f, 0.224f, 0.225f};
  const IN* src = frames + (long)n * 3 * H * W;
  const int flip = flip_mask ? (int)flip_mask[n] : flip_all;      // per-frame h-flip (train-time augment) or one flag
  for (int i = threadIdx.x; i < 3 * 33 * 33; i += 256) {
    int c = i / (33 * 33);
    int r = i - c * 33 * 33;
    int y = r / 33, x = r - y * 33;
    int iy = iy0 + y, ix = ix0 + x;
    float v = 0.f;
    if (iy >= 0 && iy < ch && ix >= 0 && ix < cw) {
      int sx = flip ? (cw - 1 - ix) : ix;
      float u = (float)src[((long)c * H + (top + iy)) * W + (left + sx)];
      v = (u / 255.0f - mean[c]) / stdv[c];
    }
    tile[c][y][x] = v;
  }
  __syncthreads();
  const int ty = threadIdx.x >> 4, tx = threadIdx.x & 15;
  const int oy = oy0 + ty, ox = ox0 + tx;
  float in[27];
#pragma unroll
  for (int c = 0; c < 3; ++c)
#pragma unroll
    for (int ky = 0; ky < 3; ++ky)
#pragma unroll
      for (int kx = 0; kx < 3; ++kx) in[c * 9 + ky * 3 + kx] = tile[c][2 * ty + ky][2 * tx + kx];
  if (oy >= Ho || ox >= Wo) return;
  T* dst = out + (((long)n * Ho + oy) * Wo + ox) * 32;
  constexpr int EPC = Chunk<T>::N;
#pragma unroll
  for (int o0 = 0; o0 < 32; o0 += EPC) {
    float v[EPC];
#pragma unroll
    for (int e = 0; e < EPC; ++e) {
      const float* wo = w + (o0 + e) * 27;
      float a = 0.f;
#pragma unroll
      for (int i = 0; i < 27; ++i) a = fmaf(in[i], wo[i], a);
      v[e] = relu ? fmaxf(a * scale[o0 + e] + shift[o0 + e], 0.f) : a * scale[o0 + e] + shift[o0 + e];
    }
    Chunk<T>::store(dst + o0, v);
  }
}

extern "C" int tdeed_stem_fwd(const void* frames, int frames_f32, int N, int H, int W, int crop_top, int crop_left,
                              int crop_h, int crop_w, int flip, const unsigned char* flip_mask, const float* w,
                              const float* scale, const float* shift, void* out, int relu, int dtype, void* stream) {
  TD_CHECK(frames && w && scale && shift && out, "stem: null pointer");
  TD_CHECK(N > 0 && crop_h > 0 && crop_w > 0 && crop_top >= 0 && crop_left >= 0 &&
               crop_top + crop_h <= H && crop_left + crop_w <= W,
           "stem: bad geometry N=%d H=%d W=%d crop=(%d,%d,%d,%d)", N, H, W, crop_top, crop_left, crop_h, crop_w);
  TD_CHECK(N <= 65535, "stem: at most 65535 frames per launch (got %d)", N);
  const int Ho = (crop_h + 1) / 2, Wo = (crop_w + 1) / 2;
  dim3 grid(cdiv(Wo, 16), cdiv(Ho, 16), N);
  hipStream_t st = (hipStream_t)stream;
#define TD_STEM(TT, IN)                                                                                                \
  hipLaunchKernelGGL((stem_kernel<TT, IN>), grid, dim3(256), 0, st, (const IN*)frames, H, W, crop_top, crop_left, crop_h, \
                     crop_w, flip, flip_mask, w, scale, shift, (TT*)out, Ho, Wo, relu)
  if (dtype == TDEED_F32) { if (frames_f32) TD_STEM(float, float); else TD_STEM(float, uint8_t); }
  else if (dtype == TDEED_BF16) { if (frames_f32) TD_STEM(bf16_t, float); else TD_STEM(bf16_t, uint8_t); }
#undef TD_STEM
  else { tdeed_set_error("stem: bad dtype %d", dtype); return TDEED_ERR_ARG; }
  TD_LAUNCH_CHECK("stem");
  return TDEED_OK;
}

// =========================================================================== grouped 3x3
// One block = one (frame, group); lanes walk the output pixels, the group's gw*gw*9 weights are
// block-uniform (packed [G][9][gw_in][gw_out] so the inner o-loop is one s_load_dwordx8/x16).
// The SE squeeze (mean over Ho*Wo) is completed inside the block: no atomics, deterministic.
template <typename T, int GW, int NTHR>
__global__ __launch_bounds__(NTHR) void gconv3x3_kernel(const T* __restrict__ x, int Hi, int Wi, int C,
                                                        int stride, const float* __restrict__ wp,
                                                        const float* __restrict__ scale,
                                                        const float* __restrict__ shift, T* __restrict__ y,
                                                        float* __restrict__ pooled, int Ho, int Wo, int relu) {
  constexpr int EPC = Chunk<T>::N;
  constexpr int NCH = GW / EPC;   // 16-B chunks per pixel per group
  constexpr int NW = NTHR / 64;
  __shared__ float red[NW][GW];
  const int g = blockIdx.x, n = blockIdx.y;
  const float* wg = wp + (long)g * 9 * GW * GW;
  const T* xin = x + (long)n * Hi * Wi * C + g * GW;
  T* yout = y + (long)n * Ho * Wo * C + g * GW;
  float sc[GW], sh[GW], psum[GW];
#pragma unroll
  for (int o = 0; o < GW; ++o) {
    sc[o] = scale[g * GW + o];
    sh[o] = shift[g * GW + o];
    psum[o] = 0.f;
  }
  const int npix = Ho * Wo;
  for (int p = threadIdx.x; p < npix; p += NTHR) {
    const int oy = p / Wo, ox = p - oy * Wo;
    float acc[GW];
#pragma unroll
    for (int o = 0; o < GW; ++o) acc[o] = 0.f;
#pragma unroll
    for (int ky = 0; ky < 3; ++ky) {
      const int iy = oy * stride - 1 + ky;
#pragma unroll
      for (int kx = 0; kx < 3; ++kx) {
        const int ix = ox * stride - 1 + kx;
        if (iy >= 0 && iy < Hi && ix >= 0 && ix < Wi) {
          const T* src = xin + ((long)iy * Wi + ix) * C;
          float in[GW];
#pragma unroll
          for (int c = 0; c < NCH; ++c) {
            float v[EPC];
            Chunk<T>::load(src + c * EPC, v);
#pragma unroll
            for (int e = 0; e < EPC; ++e) in[c * EPC + e] = v[e];
          }
          const float* wt = wg + (ky * 3 + kx) * GW * GW;
#pragma unroll
          for (int i = 0; i < GW; ++i)
#pragma unroll
            for (int o = 0; o < GW; ++o) acc[o] = fmaf(in[i], wt[i * GW + o], acc[o]);
        }
      }
    }
    float outv[GW];
#pragma unroll
    for (int o = 0; o < GW; ++o) {
      outv[o] = relu ? fmaxf(acc[o] * sc[o] + sh[o], 0.f) : acc[o] * sc[o] + sh[o];
    }
    T* dst = yout + (long)p * C;
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      float v[EPC];
#pragma unroll
      for (int e = 0; e < EPC; ++e) v[e] = outv[c * EPC + e];
      Chunk<T>::store(dst + c * EPC, v);
      // the squeeze sees what the next layer sees: the stored (rounded) activation
#pragma unroll
      for (int e = 0; e < EPC; ++e) psum[c * EPC + e] += round_to<T>(v[e]);
    }
  }
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
#pragma unroll
  for (int o = 0; o < GW; ++o) {
    float s = wave_sum(psum[o]);
    if (lane == 0) red[wv][o] = s;
  }
  __syncthreads();
  if (threadIdx.x < GW) {
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NW; ++i) s += red[i][threadIdx.x];
    pooled[(long)n * C + g * GW + threadIdx.x] = s;   // sum; the SE kernel applies 1/(Ho*Wo)
  }
}

template <typename T, int GW>
static int launch_gconv(const void* x, int N, int Hi, int Wi, int C, int stride, const float* w,
                        const float* scale, const float* shift, void* y, float* pooled, int relu, hipStream_t st) {
  const int Ho = (Hi - 1) / stride + 1, Wo = (Wi - 1) / stride + 1;
  dim3 grid(C / GW, N);
  if (Ho * Wo <= 64)
    hipLaunchKernelGGL((gconv3x3_kernel<T, GW, 64>), grid, dim3(64), 0, st, (const T*)x, Hi, Wi, C, stride, w,
                       scale, shift, (T*)y, pooled, Ho, Wo, relu);
  else
    hipLaunchKernelGGL((gconv3x3_kernel<T, GW, 256>), grid, dim3(256), 0, st, (const T*)x, Hi, Wi, C, stride, w,
                       scale, shift, (T*)y, pooled, Ho, Wo, relu);
  TD_LAUNCH_CHECK("gconv3x3");
  return TDEED_OK;
}

// --------------------------------------------------------------------------- bf16 MFMA variant
// Implicit GEMM per "unit" of 16 output channels (two gw=8 groups block-diagonal, or one gw=16
// group): D[n][pixel] = sum_k Wt[n][k] X[k][pixel], k = (half, tap, in-ch 8) -> 18 slots of 8 = 5
// k-steps of v_mfma_f32_16x16x32_bf16.  The X fragment of a lane (pixel l&15, slot 4ks+(l>>4)) is
// exactly one 16-byte chunk of the channels-last input at pixel+tap, so a zero-padded halo band of
// the input (rows x (Wi+2) x slab of <=64 channels, pixel stride slab*2+16 B => conflict-free
// ds_read_b128) is the only staging needed.  Weights come pre-packed per (unit, k-step, lane)
// and stay in registers.  The weights are the MFMA A operand so that each lane ends up with 4
// consecutive channels of one pixel (8-byte stores) rather than 4 pixels of one channel.
#include <stdlib.h>
// 48 KB: three workgroups per CU.  Measured best since three batches share the chip (same-box A/B, cfg2: 4110 -> 4196
// clips/s together with the fused front's budget; 64 KB = two per CU was best with two sub-batches per batch)
static long gc_cap() { return 48 * 1024; }
struct GcGeom { int band, nbands, CSP, nslabs, PS, rows_in; };
static GcGeom gc_geom(int Hi, int Wi, int C, int stride) {
  GcGeom g;
  const int Ho = (Hi - 1) / stride + 1;
  g.CSP = C >= 64 ? 64 : (C > 16 ? 32 : 16);
  if (C > 32 && C < 64) g.CSP = 64;
  g.nslabs = (C + g.CSP - 1) / g.CSP;
  g.PS = g.CSP * 2 + 16;
  const long rowb = (long)(Wi + 2) * g.PS;
  int rows = (int)(gc_cap() / rowb);
  if (rows < 3 && 3 * rowb <= 64 * 1024) rows = 3;
  int band = rows >= 3 ? (rows - 3) / stride + 1 : 0;
  if (band > Ho) band = Ho;
  g.band = band;
  g.nbands = band > 0 ? (Ho + band - 1) / band : 0;
  g.rows_in = band > 0 ? (band - 1) * stride + 3 : 0;
  return g;
}

// AFF (training): the input is a RAW conv output z and the BatchNorm affine + ReLU of the layer in front,
// relu(in_a[c] * z + in_b[c]), is applied while the band is staged (the halo stays zero): the post-BN map is never
// materialised.  Rounded to bf16 like the materialised map would be, so the result is bit-identical.
// The grouped 3x3 itself, from a zero-haloed band of the (post-BN-ReLU) input in LDS: implicit GEMM per 16-channel unit,
// BatchNorm (+ ReLU), output rows, SE squeeze partial sums (and sums of squares in training).  Shared by the kernel that
// stages the band from a stored map and by the one that computes it (conv1 in front, c1_gconv_mfma_kernel).
// Training backward (the stride-1 input gradient of conv2 is this kernel with flipped / transposed weights): when the output
// d_y1 arrives at conv1's BatchNorm + ReLU, that BatchNorm's backward needs sum g and sum g * (z1 - mean) with g = d_y1 masked by
// [fa z1 + fb > 0].  With `bz` given the per-(frame, band) partial rows `pooled` / `pooled_sq` hold exactly those sums (the
// stored output stays unmasked): the statistics pass over (d_y1, z1) disappears, z1 is read here once, 8 bytes per lane and tile.
struct GcStat {
  const bf16_t* z; const float* fa; const float* fb; const float* mean;
};
// the weight fragments of a wave (pair form: units 2 pr, 2 pr + 1 of the slab; single-unit form: a only).  Requested by the
// kernels BEFORE they stage the band, so that the L2 round trip of the weights runs under the band's loads instead of behind
// the barrier that ends the staging (a workgroup lives ~5 us; this was ~1 us of it).
struct GcW { bf16x8 a[5], b[5]; };
__device__ __forceinline__ GcW gconv_load_w(const bf16x8* __restrict__ wfrag, int slab, int CSP) {
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int units = CSP >> 4;
  GcW w;
  if (units >= 2) {
    const int pairs = units >> 1, pr = wv % pairs;
    const long ubase = ((long)(slab * 4 + 2 * pr) * 5) * 64 + lane;
#pragma unroll
    for (int ks = 0; ks < 5; ++ks) {
      w.a[ks] = wfrag[ubase + ks * 64];
      w.b[ks] = wfrag[ubase + (5 + ks) * 64];
    }
  } else {
    const long ubase = ((long)(slab * 4) * 5) * 64 + lane;
#pragma unroll
    for (int ks = 0; ks < 5; ++ks) {
      w.a[ks] = wfrag[ubase + ks * 64];
      w.b[ks] = w.a[ks];
    }
  }
  return w;
}
template <int STRIDE>
__device__ __forceinline__ void gconv_band_mma(const GcW& gw_, const unsigned char* tile, float (*red)[32], float (*redq)[32], int Wi, int C,
                                               const bf16x8* __restrict__ wfrag, const float* __restrict__ scale,
                                               const float* __restrict__ shift, bf16_t* __restrict__ y,
                                               float* __restrict__ pooled, float* __restrict__ pooled_sq, int Ho, int Wo,
                                               int nbands, int CSP, int PS, int relu, int n, int bnd, int slab, int oy0,
                                               int nrows_out, const GcStat bst = GcStat{nullptr, nullptr, nullptr, nullptr},
                                               long long* dbg = nullptr) {
#define GC_STAMP(i) do { if (dbg && threadIdx.x == 0) dbg[(long)blockIdx.x * 8 + (i)] = wall_clock64(); } while (0)
  const int WP = Wi + 2;
  const int cs0 = slab * CSP;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int q = lane >> 4, pl = lane & 15;
  const int units = CSP >> 4;
  if (units >= 2) {
    // ---- pair form: a wave serves TWO adjacent 16-channel units on the same 16 pixels, then swaps accumulator rows between
    // the two tiles (v_permlane16_swap) so that a lane holds 8 consecutive channels: one 16-byte store per lane, 64 contiguous
    // bytes per pixel and wave instruction.  With one unit per wave the stores (and the statistics loads) are 16 pixels x 32
    // bytes, a shape that moves the same bytes ~1.5x slower (tools/ubench/access_shape.hip: 4.3 vs 6.2 TB/s).  Two independent
    // accumulator chains per tile also keep the MFMA pipe busier than five dependent MFMAs.
    const int pairs = units >> 1, pr = wv % pairs, mstep = 4 / pairs;
    const bf16x8 (&wfA)[5] = gw_.a;
    const bf16x8 (&wfB)[5] = gw_.b;
    int off[5];
#pragma unroll
    for (int ks = 0; ks < 5; ++ks) {
      const int sidx = 4 * ks + q;
      const int half = sidx / 9, tap = sidx - half * 9;
      const int dy = tap / 3, dx = tap - dy * 3;
      off[ks] = sidx < 18 ? (dy * WP + dx) * PS + half * 16 + pr * 64 : pr * 64;
    }
    const int chA = cs0 + pr * 32 + q * 4, chB = chA + 16;        // accumulator rows: 4 channels of unit A, 4 of unit B
    const int chS = cs0 + pr * 32 + (q & 1) * 16 + (q >> 1) * 8;  // after the swap: this lane's 8 consecutive channels
    float scA[4], shA[4], scB[4], shB[4], psA[4], psB[4], pqA[4], pqB[4];
    {
      // 16-byte loads (chA, chB are multiples of 4 and C of 8: a group of 4 channels is inside or outside)
      const bool oka = chA < C, okb = chB < C;
      const int ca = min(chA, C - 4), cb = min(chB, C - 4);
      const f32x4 vsa = *reinterpret_cast<const f32x4*>(scale + ca), vha = *reinterpret_cast<const f32x4*>(shift + ca);
      const f32x4 vsb = *reinterpret_cast<const f32x4*>(scale + cb), vhb = *reinterpret_cast<const f32x4*>(shift + cb);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        scA[r] = oka ? vsa[r] : 0.f;
        shA[r] = oka ? vha[r] : 0.f;
        scB[r] = okb ? vsb[r] : 0.f;
        shB[r] = okb ? vhb[r] : 0.f;
        psA[r] = psB[r] = pqA[r] = pqB[r] = 0.f;
      }
    }
    float faA[4], fbA[4], muA[4], faB[4], fbB[4], muB[4];
    if (bst.z) {
      const int ca = min(chA, C - 4), cb = min(chB, C - 4);       // (groups beyond C are never used: their sums are gated)
      const f32x4 a0 = *reinterpret_cast<const f32x4*>(bst.fa + ca), a1_ = *reinterpret_cast<const f32x4*>(bst.fb + ca),
                  a2 = *reinterpret_cast<const f32x4*>(bst.mean + ca);
      const f32x4 b0 = *reinterpret_cast<const f32x4*>(bst.fa + cb), b1_ = *reinterpret_cast<const f32x4*>(bst.fb + cb),
                  b2_ = *reinterpret_cast<const f32x4*>(bst.mean + cb);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        faA[r] = a0[r]; fbA[r] = a1_[r]; muA[r] = a2[r];
        faB[r] = b0[r]; fbB[r] = b1_[r]; muB[r] = b2_[r];
      }
    }
    const int npix = nrows_out * Wo;
    const int ntiles = (npix + 15) >> 4;
    bf16_t* yout = y + ((long)n * Ho + oy0) * Wo * C;
    const bf16_t* zin = bst.z ? bst.z + ((long)n * Ho + oy0) * Wo * C : nullptr;
    const IDiv dwo(Wo);
    const bool sok = chS < C;                                     // (C is a multiple of 8: a chunk is inside or outside)
    GC_STAMP(4);
    for (int mt = wv / pairs; mt < ntiles; mt += mstep) {
      const int p = mt * 16 + pl;
      const bool pok = p < npix;
      const int pc = pok ? p : 0;
      int oyl, ox;
      dwo.divmod(pc, oyl, ox);
      const unsigned char* base = tile + ((long)(oyl * STRIDE) * WP + ox * STRIDE) * PS;
      u32x4 zs = {0u, 0u, 0u, 0u};
      if (zin) zs = *reinterpret_cast<const u32x4*>(zin + (long)pc * C + (sok ? chS : 0));   // travels under the MFMA chains
      f32x4 accA = {0.f, 0.f, 0.f, 0.f}, accB = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < 5; ++ks) {
        const bf16x8 xa = *reinterpret_cast<const bf16x8*>(base + off[ks]);
        const bf16x8 xb = *reinterpret_cast<const bf16x8*>(base + off[ks] + 32);
        accA = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wfA[ks], xa, accA, 0, 0, 0);
        accB = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wfB[ks], xb, accB, 0, 0, 0);
      }
      bf16x4 oA, oB;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float va = accA[r] * scA[r] + shA[r], vb = accB[r] * scB[r] + shB[r];
        oA[r] = (bf16_t)(relu ? fmaxf(va, 0.f) : va);
        oB[r] = (bf16_t)(relu ? fmaxf(vb, 0.f) : vb);
      }
      if (zin) {
        // the statistics map arrives in the stored (swapped) layout: the same swap takes it back to accumulator rows
        const auto z0 = __builtin_amdgcn_permlane16_swap(zs[0], zs[2], false, false);
        const auto z1 = __builtin_amdgcn_permlane16_swap(zs[1], zs[3], false, false);
        const u32x2 za2 = {z0[0], z1[0]}, zb2 = {z0[1], z1[1]};
        const bf16x4 zA = *reinterpret_cast<const bf16x4*>(&za2), zB = *reinterpret_cast<const bf16x4*>(&zb2);
        if (pok) {
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float za = (float)zA[r], zb = (float)zB[r];
            const float ga = (chA + r < C && fmaf(za, faA[r], fbA[r]) > 0.f) ? (float)oA[r] : 0.f;
            const float gb = (chB + r < C && fmaf(zb, faB[r], fbB[r]) > 0.f) ? (float)oB[r] : 0.f;
            psA[r] += ga;
            pqA[r] = fmaf(ga, za - muA[r], pqA[r]);
            psB[r] += gb;
            pqB[r] = fmaf(gb, zb - muB[r], pqB[r]);
          }
        }
      } else if (pok) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          psA[r] += (float)oA[r];
          pqA[r] = fmaf((float)oA[r], (float)oA[r], pqA[r]);
          psB[r] += (float)oB[r];
          pqB[r] = fmaf((float)oB[r], (float)oB[r], pqB[r]);
        }
      }
      const u32x2 a2 = *reinterpret_cast<const u32x2*>(&oA), b2 = *reinterpret_cast<const u32x2*>(&oB);
      const auto s0 = __builtin_amdgcn_permlane16_swap(a2[0], b2[0], false, false);
      const auto s1 = __builtin_amdgcn_permlane16_swap(a2[1], b2[1], false, false);
      if (pok && sok) *reinterpret_cast<u32x4*>(yout + (long)pc * C + chS) = (u32x4){s0[0], s1[0], s0[1], s1[1]};
    }
    GC_STAMP(5);
    // ---- squeeze partial sums: lanes sharing q, then the waves sharing the pair (fixed order)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float va = td_row16_sum(psA[r]), vb = td_row16_sum(psB[r]);
      if (pl == 0) {
        red[wv][q * 4 + r] = va;
        red[wv][16 + q * 4 + r] = vb;
      }
      if (pooled_sq) {
        const float wa = td_row16_sum(pqA[r]), wb = td_row16_sum(pqB[r]);
        if (pl == 0) {
          redq[wv][q * 4 + r] = wa;
          redq[wv][16 + q * 4 + r] = wb;
        }
      }
    }
    __syncthreads();
    if (threadIdx.x < units * 16) {
      const int u = threadIdx.x >> 4, cc = threadIdx.x & 15;
      float sres = 0.f, qres = 0.f;
      for (int w2 = u >> 1; w2 < 4; w2 += pairs) {
        sres += red[w2][(u & 1) * 16 + cc];
        if (pooled_sq) qres += redq[w2][(u & 1) * 16 + cc];
      }
      const int ch = cs0 + u * 16 + cc;
      if (ch < C) {
        pooled[((long)n * nbands + bnd) * C + ch] = sres;
        if (pooled_sq) pooled_sq[((long)n * nbands + bnd) * C + ch] = qres;
      }
    }
    GC_STAMP(6);
    return;
  }
  const int unit = wv % units;
  const int mstep = 4 / units;
  // weights of this unit (units == 1 here: unit 0 of the slab): 5 k-steps, wfrag laid out [slab*4+unit][ks][lane]
  const bf16x8 (&wf)[5] = gw_.a;
  int off[5];
#pragma unroll
  for (int ks = 0; ks < 5; ++ks) {
    const int sidx = 4 * ks + q;
    const int half = sidx / 9, tap = sidx - half * 9;
    const int dy = tap / 3, dx = tap - dy * 3;
    off[ks] = sidx < 18 ? (dy * WP + dx) * PS + half * 16 + unit * 32 : unit * 32;
  }
  const int ch0 = cs0 + unit * 16 + q * 4;      // this lane's 4 output channels
  float sc[4], sh[4], psum[4], psq[4];
  {
    const bool ok = ch0 < C;                      // (ch0 is a multiple of 4, C of 8)
    const int c_ = min(ch0, C - 4);
    const f32x4 vs = *reinterpret_cast<const f32x4*>(scale + c_), vh = *reinterpret_cast<const f32x4*>(shift + c_);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      sc[r] = ok ? vs[r] : 0.f;
      sh[r] = ok ? vh[r] : 0.f;
      psum[r] = 0.f;
      psq[r] = 0.f;
    }
  }
  float bfa[4], bfb[4], bmu[4];
  if (bst.z) {
    const int c_ = min(ch0, C - 4);
    const f32x4 v0 = *reinterpret_cast<const f32x4*>(bst.fa + c_), v1 = *reinterpret_cast<const f32x4*>(bst.fb + c_),
                v2 = *reinterpret_cast<const f32x4*>(bst.mean + c_);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      bfa[r] = v0[r];
      bfb[r] = v1[r];
      bmu[r] = v2[r];
    }
  }
  const int npix = nrows_out * Wo;
  const int ntiles = (npix + 15) >> 4;
  bf16_t* yout = y + ((long)n * Ho + oy0) * Wo * C;
  const bf16_t* zin = bst.z ? bst.z + ((long)n * Ho + oy0) * Wo * C : nullptr;
  const IDiv dwo(Wo);
  for (int mt = wv / units; mt < ntiles; mt += mstep) {
    const int p = mt * 16 + pl;
    const bool pok = p < npix;
    const int pc = pok ? p : 0;
    int oyl, ox;
    dwo.divmod(pc, oyl, ox);
    const unsigned char* base = tile + ((long)(oyl * STRIDE) * WP + ox * STRIDE) * PS;
    bf16x4 z4 = {};
    if (zin) z4 = *reinterpret_cast<const bf16x4*>(zin + (long)pc * C + min(ch0, C - 4));   // travels under the MFMA chain
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < 5; ++ks) {
      const bf16x8 xf = *reinterpret_cast<const bf16x8*>(base + off[ks]);
      acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[ks], xf, acc, 0, 0, 0);
    }
    if (pok && ch0 < C) {
      bf16x4 o;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float v = relu ? fmaxf(acc[r] * sc[r] + sh[r], 0.f) : acc[r] * sc[r] + sh[r];
        o[r] = (bf16_t)v;
        if (zin) {
          const float zv = (float)z4[r];
          const float g = fmaf(zv, bfa[r], bfb[r]) > 0.f ? (float)o[r] : 0.f;
          psum[r] += g;
          psq[r] = fmaf(g, zv - bmu[r], psq[r]);
        } else {
          psum[r] += (float)o[r];
          psq[r] = fmaf((float)o[r], (float)o[r], psq[r]);
        }
      }
      *reinterpret_cast<bf16x4*>(yout + (long)pc * C + ch0) = o;
    }
  }
  // ---- SE squeeze partial sums: lanes sharing q, then waves sharing the unit
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const float v = td_row16_sum(psum[r]);
    if (pl == 0) red[wv][q * 4 + r] = v;
    if (pooled_sq) {            // training: sums of squares too (BatchNorm statistics of the raw conv output)
      const float w = td_row16_sum(psq[r]);
      if (pl == 0) redq[wv][q * 4 + r] = w;
    }
  }
  __syncthreads();
  if (threadIdx.x < units * 16) {
    const int u = threadIdx.x >> 4, cc = threadIdx.x & 15;
    float sres = 0.f, qres = 0.f;
    for (int w2 = u; w2 < 4; w2 += units) {
      sres += red[w2][cc];
      if (pooled_sq) qres += redq[w2][cc];
    }
    const int ch = cs0 + u * 16 + cc;
    if (ch < C) {
      pooled[((long)n * nbands + bnd) * C + ch] = sres;
      if (pooled_sq) pooled_sq[((long)n * nbands + bnd) * C + ch] = qres;
    }
  }
}

template <int STRIDE, bool AFF>
__global__ __launch_bounds__(256) void gconv3x3_mfma_kernel(const bf16_t* __restrict__ x, int Hi, int Wi, int C,
                                                            const float* __restrict__ in_a, const float* __restrict__ in_b,
                                                            const bf16x8* __restrict__ wfrag,
                                                            const float* __restrict__ scale,
                                                            const float* __restrict__ shift,
                                                            bf16_t* __restrict__ y, float* __restrict__ pooled,
                                                            float* __restrict__ pooled_sq,
                                                            int Ho, int Wo, int band, int nbands, int CSP, int PS,
                                                            int rows_in, int relu, const GcStat bst) {
  extern __shared__ __attribute__((aligned(16))) unsigned char tile[];
  __shared__ float red[4][32];
  __shared__ float redq[4][32];
  // slabs and bands of one frame read the same pixel rows (different channel slices / halo rows): one XCD
  const long lid = xcd_logical_id(blockIdx.x, gridDim.x);
  const int nslabs_ = (C + CSP - 1) >> (CSP == 64 ? 6 : (CSP == 32 ? 5 : 4));        // CSP is 16, 32 or 64
  int slab, bnd, n, t_;
  td_split(lid, nslabs_, t_, slab);
  td_split(t_, nbands, n, bnd);
  const int cs0 = slab * CSP;
  const GcW gw_ = gconv_load_w(wfrag, slab, CSP);                // (travels under the staging below)
  const int oy0 = bnd * band;
  const int nrows_out = min(band, Ho - oy0);
  const int WP = Wi + 2;
  const int iy0 = oy0 * STRIDE - 1;
  // ---- stage the zero-padded input band: (row, col, chunk) with chunk fastest
  const int cpp = CSP >> 3;
  const int nrow_used = (nrows_out - 1) * STRIDE + 3;
  const bf16_t* xin = x + (long)n * Hi * Wi * C;
  // batches of 8 independent 16-B loads per thread before the LDS stores (memory-level parallelism)
  const int total = nrow_used * WP * cpp;
  const IDiv dcpp(cpp), dwp(WP);
  // cpp (4 or 8) divides 256: a thread stages the same 8-channel chunk in every iteration, its affine lives in registers
  float ia[8], ib[8];
  if constexpr (AFF) {
    const int jt = threadIdx.x % cpp;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int c = min(cs0 + jt * 8 + e, C - 1);
      ia[e] = in_a[c];
      ib[e] = in_b[c];
    }
  }
  for (int i0 = threadIdx.x; i0 < total; i0 += 256 * 8) {
    u32x4 v[8];
    bool ok[8];
#pragma unroll
    for (int b8 = 0; b8 < 8; ++b8) {
      const int i = min(i0 + b8 * 256, total - 1);
      int j, pix, r, xx;
      dcpp.divmod(i, pix, j);
      dwp.divmod(pix, r, xx);
      const int iy = iy0 + r, ix = xx - 1, c = cs0 + j * 8;
      ok[b8] = iy >= 0 && iy < Hi && ix >= 0 && ix < Wi && c < C;
      v[b8] = *reinterpret_cast<const u32x4*>(ok[b8] ? xin + ((long)iy * Wi + ix) * C + c : xin);   // branch-free
    }
    TD_ISSUE_FENCE();
#pragma unroll
    for (int b8 = 0; b8 < 8; ++b8) {
      const int i = i0 + b8 * 256;
      if (i < total) {
        int j, pix;
        dcpp.divmod(i, pix, j);
        if constexpr (AFF) {
          const bf16x8 t8 = *reinterpret_cast<const bf16x8*>(&v[b8]);
          bf16x8 o8;
#pragma unroll
          for (int e = 0; e < 8; ++e) o8[e] = (bf16_t)fmaxf(fmaf((float)t8[e], ia[e], ib[e]), 0.f);
          v[b8] = *reinterpret_cast<const u32x4*>(&o8);
        }
        *reinterpret_cast<u32x4*>(tile + (long)pix * PS + j * 16) = ok[b8] ? v[b8] : (u32x4){0u, 0u, 0u, 0u};
      }
    }
  }
  __syncthreads();
  gconv_band_mma<STRIDE>(gw_, tile, red, redq, Wi, C, wfrag, scale, shift, y, pooled, pooled_sq, Ho, Wo, nbands, CSP, PS, relu, n, bnd,
                         slab, oy0, nrows_out, bst);
}

// conv1 (1x1 + BN + ReLU, optionally behind the gate-shift splice) IN FRONT of the grouped 3x3 of the same bottleneck, one
// launch: the band of y1 rows that gconv3x3_mfma_kernel stages from HBM is COMPUTED here from the block input x -- each
// workgroup contracts the pixels of its band with the 64 conv1 output channels of its slab (a grouped conv only reads its
// own group's channels, so a slab needs no other) and writes them, BN + ReLU applied, into the zero-haloed LDS band.  The
// y1 map (at the block's INPUT resolution: the largest intermediate of a stride-2 block, 281 MB for s2.b1 of RegNetY-200MF
// at 800 frames, written once and read 2.25 x) never exists.  x rows are MFMA B operands straight from global memory
// (lane = pixel l & 15, k-chunk l >> 4), the conv1 weights of the slab are A-operand fragments in registers
// (w1f: [slabs * CSP / 16][KS1][64], engine.pack_mfma_frags, zero padded).
// (A form that also produced the downsample shortcut from the same x fragments was measured slower and is parked:
// experiments/r4_parked/conv_with_c1_gconv_ds.hip.)
template <int STRIDE, int KS1>
__global__ __launch_bounds__(256, KS1 == 4 ? 3 : 1) void c1_gconv_mfma_kernel(const bf16_t* __restrict__ x, const bf16_t* __restrict__ G, int Fp,
                                                            int Hi, int Wi, int Cin, int C, const bf16x8* __restrict__ w1f,
                                                            const float* __restrict__ s1, const float* __restrict__ h1,
                                                            const bf16x8* __restrict__ wfrag, const float* __restrict__ scale,
                                                            const float* __restrict__ shift, bf16_t* __restrict__ y,
                                                            float* __restrict__ pooled, int Ho, int Wo, int band, int nbands,
                                                            int CSP, int PS, int rows_in, int relu, long long* dbg) {
  extern __shared__ __attribute__((aligned(16))) unsigned char tile[];
  __shared__ float red[4][32];
  __shared__ float redq[4][32];
  GC_STAMP(0);
  const long lid = xcd_logical_id(blockIdx.x, gridDim.x);
  const int nslabs_ = (C + CSP - 1) >> (CSP == 64 ? 6 : (CSP == 32 ? 5 : 4));        // CSP is 16, 32 or 64
  int slab, bnd, n, t_;
  td_split(lid, nslabs_, t_, slab);
  td_split(t_, nbands, n, bnd);
  const int oy0 = bnd * band;
  const int nrows_out = min(band, Ho - oy0);
  const int WP = Wi + 2;
  const int iy0 = oy0 * STRIDE - 1;
  const int nrow_used = (nrows_out - 1) * STRIDE + 3;
  const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int pl = lane & 15, q = lane >> 4;
  const int nts = CSP >> 4;                                // conv1 output tiles of this slab (<= 4)
  GcW gw_;                                                 // the grouped conv's weights travel under conv1 -- where the
  if constexpr (KS1 <= 2) gw_ = gconv_load_w(wfrag, slab, CSP);   // registers allow (at KS1 >= 4 they cost the third workgroup per CU)
  // conv1 weights of the slab, BN affine of this lane's 4 channels per tile
  bf16x8 w1r[4][KS1];
  float a1[4][4], b1[4][4];
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    const int tt = slab * nts + min(t, nts - 1);
#pragma unroll
    for (int ks = 0; ks < KS1; ++ks) w1r[t][ks] = w1f[((long)tt * KS1 + ks) * 64 + lane];
    // this lane's 4 channels of the tile as ONE 16-byte load each (round 6: as 8 dword loads per tile the prologue was 32
    // vector-memory instructions per wave -- time stamps put 2.6 us of a ~9 us workgroup in front of its first x load, and a
    // CU's memory front end serves three such workgroups at a time); C is a multiple of 8: a group of 4 is inside or outside
    const int c4 = slab * CSP + t * 16 + 4 * q;
    const bool ok = t < nts && c4 < C;
    const int cc4 = min(c4, C - 4);
    const f32x4 av = *reinterpret_cast<const f32x4*>(s1 + cc4), bv = *reinterpret_cast<const f32x4*>(h1 + cc4);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      a1[t][e] = ok ? av[e] : 0.f;                          // channels beyond C: exact zeros
      b1[t][e] = ok ? bv[e] : 0.f;
    }
  }
  // ---- zero the halo columns of every band row and the rows that fall outside the map
  {
    const int cpp = PS >> 4;
    for (int i = tid; i < nrow_used * 2 * cpp; i += 256) {
      const int j = i % cpp, rc = i / cpp;
      const int rr = rc >> 1, col = (rc & 1) ? (Wi + 1) : 0;
      *reinterpret_cast<u32x4*>(tile + ((long)rr * WP + col) * PS + j * 16) = (u32x4){0u, 0u, 0u, 0u};
    }
    for (int rr = 0; rr < nrow_used; ++rr) {
      const int r = iy0 + rr;
      if (r >= 0 && r < Hi) continue;
      for (int i = tid; i < WP * cpp; i += 256)
        *reinterpret_cast<u32x4*>(tile + (long)rr * WP * PS + (long)i * 16) = (u32x4){0u, 0u, 0u, 0u};
    }
    // the 16 pad bytes behind the channels of every interior pixel are never read (k-slot offsets stay below CSP * 2)
  }
  GC_STAMP(1);
  // ---- conv1 over the band's pixels (rows inside the map), 16 pixels per MFMA tile, tiles dealt over the four waves
  {
    const int r_lo = max(iy0, 0), r_hi = min(iy0 + nrow_used, Hi);
    const int npx = (r_hi - r_lo) * Wi;
    const int ntl = (npx + 15) >> 4;
    const IDiv dwi(Wi);
    const bf16_t* xn = x + (long)n * Hi * Wi * Cin;
    const bf16_t* gn = G ? G + (long)n * Hi * Wi * Fp : nullptr;
    // A wave's tiles go in groups of NPF: the x fragments of the whole group are requested before the first MFMA, so a group
    // costs ONE exposed memory round trip instead of one per tile (narrow inputs: s2.b1 of RegNetY-800MF is 44 800 workgroups
    // of ~0.4 us of MFMA work whose four to five dependent load -> MFMA -> LDS-store rounds per wave were the launch's time).
    // At KS1 >= 4 a second set of fragments costs the third workgroup per CU (measured slower, also with the BatchNorm fold
    // moved to LDS): those stay one tile at a time.
    constexpr int NPF = KS1 <= 2 ? 4 : 1;
    if constexpr (KS1 == 4) {
      // Cin = 128: one tile's fragments are 16 registers and a wave has ~5 tiles, i.e. five dependent load -> MFMA -> LDS-store
      // rounds (time stamps: 6.7 - 8 us of a 12 - 17 us workgroup).  The NEXT tile's fragments are requested before the current
      // tile is multiplied, UNCONDITIONALLY (the tile index is clamped: the last request is a repeat nobody uses) -- behind a
      // branch the compiler cannot count the loads in flight and waits for all of them (vmcnt(0)), which is what made the first
      // attempt slower.  Two alternating register sets, no copies.
      auto tile_px = [&](int t0, int& rr, int& cc) {
        const int p = min(t0, ntl - 1) * 16 + pl;
        const bool ok = t0 < ntl && p < npx;
        dwi.divmod(p < npx ? p : 0, rr, cc);
        return ok;
      };
      auto tile_load = [&](int rr, int cc, bf16x8 (&xfo)[KS1]) {
        const long pix = (long)(r_lo + rr) * Wi + cc;
#pragma unroll
        for (int ks = 0; ks < KS1; ++ks) {
          const int k = 32 * ks + 8 * q;
          const bf16_t* src = (gn && k < Fp) ? gn + pix * Fp + k : xn + pix * Cin + k;
          xfo[ks] = *reinterpret_cast<const bf16x8*>(src);           // Cin = 128: every k-step is inside the row
        }
      };
      auto tile_mma = [&](bool pok_, int rr, int cc, const bf16x8 (&xfi)[KS1]) {
        unsigned char* dst = tile + ((long)(r_lo + rr - iy0) * WP + cc + 1) * PS + 8 * q;
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          if (t < nts) {
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < KS1; ++ks) acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w1r[t][ks], xfi[ks], acc, 0, 0, 0);
            if (pok_) {
              bf16x4 o;
#pragma unroll
              for (int e = 0; e < 4; ++e) o[e] = (bf16_t)fmaxf(acc[e] * a1[t][e] + b1[t][e], 0.f);
              *reinterpret_cast<bf16x4*>(dst + t * 32) = o;
            }
          }
        }
      };
      bf16x8 xa[KS1], xb[KS1];
      int ra, ca, rb, cb;
      bool pa = tile_px(wv, ra, ca), pb;
      tile_load(ra, ca, xa);
      for (int tb = wv; tb < ntl; tb += 8) {
        pb = tile_px(tb + 4, rb, cb);
        tile_load(rb, cb, xb);                                      // (always: see above)
        tile_mma(pa, ra, ca, xa);
        pa = tile_px(tb + 8, ra, ca);
        tile_load(ra, ca, xa);
        tile_mma(pb, rb, cb, xb);
      }
    } else
    for (int tb = wv; tb < ntl; tb += 4 * NPF) {
      bf16x8 xf[NPF][KS1];
      int rrs[NPF], ccs[NPF];
      bool poks[NPF];
#pragma unroll
      for (int g = 0; g < NPF; ++g) {
        const int t0 = tb + 4 * g;
        const int p = t0 * 16 + pl;
        poks[g] = t0 < ntl && p < npx;
        dwi.divmod(poks[g] ? p : 0, rrs[g], ccs[g]);
        const long pix = (long)(r_lo + rrs[g]) * Wi + ccs[g];
#pragma unroll
        for (int ks = 0; ks < KS1; ++ks) {
          const int k = 32 * ks + 8 * q;
          const bool ok = poks[g] && k < Cin;
          const bf16_t* src = (gn && k < Fp) ? gn + pix * Fp + k : xn + pix * Cin + k;
          const u32x4 v = *reinterpret_cast<const u32x4*>(ok ? src : xn);
          const u32x4 z = {0u, 0u, 0u, 0u};
          const u32x4 w = ok ? v : z;
          xf[g][ks] = *reinterpret_cast<const bf16x8*>(&w);
        }
      }
#pragma unroll
      for (int g = 0; g < NPF; ++g) {
        if (tb + 4 * g >= ntl) break;                             // (wave-uniform)
        const bool pok = poks[g];
        unsigned char* dst = tile + ((long)(r_lo + rrs[g] - iy0) * WP + ccs[g] + 1) * PS + 8 * q;
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          if (t < nts) {
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < KS1; ++ks) acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w1r[t][ks], xf[g][ks], acc, 0, 0, 0);
            if (pok) {
              bf16x4 o;
#pragma unroll
              for (int e = 0; e < 4; ++e) o[e] = (bf16_t)fmaxf(acc[e] * a1[t][e] + b1[t][e], 0.f);
              *reinterpret_cast<bf16x4*>(dst + t * 32) = o;
            }
          }
        }
      }
    }
  }
  GC_STAMP(2);
  if constexpr (KS1 > 2) gw_ = gconv_load_w(wfrag, slab, CSP);
  __syncthreads();
  GC_STAMP(3);
  gconv_band_mma<STRIDE>(gw_, tile, red, redq, Wi, C, wfrag, scale, shift, y, pooled, nullptr, Ho, Wo, nbands, CSP, PS, relu, n, bnd,
                         slab, oy0, nrows_out, GcStat{nullptr, nullptr, nullptr, nullptr}, dbg);
}

static long long* g_c1g_dbg = nullptr;
extern "C" int tdeed_c1_gconv_set_debug(void* buf) { g_c1g_dbg = (long long*)buf; return TDEED_OK; }

// 1 if the bf16 MFMA kernel serves this geometry (a band of input rows fits LDS), i.e. if wfrag / in_a are usable
extern "C" int tdeed_gconv3x3_mfma_fits(int Hi, int Wi, int C, int stride) {
  return gc_geom(Hi, Wi, C, stride).band > 0 ? 1 : 0;
}

extern "C" int tdeed_gconv3x3_parts(int Hi, int Wi, int C, int stride, int dtype) {
  if (dtype != TDEED_BF16) return 1;
  GcGeom g = gc_geom(Hi, Wi, C, stride);
  return g.band > 0 ? g.nbands : 1;
}

static int gconv3x3_launch(const void* x, int N, int Hi, int Wi, int C, int gw, int stride, const float* w, const void* wfrag,
                           const float* scale, const float* shift, void* y, float* pooled, float* pooled_sq, const float* in_a,
                           const float* in_b, int relu, int dtype, void* stream, const GcStat bst);
extern "C" int tdeed_gconv3x3_fwd(const void* x, int N, int Hi, int Wi, int C, int gw, int stride,
                                  const float* w, const void* wfrag, const float* scale, const float* shift,
                                  void* y, float* pooled, float* pooled_sq, const float* in_a, const float* in_b, int relu,
                                  int dtype, void* stream) {
  return gconv3x3_launch(x, N, Hi, Wi, C, gw, stride, w, wfrag, scale, shift, y, pooled, pooled_sq, in_a, in_b, relu, dtype,
                         stream, GcStat{nullptr, nullptr, nullptr, nullptr});
}
// Stride-1 input gradient of a training bottleneck's conv2 (a grouped conv of dy with the flipped / transposed weights, wfrag_t)
// that also leaves the statistics of conv1's BatchNorm backward: part_s / part_q fp32 [N][tdeed_gconv3x3_parts()][C] = per
// (frame, band) sums of g and g * (bz - bmean), g = dx masked by [bfa * bz + bfb > 0]  (bz: conv1's raw output, like dx).  bf16.
extern "C" int tdeed_gconv3x3_dgrad_stats(const void* dy, int N, int Hi, int Wi, int C, int gw, const void* wfrag_t,
                                          const float* one, const float* zero, void* dx, const void* bz, const float* bfa,
                                          const float* bfb, const float* bmean, float* part_s, float* part_q, void* stream) {
  TD_CHECK(bz && bfa && bfb && bmean && part_s && part_q && wfrag_t && one && zero, "gconv3x3_dgrad_stats: null pointer");
  TD_CHECK(tdeed_gconv3x3_mfma_fits(Hi, Wi, C, 1), "gconv3x3_dgrad_stats: %dx%dx%d is not served by the MFMA kernel", Hi, Wi, C);
  return gconv3x3_launch(dy, N, Hi, Wi, C, gw, 1, nullptr, wfrag_t, one, zero, dx, part_s, part_q, nullptr, nullptr, 0,
                         TDEED_BF16, stream, GcStat{(const bf16_t*)bz, bfa, bfb, bmean});
}
static int gconv3x3_launch(const void* x, int N, int Hi, int Wi, int C, int gw, int stride, const float* w, const void* wfrag,
                           const float* scale, const float* shift, void* y, float* pooled, float* pooled_sq, const float* in_a,
                           const float* in_b, int relu, int dtype, void* stream, const GcStat bst) {
  TD_CHECK(x && scale && shift && y && pooled, "gconv3x3: null pointer");
  TD_CHECK(!in_a == !in_b, "gconv3x3: in_a and in_b come together");
  TD_CHECK(!in_a || (dtype == TDEED_BF16 && wfrag), "gconv3x3: the on-load input affine exists in the bf16 MFMA kernel only");
  TD_CHECK(!pooled_sq || (dtype == TDEED_BF16 && wfrag), "gconv3x3: sums of squares come from the bf16 MFMA kernel only");
  TD_CHECK((gw == 8 || gw == 16) && C % gw == 0, "gconv3x3: group width %d / C %d unsupported", gw, C);
  TD_CHECK(stride == 1 || stride == 2, "gconv3x3: stride %d", stride);
  TD_CHECK(N > 0 && N <= 65535 && Hi > 0 && Wi > 0, "gconv3x3: bad sizes");
  hipStream_t st = (hipStream_t)stream;
  if (dtype == TDEED_F32) {
    TD_CHECK(w, "gconv3x3: fp32 path needs the packed fp32 weights");
    return gw == 8 ? launch_gconv<float, 8>(x, N, Hi, Wi, C, stride, w, scale, shift, y, pooled, relu, st)
                   : launch_gconv<float, 16>(x, N, Hi, Wi, C, stride, w, scale, shift, y, pooled, relu, st);
  }
  if (dtype == TDEED_BF16) {
    GcGeom g = gc_geom(Hi, Wi, C, stride);
    if (!wfrag || g.band <= 0) {     // no MFMA fragments given (or a row does not fit LDS): VALU kernel
      TD_CHECK(w, "gconv3x3: no weights");
      TD_CHECK(!in_a, "gconv3x3: a row of %d px x %d ch does not fit LDS; no on-load input affine on the VALU path", Wi, C);
      return gw == 8 ? launch_gconv<bf16_t, 8>(x, N, Hi, Wi, C, stride, w, scale, shift, y, pooled, relu, st)
                     : launch_gconv<bf16_t, 16>(x, N, Hi, Wi, C, stride, w, scale, shift, y, pooled, relu, st);
    }
    const int Ho = (Hi - 1) / stride + 1, Wo = (Wi - 1) / stride + 1;
    dim3 grid((unsigned)((long)g.nbands * g.nslabs * N));
    size_t smem = (size_t)g.rows_in * (Wi + 2) * g.PS;
#define TD_GCF(Sv, Av)                                                                                                   \
  hipLaunchKernelGGL((gconv3x3_mfma_kernel<Sv, Av>), grid, dim3(256), smem, st, (const bf16_t*)x, Hi, Wi, C, in_a, in_b,      \
                     (const bf16x8*)wfrag, scale, shift, (bf16_t*)y, pooled, pooled_sq, Ho, Wo, g.band, g.nbands, g.CSP, g.PS, \
                     g.rows_in, relu, bst)
    if (stride == 1) { if (in_a) TD_GCF(1, true); else TD_GCF(1, false); }
    else { if (in_a) TD_GCF(2, true); else TD_GCF(2, false); }
#undef TD_GCF
    TD_LAUNCH_CHECK("gconv3x3_mfma");
    return TDEED_OK;
  }
  tdeed_set_error("gconv3x3: bad dtype %d", dtype);
  return TDEED_ERR_ARG;
}

// conv1 + grouped 3x3 of one bottleneck in one launch (bf16).  x [N][Hi][Wi][Cin]; G optional [N*Hi*Wi][Fp] compact
// gate-shift output replacing channels [0, Fp) of x; w1f: conv1 weight [C][Cin] as MFMA A fragments padded to whole slabs
// ([nslabs * CSP / 16][ceil(Cin / 32)][64][8]); the rest as tdeed_gconv3x3_fwd.  tdeed_c1_gconv_fits: Cin <= 64 and a band of
// rows fits LDS.
extern "C" int tdeed_c1_gconv_fits(int Hi, int Wi, int Cin, int C, int stride) {
  const int ks1 = (Cin + 31) / 32;
  if (Cin % 8 != 0 || Cin < 8 || !(ks1 == 1 || ks1 == 2 || ks1 == 4 || ks1 == 5) || C % 8 != 0) return 0;
  return gc_geom(Hi, Wi, C, stride).band > 0 ? 1 : 0;
}
extern "C" int tdeed_c1_gconv_slab_tiles(int Hi, int Wi, int C, int stride) {     // rows of w1f: nslabs * CSP / 16
  const GcGeom g = gc_geom(Hi, Wi, C, stride);
  return g.nslabs * (g.CSP >> 4);
}
extern "C" int tdeed_c1_gconv_fwd(const void* x, const void* G, int Fp, int N, int Hi, int Wi, int Cin, int C, int gw,
                                  int stride, const void* w1f, const float* s1, const float* h1, const void* wfrag,
                                  const float* scale, const float* shift, void* y, float* pooled, void* stream) {
  TD_CHECK(x && w1f && s1 && h1 && wfrag && scale && shift && y && pooled, "c1_gconv: null pointer");
  TD_CHECK((gw == 8 || gw == 16) && C % gw == 0, "c1_gconv: group width %d / C %d unsupported", gw, C);
  TD_CHECK(stride == 1 || stride == 2, "c1_gconv: stride %d", stride);
  TD_CHECK(N > 0 && N <= 65535 && tdeed_c1_gconv_fits(Hi, Wi, Cin, C, stride), "c1_gconv: Hi=%d Wi=%d Cin=%d C=%d unsupported",
           Hi, Wi, Cin, C);
  TD_CHECK(!G || (Fp % 8 == 0 && Fp > 0 && Fp <= Cin), "c1_gconv: bad splice width %d", Fp);
  const GcGeom g = gc_geom(Hi, Wi, C, stride);
  const int Ho = (Hi - 1) / stride + 1, Wo = (Wi - 1) / stride + 1;
  dim3 grid((unsigned)((long)g.nbands * g.nslabs * N));
  const size_t smem = (size_t)g.rows_in * (Wi + 2) * g.PS;
  hipStream_t st = (hipStream_t)stream;
  const int KS1 = (Cin + 31) / 32;
#define TD_C1G(Sv, Kv)                                                                                                     \
  hipLaunchKernelGGL((c1_gconv_mfma_kernel<Sv, Kv>), grid, dim3(256), smem, st, (const bf16_t*)x, (const bf16_t*)G,           \
                     G ? Fp : 0, Hi, Wi, Cin, C, (const bf16x8*)w1f, s1, h1, (const bf16x8*)wfrag, scale, shift, (bf16_t*)y,  \
                     pooled, Ho, Wo, g.band, g.nbands, g.CSP, g.PS, g.rows_in, 1, g_c1g_dbg)
#define TD_C1G_K(Sv)                                                                                                       \
  do {                                                                                                                     \
    if (KS1 == 1) TD_C1G(Sv, 1); else if (KS1 == 2) TD_C1G(Sv, 2); else if (KS1 == 4) TD_C1G(Sv, 4);                      \
    else TD_C1G(Sv, 5);                                                                                                    \
  } while (0)
  if (stride == 2) TD_C1G_K(2); else TD_C1G_K(1);
#undef TD_C1G_K
#undef TD_C1G
  TD_LAUNCH_CHECK("c1_gconv");
  return TDEED_OK;
}

// =========================================================================== SE excitation
// FPB frames per block share every weight read.  Both phases are thread-per-output with the weights
// transposed on the host (w1t [C][R], w2t [R][C]) so adjacent threads read adjacent addresses, and
// deeply unrolled so every thread keeps >= 8 independent L2 loads in flight (the kernel is pure
// latency: 270 KB of weights, 800 frames).  Phase 1 additionally splits C over 256/R thread slices.
// pooled holds SUMS over pixels in n_parts partial rows per frame ([N][n_parts][C]); inv_cnt = 1/(Ho*Wo).
#define SE_FPB 4
__global__ __launch_bounds__(256) void se_gate_kernel(const float* __restrict__ pooled, int n_parts, float inv_cnt,
                                                      int N, int C, int R, const float* __restrict__ w1t,
                                                      const float* __restrict__ b1,
                                                      const float* __restrict__ w2t,
                                                      const float* __restrict__ b2, float* __restrict__ gate) {
  extern __shared__ float sm[];   // [FPB][C] pooled means, [FPB][R] hidden
  float* sp = sm;
  float* shid = sm + SE_FPB * C;
  const int f0 = blockIdx.x * SE_FPB;
  for (int i = threadIdx.x; i < SE_FPB * C; i += 256) {
    const int f = i / C, c = i - f * C;
    float v = 0.f;
    if (f0 + f < N) {
      const float* src = pooled + ((long)(f0 + f) * n_parts) * C + c;
      for (int q = 0; q < n_parts; ++q) v += src[(long)q * C];
    }
    sp[i] = v * inv_cnt;
  }
  __syncthreads();
  // phase 1: thread (j, slice) walks its slice of C with independent coalesced loads of w1t[c][j]
  {
    const int RP = R <= 32 ? 32 : (R <= 64 ? 64 : (R <= 128 ? 128 : 256));
    const int nsl = 256 / RP;                       // C slices handled in parallel
    const int j = threadIdx.x % RP, sl = threadIdx.x / RP;
    float a[SE_FPB];
#pragma unroll
    for (int f = 0; f < SE_FPB; ++f) a[f] = 0.f;
    if (j < R) {
      const int cper = (C + nsl - 1) / nsl;
      const int c0 = sl * cper, c1 = min(C, c0 + cper);
#pragma unroll 8
      for (int c = c0; c < c1; ++c) {
        const float wv_ = w1t[(long)c * R + j];
#pragma unroll
        for (int f = 0; f < SE_FPB; ++f) a[f] = fmaf(sp[f * C + c], wv_, a[f]);
      }
    }
    float* part = shid + SE_FPB * R;                // [nsl][FPB][R]
    if (j < R) {
#pragma unroll
      for (int f = 0; f < SE_FPB; ++f) part[(sl * SE_FPB + f) * R + j] = a[f];
    }
    __syncthreads();
    for (int i = threadIdx.x; i < SE_FPB * R; i += 256) {
      float v = 0.f;
      for (int q = 0; q < nsl; ++q) v += part[q * SE_FPB * R + i];
      shid[i] = fmaxf(v + b1[i % R], 0.f);
    }
  }
  __syncthreads();
  for (int c = threadIdx.x; c < C; c += 256) {
    float a[SE_FPB];
#pragma unroll
    for (int f = 0; f < SE_FPB; ++f) a[f] = 0.f;
#pragma unroll 8
    for (int j = 0; j < R; ++j) {
      const float wv_ = w2t[(long)j * C + c];
#pragma unroll
      for (int f = 0; f < SE_FPB; ++f) a[f] = fmaf(shid[f * R + j], wv_, a[f]);
    }
#pragma unroll
    for (int f = 0; f < SE_FPB; ++f)
      if (f0 + f < N) gate[(long)(f0 + f) * C + c] = sigmoidf_(a[f] + b2[c]);
  }
}

extern "C" int tdeed_se_gate_fwd(const float* pooled, int n_parts, float inv_cnt, int N, int C, int R,
                                 const float* w1t, const float* b1, const float* w2t, const float* b2, float* gate,
                                 void* stream) {
  TD_CHECK(pooled && w1t && b1 && w2t && b2 && gate, "se_gate: null pointer");
  TD_CHECK(N > 0 && C > 0 && R > 0 && R <= 256 && n_parts > 0, "se_gate: bad sizes");
  const int RP = R <= 32 ? 32 : (R <= 64 ? 64 : (R <= 128 ? 128 : 256));
  size_t smem = (size_t)SE_FPB * (C + R + (256 / RP) * R) * sizeof(float);
  hipLaunchKernelGGL(se_gate_kernel, dim3(cdiv(N, SE_FPB)), dim3(256), smem, (hipStream_t)stream, pooled, n_parts,
                     inv_cnt, N, C, R, w1t, b1, w2t, b2, gate);
  TD_LAUNCH_CHECK("se_gate");
  return TDEED_OK;
}

// ---- bf16-weight variant (throughput mode): 16-byte weight loads (8 outputs each), every thread issues its
// whole share of a weight matrix as ONE batch of independent loads, so each phase costs one L2 round trip.
// w1p: bf16 [C][R8] (fc1.weight^T, R padded to 8), w2p: bf16 [R][C] (fc2.weight^T).
__global__ __launch_bounds__(256) void se_gate_bf16_kernel(const float* __restrict__ pooled, int n_parts, float inv_cnt,
                                                           int N, int C, int R, const bf16_t* __restrict__ w1p,
                                                           const float* __restrict__ b1,
                                                           const bf16_t* __restrict__ w2p,
                                                           const float* __restrict__ b2, float* __restrict__ gate) {
  extern __shared__ float sm[];
  const int R8 = (R + 7) & ~7;
  float* sp = sm;                                   // [FPB][C] means
  float* shid = sp + SE_FPB * C;                    // [FPB][R8]
  float* part = shid + SE_FPB * R8;                 // partial sums of either phase
  const int f0 = blockIdx.x * SE_FPB;
  constexpr int MAXB = 20;
  // thread roles of both phases; BOTH weight shares are requested before anything else so that the pooled
  // sums, fc1 and fc2 weights travel in one memory round trip (the kernel is pure latency)
  const int NJ = R8 >> 3, nsl1 = 256 / NJ;
  const int jo = threadIdx.x % NJ, sl1 = threadIdx.x / NJ;
  const int cper = (C + nsl1 - 1) / nsl1;
  const int c0 = sl1 * cper, c1 = sl1 < nsl1 ? min(C, c0 + cper) : c0;
  const int NC = C >> 3;
  const int nsl2 = 256 / NC > 0 ? 256 / NC : 1;
  const int co = threadIdx.x % NC, sl2 = threadIdx.x / NC;
  const bool act2 = sl2 < nsl2 && threadIdx.x < NC * nsl2;
  const int jper = (R + nsl2 - 1) / nsl2;
  const int j0 = sl2 * jper, j1 = act2 ? min(R, j0 + jper) : j0;
  bf16x8 w1r[MAXB], w2r[MAXB];
  // branch-free (clamped rows: a guarded load compiles to load / wait per item); unused registers are never read
#pragma unroll
  for (int i = 0; i < MAXB; ++i)
    w1r[i] = *reinterpret_cast<const bf16x8*>(w1p + (long)min(c0 + i, C - 1) * R8 + min(jo, NJ - 1) * 8);
#pragma unroll
  for (int i = 0; i < MAXB; ++i)
    w2r[i] = *reinterpret_cast<const bf16x8*>(w2p + (long)min(j0 + i, R - 1) * C + co * 8);
  for (int i = threadIdx.x; i < SE_FPB * C; i += 256) {
    const int f = i / C, c = i - f * C;
    const float* src = pooled + ((long)min(f0 + f, N - 1) * n_parts) * C + c;
    float v = 0.f;
    if (n_parts == 1) {
      v = src[0];
    } else {
      for (int q0 = 0; q0 < n_parts; q0 += 8) {
        float pv[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) pv[q] = src[(long)min(q0 + q, n_parts - 1) * C];
#pragma unroll
        for (int q = 0; q < 8; ++q) v += q0 + q < n_parts ? pv[q] : 0.f;
      }
    }
    sp[i] = v * inv_cnt;
  }
  __syncthreads();
  {
    float a[SE_FPB][8];
#pragma unroll
    for (int f = 0; f < SE_FPB; ++f)
#pragma unroll
      for (int e = 0; e < 8; ++e) a[f][e] = 0.f;
#pragma unroll
    for (int i = 0; i < MAXB; ++i)
      if (c0 + i < c1) {
#pragma unroll
        for (int f = 0; f < SE_FPB; ++f) {
          const float pv = sp[f * C + c0 + i];
#pragma unroll
          for (int e = 0; e < 8; ++e) a[f][e] = fmaf(pv, (float)w1r[i][e], a[f][e]);
        }
      }
    for (int cb = c0 + MAXB; cb < c1; ++cb) {          // shapes beyond the register batch (not hit by RegNetY-200/800MF)
      const bf16x8 w = *reinterpret_cast<const bf16x8*>(w1p + (long)cb * R8 + jo * 8);
#pragma unroll
      for (int f = 0; f < SE_FPB; ++f)
#pragma unroll
        for (int e = 0; e < 8; ++e) a[f][e] = fmaf(sp[f * C + cb], (float)w[e], a[f][e]);
    }
    if (sl1 < nsl1) {
#pragma unroll
      for (int f = 0; f < SE_FPB; ++f)
#pragma unroll
        for (int e = 0; e < 8; ++e) part[(sl1 * SE_FPB + f) * R8 + jo * 8 + e] = a[f][e];
    }
    __syncthreads();
    for (int i = threadIdx.x; i < SE_FPB * R8; i += 256) {
      const int j = i % R8;
      float v = 0.f;
      for (int s_ = 0; s_ < nsl1; ++s_) v += part[s_ * SE_FPB * R8 + i];
      shid[i] = j < R ? fmaxf(v + b1[j], 0.f) : 0.f;
    }
    __syncthreads();
  }
  {
    float a[SE_FPB][8];
#pragma unroll
    for (int f = 0; f < SE_FPB; ++f)
#pragma unroll
      for (int e = 0; e < 8; ++e) a[f][e] = 0.f;
#pragma unroll
    for (int i = 0; i < MAXB; ++i)
      if (j0 + i < j1) {
#pragma unroll
        for (int f = 0; f < SE_FPB; ++f) {
          const float hv = shid[f * R8 + j0 + i];
#pragma unroll
          for (int e = 0; e < 8; ++e) a[f][e] = fmaf(hv, (float)w2r[i][e], a[f][e]);
        }
      }
    for (int jb = j0 + MAXB; jb < j1; ++jb) {
      const bf16x8 w = *reinterpret_cast<const bf16x8*>(w2p + (long)jb * C + co * 8);
#pragma unroll
      for (int f = 0; f < SE_FPB; ++f)
#pragma unroll
        for (int e = 0; e < 8; ++e) a[f][e] = fmaf(shid[f * R8 + jb], (float)w[e], a[f][e]);
    }
    if (act2) {
#pragma unroll
      for (int f = 0; f < SE_FPB; ++f)
#pragma unroll
        for (int e = 0; e < 8; ++e) part[(sl2 * SE_FPB + f) * C + co * 8 + e] = a[f][e];
    }
    __syncthreads();
    for (int i = threadIdx.x; i < SE_FPB * C; i += 256) {
      const int f = i / C, c = i - f * C;
      float v = 0.f;
      for (int s_ = 0; s_ < nsl2; ++s_) v += part[s_ * SE_FPB * C + i];
      if (f0 + f < N) gate[(long)(f0 + f) * C + c] = sigmoidf_(v + b2[c]);
    }
  }
}

extern "C" int tdeed_se_gate_bf16_fwd(const float* pooled, int n_parts, float inv_cnt, int N, int C, int R,
                                      const void* w1p, const float* b1, const void* w2p, const float* b2,
                                      float* gate, void* stream) {
  TD_CHECK(pooled && w1p && b1 && w2p && b2 && gate, "se_gate_bf16: null pointer");
  TD_CHECK(N > 0 && C > 0 && C % 8 == 0 && C <= 2048 && R > 0 && R <= 2048 && n_parts > 0, "se_gate_bf16: bad sizes");
  const int R8 = (R + 7) & ~7;
  const size_t p1 = (size_t)(256 / (R8 / 8)) * SE_FPB * R8;
  const size_t p2 = (size_t)(256 / (C / 8) > 0 ? 256 / (C / 8) : 1) * SE_FPB * C;
  const size_t smem = ((size_t)SE_FPB * (C + R8) + (p1 > p2 ? p1 : p2)) * sizeof(float);
  TD_CHECK(smem <= 64 * 1024, "se_gate_bf16: C=%d R=%d needs %zu B of LDS", C, R, smem);
  hipLaunchKernelGGL(se_gate_bf16_kernel, dim3(cdiv(N, SE_FPB)), dim3(256), smem, (hipStream_t)stream, pooled,
                     n_parts, inv_cnt, N, C, R, (const bf16_t*)w1p, b1, (const bf16_t*)w2p, b2, gate);
  TD_LAUNCH_CHECK("se_gate_bf16");
  return TDEED_OK;
}

// =========================================================================== avg-pool + pos-enc
// One workgroup per frame, 256 lanes = (pixel slice, channel chunk); every lane fetches its pixels in batches of 8
// independent loads (a plain accumulate loop costs one memory round trip per pixel), then an ordered reduce in LDS.
template <typename T, typename TO>
__global__ __launch_bounds__(256) void avgpool_posenc_kernel(const T* __restrict__ x, int T_len, int hw, int C,
                                                             const float* __restrict__ temp_enc,
                                                             TO* __restrict__ feat, float* __restrict__ rowstat,
                                                             float ln_eps) {
  constexpr int EPC = Chunk<T>::N;
  extern __shared__ float red[];       // [S][C]
  const int f = (int)xcd_logical_id(blockIdx.x, gridDim.x);    // frame = b*T + t, in the per-XCD chunks of the last block's launch
  const int t = f % T_len;
  const int nch = C / EPC;
  const int S = 256 / nch > 0 ? 256 / nch : 1;
  for (int ch = threadIdx.x % nch, s = threadIdx.x / nch; s < S && ch < nch; ch += 256) {   // one pass when nch <= 256
    const int c0 = ch * EPC;
    float a[EPC];
#pragma unroll
    for (int e = 0; e < EPC; ++e) a[e] = 0.f;
    const T* src = x + (long)f * hw * C + c0;
    for (int p0 = s; p0 < hw; p0 += S * 8) {
      float v[8][EPC];
#pragma unroll
      for (int b = 0; b < 8; ++b) Chunk<T>::load(src + (long)min(p0 + b * S, hw - 1) * C, v[b]);
#pragma unroll
      for (int b = 0; b < 8; ++b)
        if (p0 + b * S < hw) {
#pragma unroll
          for (int e = 0; e < EPC; ++e) a[e] += v[b][e];
        }
    }
#pragma unroll
    for (int e = 0; e < EPC; ++e) red[s * C + c0 + e] = a[e];
  }
  __syncthreads();
  float rs1 = 0.f, rs2 = 0.f;
  for (int ch = threadIdx.x; ch < nch; ch += 256) {
    const int c0 = ch * EPC;
    float a[EPC];
#pragma unroll
    for (int e = 0; e < EPC; ++e) {
      float v = 0.f;
      for (int s = 0; s < S; ++s) v += red[s * C + c0 + e];
      a[e] = v / (float)hw + temp_enc[(long)t * C + c0 + e];
    }
    // (TO = float under a bf16 trunk: the temporal stage keeps its residual stream in fp32, round 5)
    constexpr int EPO = Chunk<TO>::N;
#pragma unroll
    for (int g = 0; g < EPC / EPO; ++g) {
      float o[EPO];
#pragma unroll
      for (int e = 0; e < EPO; ++e) o[e] = a[g * EPO + e];
      Chunk<TO>::store(feat + (long)f * C + c0 + g * EPO, o);
    }
#pragma unroll
    for (int e = 0; e < EPC; ++e) {
      const float v = round_to<TO>(a[e]);
      rs1 += v;
      rs2 = fmaf(v, v, rs2);
    }
  }
  if (rowstat) {
    // LayerNorm statistics of the stored row (mean, rstd over C; modules.py:353-357) for the first SGP block's front kernel
    __shared__ float scr[8];
    const float t1 = block_sum<4>(rs1, scr);
    const float t2 = block_sum<4>(rs2, scr + 4);
    if (threadIdx.x == 0) {
      const float m = t1 / (float)C;
      rowstat[(long)f * 2] = m;
      rowstat[(long)f * 2 + 1] = 1.0f / sqrtf(fmaxf(t2 / (float)C - m * m, 0.f) + ln_eps);
    }
  }
}

extern "C" int tdeed_avgpool_posenc_fwd(const void* x, int B, int T, int hw, int C, const float* temp_enc,
                                        void* feat, float* rowstat, int dtype, int dtype_out, void* stream) {
  TD_CHECK(x && temp_enc && feat, "avgpool: null pointer");
  TD_CHECK(dtype_out == dtype || dtype_out == TDEED_F32, "avgpool: feat is stored in the input's type or in fp32");
  TD_CHECK(B > 0 && T > 0 && hw > 0 && C > 0 && C % 8 == 0 && C <= 2048, "avgpool: bad sizes");
  hipStream_t st = (hipStream_t)stream;
  if (dtype == TDEED_F32) {
    const int nch = C / 4, S = 256 / nch > 0 ? 256 / nch : 1;
    hipLaunchKernelGGL((avgpool_posenc_kernel<float, float>), dim3(B * T), dim3(256), (size_t)S * C * sizeof(float), st,
                       (const float*)x, T, hw, C, temp_enc, (float*)feat, rowstat, 1e-5f);
  } else if (dtype == TDEED_BF16) {
    const int nch = C / 8, S = 256 / nch > 0 ? 256 / nch : 1;
    if (dtype_out == TDEED_F32)
      hipLaunchKernelGGL((avgpool_posenc_kernel<bf16_t, float>), dim3(B * T), dim3(256), (size_t)S * C * sizeof(float), st,
                         (const bf16_t*)x, T, hw, C, temp_enc, (float*)feat, rowstat, 1e-5f);
    else
      hipLaunchKernelGGL((avgpool_posenc_kernel<bf16_t, bf16_t>), dim3(B * T), dim3(256), (size_t)S * C * sizeof(float), st,
                         (const bf16_t*)x, T, hw, C, temp_enc, (bf16_t*)feat, rowstat, 1e-5f);
  } else { tdeed_set_error("avgpool: bad dtype %d", dtype); return TDEED_ERR_ARG; }
  TD_LAUNCH_CHECK("avgpool_posenc");
  return TDEED_OK;
}

// =========================================================================== SE excitation on the MFMA pipe
// 16 frames per workgroup are the 16 "pixels" of an MFMA tile: hid[r][frame] = relu(W1 . mean + b1), then
// gate[c][frame] = sigmoid(W2 . hid + b2).  Weights arrive as MFMA A-operand fragments (engine.pack_se_mfma) and
// EVERY weight / bias / pooled load of the launch is issued before the first LDS write, so the kernel is one memory
// round trip + ~50 MFMAs per wave instead of the VALU kernel's three dependent round trips and ~2000 FMAs per lane.
// The fp32 operands (pooled means, hidden units) are split hi + lo into two bf16 MFMAs: bf16 weights, fp32-accurate
// activations, the numerics of se_gate_bf16_kernel.
// NW waves; LATE2: the fc2 fragments are requested behind the fc1 MFMAs instead of up front (the wide form -- C <= 768,
// R <= 192: 48 + 36 fragments per lane do not fit the register file together; their round trip then overlaps the barrier and
// the hidden units' LDS writes).
template <int KS1M, int NT1M, int NT2M, int KS2M, int NW = 4, bool LATE2 = false>
__global__ __launch_bounds__(NW * 64) void se_gate_mfma_kernel(const float* __restrict__ pooled, int n_parts, float inv_cnt,
                                                           int N, int C, int R, const bf16x8* __restrict__ w1f,
                                                           const float* __restrict__ b1,
                                                           const bf16x8* __restrict__ w2f,
                                                           const float* __restrict__ b2, float* __restrict__ gate) {
  extern __shared__ __attribute__((aligned(16))) unsigned char sme[];
  const int KS1 = (C + 31) >> 5, RT = (R + 15) >> 4, CT = (C + 15) >> 4, KS2 = (R + 31) >> 5;
  const int PS1 = KS1 * 32 + 8, PS2 = KS2 * 32 + 8;             // row strides (elements); +8 spreads the banks
  bf16_t* Phi = reinterpret_cast<bf16_t*>(sme);                 // [16][PS1]
  bf16_t* Plo = Phi + 16 * PS1;
  bf16_t* Hhi = Plo + 16 * PS1;                                 // [16][PS2]
  bf16_t* Hlo = Hhi + 16 * PS2;
  f32x4* red = reinterpret_cast<f32x4*>(Hlo + 16 * PS2);        // [NS][16 * C/4] partial sums (n_parts > 1 only)
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, pl = lane & 15, q = lane >> 4;
  const int f0 = blockIdx.x * 16;
  // ---- issue: weights, biases (clamped indices: never a branch around a load)
  bf16x8 w1r[NT1M][KS1M], w2r[NT2M][KS2M];
  float b1v[NT1M][4], b2v[NT2M][4];
  auto load_w1 = [&]() {
#pragma unroll
    for (int a = 0; a < NT1M; ++a) {
      const int tc = min(wv + NW * a, RT - 1);
#pragma unroll
      for (int ks = 0; ks < KS1M; ++ks) w1r[a][ks] = w1f[((long)tc * KS1 + min(ks, KS1 - 1)) * 64 + lane];
    }
  };
  if constexpr (!LATE2) load_w1();        // (wide form: behind the staging of the means, whose batches need the registers)
#pragma unroll
  for (int a = 0; a < NT1M; ++a) {
    const int tc = min(wv + NW * a, RT - 1);
#pragma unroll
    for (int e = 0; e < 4; ++e) b1v[a][e] = b1[min(tc * 16 + 4 * q + e, R - 1)];
  }
  auto load_w2 = [&]() {
#pragma unroll
    for (int a = 0; a < NT2M; ++a) {
      const int tc = min(wv + NW * a, CT - 1);
#pragma unroll
      for (int ks = 0; ks < KS2M; ++ks) w2r[a][ks] = w2f[((long)tc * KS2 + min(ks, KS2 - 1)) * 64 + lane];
    }
  };
  if constexpr (!LATE2) load_w2();
#pragma unroll
  for (int a = 0; a < NT2M; ++a) {
    const int tc = min(wv + NW * a, CT - 1);
#pragma unroll
    for (int e = 0; e < 4; ++e) b2v[a][e] = b2[min(tc * 16 + 4 * q + e, C - 1)];
  }
  // ---- pooled sums -> means -> hi/lo bf16 in LDS
  const int c4n = C >> 2, nitems = 16 * c4n;
  const int NS = nitems >= NW * 64 ? 1 : min(NW * 64 / nitems, n_parts);   // part slices summed in parallel
  const IDiv dit(nitems), dc4(c4n);
  for (int i = tid; i < 16 * PS1 / 8; i += NW * 64) {                        // zero both P arrays (pad columns must be 0)
    reinterpret_cast<u32x4*>(Phi)[i] = (u32x4){0u, 0u, 0u, 0u};
    reinterpret_cast<u32x4*>(Plo)[i] = (u32x4){0u, 0u, 0u, 0u};
  }
  for (int i = tid; i < 16 * PS2 / 8; i += NW * 64) {
    reinterpret_cast<u32x4*>(Hhi)[i] = (u32x4){0u, 0u, 0u, 0u};
    reinterpret_cast<u32x4*>(Hlo)[i] = (u32x4){0u, 0u, 0u, 0u};
  }
  __syncthreads();
  if (NS == 1) {
    // wide layers (>= one float4 item per thread, few partial rows): 8 items per lane per batch, all loads of a batch in flight
    for (int i0 = tid; i0 < nitems; i0 += NW * 64 * 8) {
      f32x4 acc[8];
      int fo[8];
#pragma unroll
      for (int b = 0; b < 8; ++b) {
        acc[b] = (f32x4){0.f, 0.f, 0.f, 0.f};
        int f, c4;
        dc4.divmod(min(i0 + b * NW * 64, nitems - 1), f, c4);
        fo[b] = f * PS1 + c4 * 4;
      }
      for (int p = 0; p < n_parts; ++p) {
        f32x4 v[8];
#pragma unroll
        for (int b = 0; b < 8; ++b) {
          int f, c4;
          dc4.divmod(min(i0 + b * NW * 64, nitems - 1), f, c4);
          v[b] = *reinterpret_cast<const f32x4*>(pooled + ((long)min(f0 + f, N - 1) * n_parts + p) * C + c4 * 4);
        }
#pragma unroll
        for (int b = 0; b < 8; ++b) { acc[b][0] += v[b][0]; acc[b][1] += v[b][1]; acc[b][2] += v[b][2]; acc[b][3] += v[b][3]; }
      }
#pragma unroll
      for (int b = 0; b < 8; ++b)
        if (i0 + b * NW * 64 < nitems) {
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const float m = acc[b][e] * inv_cnt;
            const bf16_t hi = (bf16_t)m;
            Phi[fo[b] + e] = hi;
            Plo[fo[b] + e] = (bf16_t)(m - (float)hi);
          }
        }
    }
  } else {
    // narrow layers with many partial rows (s1: 56 per frame): slices of the partial rows summed in parallel
    for (int i = tid; i < nitems * NS; i += NW * 64) {
      int slice, item, f, c4;
      dit.divmod(i, slice, item);
      dc4.divmod(item, f, c4);
      const float* src = pooled + ((long)min(f0 + f, N - 1) * n_parts) * C + c4 * 4;
      f32x4 acc = {0.f, 0.f, 0.f, 0.f};
      for (int p0 = slice; p0 < n_parts; p0 += NS * 8) {
        f32x4 v[8];
#pragma unroll
        for (int b = 0; b < 8; ++b) v[b] = *reinterpret_cast<const f32x4*>(src + (long)min(p0 + b * NS, n_parts - 1) * C);
#pragma unroll
        for (int b = 0; b < 8; ++b)
          if (p0 + b * NS < n_parts) { acc[0] += v[b][0]; acc[1] += v[b][1]; acc[2] += v[b][2]; acc[3] += v[b][3]; }
      }
      red[i] = acc;
    }
  }
  if (NS > 1) {
    __syncthreads();
    for (int item = tid; item < nitems; item += NW * 64) {
      int f, c4;
      dc4.divmod(item, f, c4);
      f32x4 acc = red[item];
      for (int s_ = 1; s_ < NS; ++s_) {
        const f32x4 t = red[s_ * nitems + item];
        acc[0] += t[0]; acc[1] += t[1]; acc[2] += t[2]; acc[3] += t[3];
      }
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float m = acc[e] * inv_cnt;
        const bf16_t hi = (bf16_t)m;
        Phi[f * PS1 + c4 * 4 + e] = hi;
        Plo[f * PS1 + c4 * 4 + e] = (bf16_t)(m - (float)hi);
      }
    }
  }
  if constexpr (LATE2) load_w1();
  __syncthreads();
  // ---- phase 1: hidden units
#pragma unroll
  for (int a = 0; a < NT1M; ++a) {
    const int tile = wv + NW * a;
    if (tile < RT) {
      f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < KS1M; ++ks)
        if (ks < KS1) {
          const bf16x8 bh = *reinterpret_cast<const bf16x8*>(Phi + pl * PS1 + ks * 32 + q * 8);
          const bf16x8 bl = *reinterpret_cast<const bf16x8*>(Plo + pl * PS1 + ks * 32 + q * 8);
          acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w1r[a][ks], bh, acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w1r[a][ks], bl, acc, 0, 0, 0);
        }
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int r = tile * 16 + 4 * q + e;
        const float hv = r < R ? fmaxf(acc[e] + b1v[a][e], 0.f) : 0.f;
        const bf16_t hi = (bf16_t)hv;
        Hhi[pl * PS2 + r] = hi;
        Hlo[pl * PS2 + r] = (bf16_t)(hv - (float)hi);
      }
    }
  }
  if constexpr (LATE2) load_w2();
  __syncthreads();
  // ---- phase 2: gates
#pragma unroll
  for (int a = 0; a < NT2M; ++a) {
    const int tile = wv + NW * a;
    if (tile < CT) {
      f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < KS2M; ++ks)
        if (ks < KS2) {
          const bf16x8 bh = *reinterpret_cast<const bf16x8*>(Hhi + pl * PS2 + ks * 32 + q * 8);
          const bf16x8 bl = *reinterpret_cast<const bf16x8*>(Hlo + pl * PS2 + ks * 32 + q * 8);
          acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w2r[a][ks], bh, acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w2r[a][ks], bl, acc, 0, 0, 0);
        }
      const int c0 = tile * 16 + 4 * q;
      if (c0 < C && f0 + pl < N) {
        f32x4 g;
#pragma unroll
        for (int e = 0; e < 4; ++e) g[e] = sigmoid_fast_(acc[e] + b2v[a][e]);
        *reinterpret_cast<f32x4*>(gate + (long)(f0 + pl) * C + c0) = g;
      }
    }
  }
}

// C <= 384, R <= 96: everything in registers up front, 4 waves; up to C = 768, R = 192 (RegNetY-800MF s4): 8 waves, LATE2
extern "C" int tdeed_se_gate_mfma_fits(int C, int R) { return C % 8 == 0 && C <= 768 && R >= 1 && R <= 192; }

extern "C" int tdeed_se_gate_mfma_fwd(const float* pooled, int n_parts, float inv_cnt, int N, int C, int R,
                                      const void* w1f, const float* b1, const void* w2f, const float* b2, float* gate,
                                      void* stream) {
  TD_CHECK(pooled && w1f && b1 && w2f && b2 && gate, "se_gate_mfma: null pointer");
  TD_CHECK(N > 0 && n_parts > 0 && tdeed_se_gate_mfma_fits(C, R), "se_gate_mfma: C=%d R=%d unsupported", C, R);
  const int KS1 = (C + 31) / 32, KS2 = (R + 31) / 32;
  const bool wide = C > 384 || R > 96;
  const int thr = wide ? 512 : 256;
  const int nitems = 16 * (C / 4);
  const int NS = nitems >= thr ? 1 : (thr / nitems < n_parts ? thr / nitems : n_parts);
  const size_t smem = (size_t)2 * 16 * (KS1 * 32 + 8) * 2 + (size_t)2 * 16 * (KS2 * 32 + 8) * 2 +
                      (NS > 1 ? (size_t)NS * nitems * 16 : 0);
  if (wide) {
    static TdDevOnce attr_w;
    if (!attr_w.get()) {
      if (hipFuncSetAttribute((const void*)se_gate_mfma_kernel<24, 2, 6, 6, 8, true>, hipFuncAttributeMaxDynamicSharedMemorySize,
                              96 * 1024) != hipSuccess) {
        tdeed_set_error("se_gate_mfma: hipFuncSetAttribute failed");
        return TDEED_ERR_RUNTIME;
      }
      attr_w.set();
    }
    TD_CHECK(smem <= 96 * 1024, "se_gate_mfma: C=%d R=%d n_parts=%d need %zu B of LDS", C, R, n_parts, smem);
    hipLaunchKernelGGL((se_gate_mfma_kernel<24, 2, 6, 6, 8, true>), dim3(cdiv(N, 16)), dim3(512), smem, (hipStream_t)stream,
                       pooled, n_parts, inv_cnt, N, C, R, (const bf16x8*)w1f, b1, (const bf16x8*)w2f, b2, gate);
  } else {
    hipLaunchKernelGGL((se_gate_mfma_kernel<12, 2, 6, 3>), dim3(cdiv(N, 16)), dim3(256), smem, (hipStream_t)stream, pooled,
                       n_parts, inv_cnt, N, C, R, (const bf16x8*)w1f, b1, (const bf16x8*)w2f, b2, gate);
  }
  TD_LAUNCH_CHECK("se_gate_mfma");
  return TDEED_OK;
}
