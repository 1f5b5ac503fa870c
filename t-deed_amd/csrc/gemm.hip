// Dense 1x1 contraction on the MFMA pipe (gfx950).
//
//   C[m][n] = act((sum_k A'[m][k] W[n][k]) * scale[n] + shift[n] + R[m][n])
//
// Both operands are K-contiguous ([M][K] activations channels-last, [N][K] weights), so one
// 16-byte chunk per lane is exactly an MFMA operand fragment:
//   bf16: v_mfma_f32_16x16x32_bf16, lane l holds A[row l&15][k = 8(l>>4)+j], j<8
//   f32 : 4 x v_mfma_f32_16x16x4_f32 on the 4 floats of the chunk (same k assignment for A and B,
//         so the k permutation is harmless); exact f32 FMA chains -> parity mode.
// Tile: 128 (M) x BN (N) per 256-thread block, K in slabs of 128 bytes per row (64 bf16 / 32 f32),
// waves 2x2, each 64 x BN/2.  LDS rows are 128 B, 16-B chunks XOR-swizzled with (row & 7) so the
// ds_read_b128 fragment reads are conflict-free; one LDS stage + the next K slab prefetched in registers
// (32 KB of LDS per block => 4-5 resident blocks per CU hide the HBM/L2 latency; the staging pass is
// where the SE gate / gate-shift splice / stride-2 row gather are applied).
// Epilogue goes through LDS so that C (and the residual) move as whole 16-B chunks per lane.
#include "common.h"
#include <stdlib.h>

struct GemmP {
  const void* A; long lda;
  const void* A0; long lda0; int k0;
  const float* a_scale; int a_scale_rows;
  int M, K, N;
  const void* W; long ldw;
  const float* scale; const float* shift;
  const void* R; long ldr;
  int act;
  void* C; long ldc;
  int g_stride, g_hi, g_wi, g_ho, g_wo;
  float* colpart;      // optional [M tiles][2][N]: per-tile column sums / sums of squares of the stored C (BatchNorm statistics)
  void* C2; long ldc2; int n2;   // optional second, compact copy of columns [0, n2) of C (the next block's gate-shift slice)
  int c2_pre;                    // C2 takes the value BEFORE residual / activation and C keeps only the residual in those columns
  // BWD epilogue (tdeed_gemm_dgrad: the input-gradient contraction of a training bottleneck's conv1, whose output rows are
  // the gradient arriving at the PREVIOUS block's output ReLU):
  const void* mask; long ldmask;                      // v = mask[m][n] > 0 ? v : 0 after the residual (that ReLU's backward)
  const void* bz; long ldbz; const float* bmean;      // per-tile column sums of v and v * (bz - bmean): the statistics of the
  const void* bzd; long ldbzd; const float* bmean_d;  // BatchNorm backward that consumes v; optionally also v * (bzd - bmean_d)
  float* bpart;                                       // [M tiles][3][N]
  int r_hi, r_wi;                                     // > 0: the residual holds rows only for the even (y, x) pixels of an r_hi x r_wi frame
};

template <typename T> struct Frag;
template <> struct Frag<bf16_t> { typedef bf16x8 type; };
template <> struct Frag<float> { typedef f32x4 type; };

template <typename T>
__device__ __forceinline__ f32x4 mma(const typename Frag<T>::type& a, const typename Frag<T>::type& b, f32x4 c);
template <>
__device__ __forceinline__ f32x4 mma<bf16_t>(const bf16x8& a, const bf16x8& b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
}
template <>
__device__ __forceinline__ f32x4 mma<float>(const f32x4& a, const f32x4& b, f32x4 c) {
#pragma unroll
  for (int j = 0; j < 4; ++j) c = __builtin_amdgcn_mfma_f32_16x16x4f32(a[j], b[j], c, 0, 0, 0);
  return c;
}

__device__ __forceinline__ int swz(int row, int chunk) { return row * 128 + ((chunk ^ (row & 7)) << 4); }

// SE: the per-frame operand re-scale (conv3 after the SE gate) is compiled in only where it is used.  SE = 2 keeps the
// gates of the (few) frames a 128-row tile touches in LDS (dynamic, frames x K floats) and reads them when a slab is
// stored; SE = 1 is the general form that prefetches them in registers next to the slab (32 VGPRs).  __launch_bounds__(256, 2) makes the compiler keep two workgroups per CU resident; a variant with a second
// register stage (two K slabs in flight) was measured: it needs > 256 VGPRs, spills, and loses (43 vs 32 us at
// M=39200, K=N=368).  (A form that COMPUTED the gate table in its prologue -- no se_gate launch -- was measured slower and is
// parked: experiments/r4_parked/gemm_with_se_in_conv3.hip.)
template <typename T, int BN, int SE, bool BWD = false>
__global__ __launch_bounds__(256, 2) void gemm_kernel(const GemmP p) {
  constexpr int EPC = Chunk<T>::N;        // elements per 16-B chunk
  constexpr int KT = 8 * EPC;             // elements of K per slab
  constexpr int NT = BN / 32;             // 16-wide column tiles per wave
  constexpr int BROWS = BN / 32;          // B rows staged per thread
  constexpr int A_BYTES = 128 * 128, B_BYTES = BN * 128;
  constexpr int STAGE = A_BYTES + B_BYTES;
  constexpr int CS_LD = BN + 4;
  constexpr int LDS_BYTES = (STAGE > 64 * CS_LD * 4) ? STAGE : 64 * CS_LD * 4;
  __shared__ __attribute__((aligned(16))) unsigned char lds[LDS_BYTES];
  extern __shared__ __attribute__((aligned(16))) float gtab[];     // SE == 2: [frames of this tile][K]
  typedef typename Frag<T>::type frag_t;

  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int wr = wid >> 1, wc = wid & 1;
  const int nb = (p.N + BN - 1) / BN;
  // XCD-aware tile order (guide T1, bijective form): workgroups are dealt round-robin over the 8 XCDs, so
  // logical tiles are renumbered such that consecutive ones (the N tiles of one M tile, which re-read the same
  // A rows) share an XCD and therefore its L2, instead of fetching A from HBM / Infinity Cache once per XCD.
  const long lid = xcd_logical_id(blockIdx.x, gridDim.x);
  int tile_n, tile_m_;
  td_split(lid, nb, tile_m_, tile_n);
  const long tile_m = tile_m_;
  const long m0 = tile_m * 128;
  const int n0 = tile_n * BN;

  // ---- per-thread staging assignment: chunk = tid&7, rows (tid>>3) + 32*i
  const int ch = tid & 7;
  const int r0 = tid >> 3;
  const T* arow[4];
  const T* a0row[4];
  const float* srow[4];
  bool aok[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    long m = m0 + r0 + 32 * i;
    aok[i] = m < p.M;
    long mm = aok[i] ? m : 0;
    long src = mm;
    const unsigned mu = (unsigned)mm;            // rows are < 2^31 (M is an int): unsigned 32-bit divisions, not the 64-bit
    if (p.g_stride > 1) {                        // sequences (~120 vector instructions each: common.h td_split)
      const unsigned per = (unsigned)(p.g_ho * p.g_wo);
      const unsigned f = mu / per, rem = mu - f * per;
      const unsigned yo = rem / (unsigned)p.g_wo, xo = rem - yo * (unsigned)p.g_wo;
      src = ((long)f * p.g_hi + (long)yo * p.g_stride) * p.g_wi + (long)xo * p.g_stride;
    }
    arow[i] = reinterpret_cast<const T*>(p.A) + src * p.lda;
    a0row[i] = p.A0 ? reinterpret_cast<const T*>(p.A0) + src * p.lda0 : nullptr;
    srow[i] = p.a_scale ? p.a_scale + (long)(mu / (unsigned)p.a_scale_rows) * (long)p.K : nullptr;
  }
  const T* brow[BROWS];
  bool bok[BROWS];
#pragma unroll
  for (int i = 0; i < BROWS; ++i) {
    int n = n0 + r0 + 32 * i;
    bok[i] = n < p.N;
    brow[i] = reinterpret_cast<const T*>(p.W) + (long)(bok[i] ? n : 0) * p.ldw;
  }

  u32x4 areg[4], breg[BROWS];
  f32x4 greg[SE == 1 ? 4 : 1][EPC / 4];  // SE gate values of the prefetched slab (applied at LDS-store time)
  int gfi[4] = {0, 0, 0, 0};             // SE == 2: row -> frame slot in gtab
  const u32x4 zero4 = {0u, 0u, 0u, 0u};

  // branch-free prefetch: out-of-range chunks read a valid dummy address and are zeroed by a select
  auto gload = [&](int kt) {
    const int k = kt * KT + ch * EPC;
    const bool kok = k < p.K;
    const int kc = kok ? k : 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const T* src = (a0row[i] && kc < p.k0) ? a0row[i] + kc : arow[i] + kc;
      const u32x4 v = *reinterpret_cast<const u32x4*>(src);
      areg[i] = (kok && aok[i]) ? v : zero4;
    }
#pragma unroll
    for (int i = 0; i < BROWS; ++i) {
      const u32x4 v = *reinterpret_cast<const u32x4*>(brow[i] + kc);
      breg[i] = (kok && bok[i]) ? v : zero4;
    }
    if constexpr (SE == 1) {
      if (p.a_scale) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int h = 0; h < EPC / 4; ++h) greg[i][h] = *reinterpret_cast<const f32x4*>(srow[i] + kc + 4 * h);
      }
    }
  };
  auto lstore = [&](int buf, int kt) {
    unsigned char* base = lds + buf * STAGE;
    if (SE == 1 && p.a_scale) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        float v[EPC];
        Chunk<T>::load(reinterpret_cast<const T*>(&areg[i]), v);
#pragma unroll
        for (int e = 0; e < EPC; ++e) v[e] *= greg[i][e >> 2][e & 3];
        Chunk<T>::store(reinterpret_cast<T*>(&areg[i]), v);
      }
    }
    if constexpr (SE >= 2) {
      const int kc = min(kt * KT + ch * EPC, p.K - EPC);        // out-of-range slabs are zero already
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        float v[EPC];
        Chunk<T>::load(reinterpret_cast<const T*>(&areg[i]), v);
#pragma unroll
        for (int h = 0; h < EPC / 4; ++h) {
          const f32x4 g = *reinterpret_cast<const f32x4*>(gtab + gfi[i] * p.K + kc + 4 * h);
#pragma unroll
          for (int e = 0; e < 4; ++e) v[4 * h + e] *= g[e];
        }
        Chunk<T>::store(reinterpret_cast<T*>(&areg[i]), v);
      }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) *reinterpret_cast<u32x4*>(base + swz(r0 + 32 * i, ch)) = areg[i];
#pragma unroll
    for (int i = 0; i < BROWS; ++i)
      *reinterpret_cast<u32x4*>(base + A_BYTES + swz(r0 + 32 * i, ch)) = breg[i];
  };

  f32x4 acc[4][NT];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const int nkt = (p.K + KT - 1) / KT;
  const int fr = lane & 15, fq = lane >> 4;
  auto compute = [&]() {
    const unsigned char* abase = lds;
    const unsigned char* bbase = abase + A_BYTES;
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      frag_t af[4], bfr[NT];
#pragma unroll
      for (int mt = 0; mt < 4; ++mt)
        af[mt] = *reinterpret_cast<const frag_t*>(abase + swz(wr * 64 + mt * 16 + fr, 4 * s + fq));
#pragma unroll
      for (int nt = 0; nt < NT; ++nt)
        bfr[nt] = *reinterpret_cast<const frag_t*>(bbase + swz(wc * (BN / 2) + nt * 16 + fr, 4 * s + fq));
#pragma unroll
      for (int mt = 0; mt < 4; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = mma<T>(af[mt], bfr[nt], acc[mt][nt]);
    }
  };
  gload(0);
  if constexpr (SE == 2) {
    // gates of the frames this tile touches -> LDS (rows of one tile span at most 128 / a_scale_rows + 2 frames)
    const long f_first = (long)((unsigned)m0 / (unsigned)p.a_scale_rows);
    const long f_last = (long)((unsigned)(p.M - 1) / (unsigned)p.a_scale_rows);
    const int nf = 128 / p.a_scale_rows + 2;
    const int ng = nf * p.K;
    for (int i0 = tid * 4; i0 < ng; i0 += 4 * 1024) {           // four pieces per thread in flight (not a round trip per piece)
      f32x4 gv[4];
#pragma unroll
      for (int b_ = 0; b_ < 4; ++b_) {
        const int i = min(i0 + b_ * 1024, ng - 4);
        const int f = i / p.K, k = i - f * p.K;
        gv[b_] = *reinterpret_cast<const f32x4*>(p.a_scale + min(f_first + f, f_last) * (long)p.K + k);
      }
      TD_ISSUE_FENCE();
#pragma unroll
      for (int b_ = 0; b_ < 4; ++b_)
        if (i0 + b_ * 1024 < ng) *reinterpret_cast<f32x4*>(gtab + i0 + b_ * 1024) = gv[b_];
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const long m = min(m0 + r0 + 32 * i, (long)p.M - 1);
      gfi[i] = (int)((long)((unsigned)m / (unsigned)p.a_scale_rows) - f_first);
    }
    __syncthreads();
  }
  lstore(0, 0);
  __syncthreads();
  if (nkt > 1) gload(1);                       // tile k+1 sits in registers while tile k is consumed from LDS
  for (int kt = 0; kt < nkt; ++kt) {
    compute();
    __syncthreads();                           // every wave is done reading tile kt
    if (kt + 1 < nkt) {
      lstore(0, kt + 1);
      if (kt + 2 < nkt) gload(kt + 2);
    }
    __syncthreads();
  }

  // ---- epilogue: two passes of 64 rows through an fp32 LDS tile
  float* cs = reinterpret_cast<float*>(lds);
  float sc[NT], sh[NT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) {
    int n = n0 + wc * (BN / 2) + nt * 16 + fr;
    bool ok = n < p.N;
    sc[nt] = (p.scale && ok) ? p.scale[n] : 1.0f;
    sh[nt] = (p.shift && ok) ? p.shift[n] : 0.0f;
  }
  constexpr int CPR = BN / EPC;   // chunks per tile row
  // training: column sum / sum of squares of what is stored (rounded to T), per M tile: the BatchNorm statistics of a raw
  // conv output come out of its own epilogue instead of a second pass over the map.  A thread always serves the same
  // column chunk (CPR divides 256), so it accumulates over its rows in registers; lanes are folded through LDS at the end.
  float cs1[EPC], cs2[EPC];
#pragma unroll
  for (int e = 0; e < EPC; ++e) cs1[e] = cs2[e] = 0.f;
  // residual chunks of both passes are requested NOW (bf16: 2 x BN / 32 x 16 bytes per thread, in the registers the operand
  // staging no longer needs): they travel while the accumulators go through LDS, instead of one exposed round trip per pass
  // (a 320 x 320 layer over 156800 rows: 112 -> ~85 us)
  constexpr bool RPRE = sizeof(T) == 2;
  constexpr int NPT = (64 * CPR + 255) / 256;
  u32x4 rres[RPRE ? 2 : 1][RPRE ? NPT : 1];
  // BWD with a stride-2 residual (the shortcut conv's input gradient lives on the even pixels only): row m = (f, y, x) of an
  // r_hi x r_wi frame reads residual row (f, y / 2, x / 2) when y and x are even, nothing otherwise
  [[maybe_unused]] unsigned rskip = 0u;                         // bit (half * NPT + it): no residual for that chunk
  auto rrow = [&](long m, bool& has) -> long {
    if constexpr (BWD) {
      if (p.r_hi > 0) {
        const long per = (long)p.r_hi * p.r_wi;
        const long f = m / per;
        const int rem = (int)(m - f * per);
        const int yy = rem / p.r_wi, xx = rem - yy * p.r_wi;
        has = !((yy | xx) & 1);
        const int ho = (p.r_hi + 1) >> 1, wo = (p.r_wi + 1) >> 1;
        return has ? (f * ho + (yy >> 1)) * wo + (xx >> 1) : 0;
      }
    }
    has = true;
    return m;
  };
  if constexpr (RPRE) {
    if (p.R) {
#pragma unroll
      for (int half = 0; half < 2; ++half)
#pragma unroll
        for (int it = 0; it < NPT; ++it) {
          const int idx = min(tid + it * 256, 64 * CPR - 1);
          const int row = idx / CPR, cj = idx - row * CPR;
          const long m = min(m0 + half * 64 + row, (long)p.M - 1);
          const int n = min(n0 + cj * EPC, p.N - EPC);
          bool has;
          const long rm = rrow(m, has);
          if (!has) rskip |= 1u << (half * NPT + it);
          rres[half][it] = *reinterpret_cast<const u32x4*>(reinterpret_cast<const T*>(p.R) + rm * p.ldr + n);
        }
    }
  }
  // BWD statistics: a thread serves one column chunk in every iteration (CPR divides 256): its means live in registers
  [[maybe_unused]] float bm[BWD ? EPC : 1], bmd[BWD ? EPC : 1], cs3[BWD ? EPC : 1];
  if constexpr (BWD) {
    const int nb0 = min(n0 + (tid % CPR) * EPC, p.N - EPC);
#pragma unroll
    for (int e = 0; e < EPC; ++e) {
      bm[e] = p.bpart ? p.bmean[nb0 + e] : 0.f;
      bmd[e] = (p.bpart && p.bzd) ? p.bmean_d[nb0 + e] : 0.f;
      cs3[e] = 0.f;
    }
  }
  for (int half = 0; half < 2; ++half) {
    if (wr == half) {
#pragma unroll
      for (int mt = 0; mt < 4; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            int row = mt * 16 + fq * 4 + r;
            int col = wc * (BN / 2) + nt * 16 + fr;
            cs[row * CS_LD + col] = acc[mt][nt][r] * sc[nt] + sh[nt];
          }
    }
    __syncthreads();
#pragma unroll
    for (int it = 0; it < NPT; ++it) {
      const int idx = tid + it * 256;
      if (idx >= 64 * CPR) break;
      int row = idx / CPR, cj = idx - row * CPR;
      long m = m0 + half * 64 + row;
      int n = n0 + cj * EPC;
      if (m < p.M && n < p.N) {
        float v[EPC];
#pragma unroll
        for (int e = 0; e < EPC; ++e) v[e] = cs[row * CS_LD + cj * EPC + e];
        if (p.c2_pre && n < p.n2) {               // input gradient of a gate-shift conv1: these columns go to the module only
          Chunk<T>::store(reinterpret_cast<T*>(p.C2) + m * p.ldc2 + n, v);
#pragma unroll
          for (int e = 0; e < EPC; ++e) v[e] = 0.f;
        }
        [[maybe_unused]] u32x4 mk4, bz4, bzd4;
        if constexpr (BWD) {                              // this chunk's mask / statistics operands: requested before the residual math
          if (p.mask) mk4 = *reinterpret_cast<const u32x4*>(reinterpret_cast<const T*>(p.mask) + m * p.ldmask + n);
          if (p.bpart) {
            bz4 = *reinterpret_cast<const u32x4*>(reinterpret_cast<const T*>(p.bz) + m * p.ldbz + n);
            if (p.bzd) bzd4 = *reinterpret_cast<const u32x4*>(reinterpret_cast<const T*>(p.bzd) + m * p.ldbzd + n);
          }
        }
        if (p.R) {
          float rv[EPC];
          bool has = true;
          if constexpr (RPRE) {
            Chunk<T>::load(reinterpret_cast<const T*>(&rres[half][it]), rv);
            if constexpr (BWD) has = !((rskip >> (half * NPT + it)) & 1u);
          } else {
            const long rm = rrow(m, has);
            Chunk<T>::load(reinterpret_cast<const T*>(p.R) + rm * p.ldr + n, rv);
          }
#pragma unroll
          for (int e = 0; e < EPC; ++e) v[e] += has ? rv[e] : 0.f;
        }
        if constexpr (BWD) {
          if (p.mask) {
            float mv[EPC];
            Chunk<T>::load(reinterpret_cast<const T*>(&mk4), mv);
#pragma unroll
            for (int e = 0; e < EPC; ++e) v[e] = mv[e] > 0.f ? v[e] : 0.f;
          }
        }
        if (p.act == TDEED_ACT_RELU) {
#pragma unroll
          for (int e = 0; e < EPC; ++e) v[e] = fmaxf(v[e], 0.f);
        } else if (p.act == TDEED_ACT_GELU) {
#pragma unroll
          for (int e = 0; e < EPC; ++e) v[e] = gelu_erf(v[e]);
        }
        Chunk<T>::store(reinterpret_cast<T*>(p.C) + m * p.ldc + n, v);
        if (p.C2 && !p.c2_pre && n < p.n2) Chunk<T>::store(reinterpret_cast<T*>(p.C2) + m * p.ldc2 + n, v);
        if (p.colpart) {
#pragma unroll
          for (int e = 0; e < EPC; ++e) {
            const float r = round_to<T>(v[e]);
            cs1[e] += r;
            cs2[e] = fmaf(r, r, cs2[e]);
          }
        }
        if constexpr (BWD) {
          if (p.bpart) {
            float zv[EPC];
            Chunk<T>::load(reinterpret_cast<const T*>(&bz4), zv);
#pragma unroll
            for (int e = 0; e < EPC; ++e) {
              const float r = round_to<T>(v[e]);
              cs1[e] += r;
              cs2[e] = fmaf(r, zv[e] - bm[e], cs2[e]);
            }
            if (p.bzd) {
              Chunk<T>::load(reinterpret_cast<const T*>(&bzd4), zv);
#pragma unroll
              for (int e = 0; e < EPC; ++e) cs3[e] = fmaf(round_to<T>(v[e]), zv[e] - bmd[e], cs3[e]);
            }
          }
        }
      }
    }
    __syncthreads();
  }
  if constexpr (BWD) {
    if (p.bpart) {                                 // one row of sums at a time through [RLN][BN] floats (fits every tile shape)
      constexpr int RLN = 256 / CPR;
      float* red = reinterpret_cast<float*>(lds);
      const int cj = tid % CPR, rl = tid / CPR;
      const int nrow = p.bzd ? 3 : 2;
      for (int which = 0; which < nrow; ++which) {
#pragma unroll
        for (int e = 0; e < EPC; ++e) red[rl * BN + cj * EPC + e] = which == 0 ? cs1[e] : (which == 1 ? cs2[e] : cs3[e]);
        __syncthreads();
        for (int col = tid; col < BN; col += 256) {
          if (n0 + col < p.N) {
            float a = 0.f;
            for (int i = 0; i < RLN; ++i) a += red[i * BN + col];
            p.bpart[((long)tile_m * 3 + which) * p.N + n0 + col] = a;
          }
        }
        __syncthreads();
      }
    }
  }
  if (p.colpart) {
    constexpr int RLN = 256 / CPR;          // row lanes per column chunk
    float* red = reinterpret_cast<float*>(lds);                 // [RLN][2][BN]  (<= 64 * CS_LD floats)
    const int cj = tid % CPR, rl = tid / CPR;
#pragma unroll
    for (int e = 0; e < EPC; ++e) {
      red[(rl * 2 + 0) * BN + cj * EPC + e] = cs1[e];
      red[(rl * 2 + 1) * BN + cj * EPC + e] = cs2[e];
    }
    __syncthreads();
    for (int j = tid; j < 2 * BN; j += 256) {
      const int which = j / BN, col = j - which * BN;
      if (n0 + col < p.N) {
        float a = 0.f;
        for (int i = 0; i < RLN; ++i) a += red[(i * 2 + which) * BN + col];
        p.colpart[((long)tile_m * 2 + which) * p.N + n0 + col] = a;
      }
    }
  }
}

template <typename T>
static int launch_gemm(const GemmP& p, hipStream_t st, bool bwd = false) {
  const long mb = (p.M + 127) / 128;
  // column tile: least padded MFMA work, narrower tiles charged for their extra A re-reads /
  // LDS traffic; then shrink while the grid would leave most of the 256 CUs idle.
  int bn = 128;
  {
    const int cand[3] = {128, 64, 32};
    const float pen[3] = {1.0f, 1.1f, 1.3f};
    float best = 1e30f;
    for (int i = 0; i < 3; ++i) {
      float c = (float)(((p.N + cand[i] - 1) / cand[i]) * cand[i]) * pen[i];
      if (c < best) { best = c; bn = cand[i]; }
    }
    while (bn > 32 && mb * ((p.N + bn - 1) / bn) < 384) bn >>= 1;
    static int force = -1;
    if (force < 0) { const char* e = getenv("TDEED_GEMM_BN"); force = e ? atoi(e) : 0; }
    if (force == 32 || force == 64 || force == 128) bn = force;
  }
  const long nb = (p.N + bn - 1) / bn;
  const long grid = mb * nb;
  if (grid > 0x7fffffffL) { tdeed_set_error("gemm: grid too large"); return TDEED_ERR_ARG; }
  // SE gate table in LDS when the frames of one 128-row tile fit 24 KB (K % 4 == 0 holds: K is a multiple of 8)
  const size_t gt_bytes = p.a_scale ? (size_t)(128 / p.a_scale_rows + 2) * p.K * sizeof(float) : 0;
  int se = !p.a_scale ? 0 : (gt_bytes <= 24 * 1024 ? 2 : 1);
#define TD_GEMM(BNv)                                                                                                   \
  do {                                                                                                                 \
    if (bwd) {                                                                                                         \
      hipLaunchKernelGGL((gemm_kernel<T, BNv, 0, true>), dim3((unsigned)grid), dim3(256), 0, st, p);                   \
      break;                                                                                                           \
    }                                                                                                                  \
    if (se == 2) hipLaunchKernelGGL((gemm_kernel<T, BNv, 2>), dim3((unsigned)grid), dim3(256), gt_bytes, st, p);      \
    else if (se == 1) hipLaunchKernelGGL((gemm_kernel<T, BNv, 1>), dim3((unsigned)grid), dim3(256), 0, st, p);         \
    else hipLaunchKernelGGL((gemm_kernel<T, BNv, 0>), dim3((unsigned)grid), dim3(256), 0, st, p);                      \
  } while (0)
  if (bn == 32) TD_GEMM(32);
  else if (bn == 64) TD_GEMM(64);
  else TD_GEMM(128);
#undef TD_GEMM
  TD_LAUNCH_CHECK("gemm");
  return TDEED_OK;
}

extern "C" int tdeed_gemm_fwd(const void* A, long lda, const void* A0, long lda0, int k0,
                              const float* a_scale, int a_scale_rows, int M, int K, int N,
                              const void* W, long ldw, const float* scale, const float* shift,
                              const void* R, long ldr, int act, void* C, long ldc,
                              int gather_stride, int gather_hi, int gather_wi, int gather_ho,
                              int gather_wo, float* colpart, void* C2, long ldc2, int n2, int c2_pre, int dtype, void* stream) {
  TD_CHECK(A && W && C, "gemm: null pointer");
  TD_CHECK(!C2 || (n2 > 0 && n2 % 8 == 0 && n2 <= N && ldc2 % 8 == 0 && ldc2 >= n2), "gemm: bad second output n2=%d ldc2=%ld",
           n2, ldc2);
  TD_CHECK(M > 0 && K > 0 && N > 0, "gemm: bad sizes M=%d K=%d N=%d", M, K, N);
  TD_CHECK(dtype == TDEED_F32 || dtype == TDEED_BF16, "gemm: bad dtype %d", dtype);
  const int q = 8;
  TD_CHECK(K % q == 0 && N % q == 0 && lda % q == 0 && ldw % q == 0 && ldc % q == 0,
           "gemm: K=%d N=%d lda=%ld ldw=%ld ldc=%ld must be multiples of 8", K, N, lda, ldw, ldc);
  TD_CHECK(!A0 || (k0 % q == 0 && lda0 % q == 0 && k0 <= K), "gemm: bad splice k0=%d lda0=%ld", k0, lda0);
  TD_CHECK(!R || ldr % q == 0, "gemm: ldr=%ld must be a multiple of 8", ldr);
  TD_CHECK(!a_scale || a_scale_rows > 0, "gemm: a_scale_rows must be > 0");
  TD_CHECK(act >= 0 && act <= 2, "gemm: bad act %d", act);
  if (gather_stride > 1)
    TD_CHECK(gather_hi > 0 && gather_wi > 0 && gather_ho > 0 && gather_wo > 0 && M % (gather_ho * gather_wo) == 0,
             "gemm: bad gather geometry");
  GemmP p{};
  p.A = A; p.lda = lda; p.A0 = A0; p.lda0 = lda0; p.k0 = A0 ? k0 : 0;
  p.a_scale = a_scale; p.a_scale_rows = a_scale_rows > 0 ? a_scale_rows : 1;
  p.M = M; p.K = K; p.N = N; p.W = W; p.ldw = ldw; p.scale = scale; p.shift = shift;
  p.R = R; p.ldr = ldr; p.act = act; p.C = C; p.ldc = ldc;
  p.g_stride = gather_stride; p.g_hi = gather_hi; p.g_wi = gather_wi; p.g_ho = gather_ho; p.g_wo = gather_wo;
  p.colpart = colpart;
  p.C2 = C2; p.ldc2 = ldc2; p.n2 = C2 ? n2 : 0; p.c2_pre = (C2 && c2_pre) ? 1 : 0;
  hipStream_t st = (hipStream_t)stream;
  return dtype == TDEED_F32 ? launch_gemm<float>(p, st) : launch_gemm<bf16_t>(p, st);
}

// Input-gradient contraction of a training bottleneck's conv1 (dx = dz1 @ W1, W given as [N = Cin][K = Cout]) whose output IS the
// gradient at the previous block's output ReLU (timm Bottleneck.forward: x = act3(bn3(conv3) + shortcut), under autograd;
// /root/reference/model/model.py:265-324): instead of leaving that ReLU's backward and the statistics pass of the BatchNorm
// backward behind it to passes of their own over the map, the epilogue
//   * adds the residual R (the block's own shortcut gradient; with r_hi > 0 R holds rows for the even pixels of an
//     r_hi x r_wi frame only: the stride-2 shortcut conv's input gradient, no scatter-add pass),
//   * sends columns [0, n2) to C2 before the residual (gate-shift blocks, as tdeed_gemm_fwd's c2_pre),
//   * masks with mask[m][n] > 0 (the previous block's output = this block's input),
//   * leaves per-tile column sums of the stored values v: sum v, sum v * (bz - bmean) and (bzd given: the previous block's
//     shortcut BatchNorm) sum v * (bzd - bmean_d) in bpart [ceil(M / 128)][3][N].
extern "C" int tdeed_gemm_dgrad(const void* A, long lda, int M, int K, int N, const void* W, long ldw, const void* R, long ldr,
                                int r_hi, int r_wi, void* C, long ldc, void* C2, long ldc2, int n2, const void* mask,
                                long ldmask, const void* bz, long ldbz, const float* bmean, const void* bzd, long ldbzd,
                                const float* bmean_d, float* bpart, int dtype, void* stream) {
  TD_CHECK(A && W && C, "gemm_dgrad: null pointer");
  TD_CHECK(M > 0 && K > 0 && N > 0, "gemm_dgrad: bad sizes M=%d K=%d N=%d", M, K, N);
  TD_CHECK(dtype == TDEED_F32 || dtype == TDEED_BF16, "gemm_dgrad: bad dtype %d", dtype);
  TD_CHECK(K % 8 == 0 && N % 8 == 0 && lda % 8 == 0 && ldw % 8 == 0 && ldc % 8 == 0, "gemm_dgrad: sizes must be multiples of 8");
  TD_CHECK(!C2 || (n2 > 0 && n2 % 8 == 0 && n2 <= N && ldc2 % 8 == 0 && ldc2 >= n2), "gemm_dgrad: bad second output");
  TD_CHECK(!R || ldr % 8 == 0, "gemm_dgrad: bad residual stride");
  TD_CHECK(r_hi == 0 || (R && r_hi > 0 && r_wi > 0 && M % (r_hi * r_wi) == 0), "gemm_dgrad: bad stride-2 residual geometry");
  TD_CHECK(!mask || ldmask % 8 == 0, "gemm_dgrad: bad mask stride");
  TD_CHECK(!bpart || (bz && bmean && ldbz % 8 == 0 && (!bzd || (bmean_d && ldbzd % 8 == 0))), "gemm_dgrad: bad statistics operands");
  GemmP p{};
  p.A = A; p.lda = lda; p.a_scale_rows = 1;
  p.M = M; p.K = K; p.N = N; p.W = W; p.ldw = ldw;
  p.R = R; p.ldr = ldr; p.act = TDEED_ACT_NONE; p.C = C; p.ldc = ldc;
  p.g_stride = 1;
  p.C2 = C2; p.ldc2 = ldc2; p.n2 = C2 ? n2 : 0; p.c2_pre = C2 ? 1 : 0;
  p.mask = mask; p.ldmask = ldmask;
  p.bz = bz; p.ldbz = ldbz; p.bmean = bmean; p.bzd = bpart ? bzd : nullptr; p.ldbzd = ldbzd; p.bmean_d = bmean_d;
  p.bpart = bpart;
  p.r_hi = r_hi; p.r_wi = r_wi;
  hipStream_t st = (hipStream_t)stream;
  return dtype == TDEED_F32 ? launch_gemm<float>(p, st, true) : launch_gemm<bf16_t>(p, st, true);
}

// =============================================================================================
// Weight-stationary streaming contraction for the narrow layers (K, N <= ~176: every 1x1 conv of
// stages s1-s3 of RegNetY-200MF).  These are pure HBM streams (2-8 flop/byte .. 150 flop/byte), so
// the kernel is built around bytes in flight, not MFMA rate:
//   * the whole weight matrix sits in LDS, pre-packed on the host in MFMA fragment order
//     ([n-tile][k-step][lane][16 B]) so a fragment read is one conflict-free 1-KiB ds_read_b128;
//   * activations go global -> registers directly in fragment shape (lane = pixel l&15, k-chunk
//     l>>4); no LDS staging, no block barriers in the loop, every wave independent;
//   * persistent blocks; 3 waves/SIMD x 10 outstanding 16-B loads per lane hide the HBM latency;
//   * weights are the MFMA A operand, so D[n][pixel] leaves each lane with 4 channels of ONE pixel;
//     weight rows are permuted on the host so that two n-tiles give 8 consecutive channels:
//     16-byte residual loads / output stores straight from the accumulators, no LDS transpose.
// The SE gate, the gate-shift splice and the stride-2 row gather are applied while loading.
struct GemmWsP {
  const void* A; long lda;
  const void* A0; long lda0; int k0;
  const float* a_scale; int a_scale_rows;
  int M, K, N;
  const void* Wf;                 // [NT][KS][64] x 16 B
  const float* scale; const float* shift;
  const void* R; long ldr;
  int act;
  void* C; long ldc;
  int g_stride, g_hi, g_wi, g_ho, g_wo;
  int NT;                         // n-tiles of the whole matrix
  int NTS;                        // n-tiles per block slice (blockIdx.y selects the slice; == NT when W fits LDS)
  void* C2; long ldc2; int n2;    // optional second, compact copy of columns [0, n2) of C
  float* colpart;                 // gemm_rs STATS: [gridDim.x][2][N] per-workgroup column sums / sums of squares of the stored C
};

// WLDS = false: the weights do not fit LDS (K = N = 368): fragments are read straight from global memory
// (1-KiB coalesced wave loads of the pre-packed array; L1/L2 resident, all waves of a CU walk it in step).
// NW: waves per workgroup (4; 8 for the sliced form, where the LDS budget leaves one workgroup per CU and two waves per
// SIMD are needed for one wave's loads to travel while the other one computes).
template <typename T, int KS, bool WLDS, int NW = 4>
__global__ __launch_bounds__(NW * 64, NW == 4 ? 2 : 1) void gemm_ws_kernel(const GemmWsP p) {
  constexpr int NTHR = NW * 64, CR = NW * 32;        // threads, rows per chunk
  constexpr int EPC = Chunk<T>::N;
  typedef typename Frag<T>::type frag_t;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  // a block owns the n-tile slice [nt0, nt0 + nts) of W (all of it when it fits LDS)
  const int nt0 = WLDS ? blockIdx.y * p.NTS : 0;
  const int nts = WLDS ? min(p.NTS, p.NT - nt0) : p.NT;
  const size_t wbytes = WLDS ? (size_t)p.NTS * KS * 64 * 16 : 0;
  const frag_t* wl = WLDS ? reinterpret_cast<const frag_t*>(smem) : reinterpret_cast<const frag_t*>(p.Wf);
  float* ssc = reinterpret_cast<float*>(smem + wbytes);               // [nts*16] scale (logical order)
  float* ssh = ssc + (WLDS ? p.NTS : p.NT) * 16;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  {
    if (WLDS) {
      frag_t* wdst = reinterpret_cast<frag_t*>(smem);
      const frag_t* src = reinterpret_cast<const frag_t*>(p.Wf) + (size_t)nt0 * KS * 64;
      TD_DEV_ASSERT(nts <= p.NTS && nt0 + nts <= p.NT);
      // 8 fragments per thread in flight (round 6: `wdst[i] = src[i]` per trip was one `global_load; s_waitcnt vmcnt(0)` per
      // 16 bytes -- a 152 x 152 weight cost every workgroup 13 dependent L2 round trips before its first row)
      const int nfr = nts * KS * 64;
      constexpr int WB = 8;
      for (int i0 = tid; i0 < nfr; i0 += NTHR * WB) {
        frag_t v[WB];
#pragma unroll
        for (int b_ = 0; b_ < WB; ++b_) v[b_] = src[min(i0 + b_ * NTHR, nfr - 1)];
        TD_ISSUE_FENCE();
#pragma unroll
        for (int b_ = 0; b_ < WB; ++b_)
          if (i0 + b_ * NTHR < nfr) wdst[i0 + b_ * NTHR] = v[b_];
      }
    }
    for (int i = tid; i < nts * 16; i += NTHR) {
      const int n = nt0 * 16 + i;
      ssc[i] = (p.scale && n < p.N) ? p.scale[n] : 1.0f;
      ssh[i] = (p.shift && n < p.N) ? p.shift[n] : 0.0f;
    }
  }
  __syncthreads();
  const int px = lane & 15, q = lane >> 4;
  const long nchunks = ((long)p.M + CR - 1) / CR;
  for (long chunk = blockIdx.x; chunk < nchunks; chunk += gridDim.x) {
    const long mbase = chunk * CR + wv * 32;
    frag_t xf[2][KS];
    bool mok[2];
    long mrow[2];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
      const long m = mbase + mt * 16 + px;
      mok[mt] = m < p.M;
      const long mm = mok[mt] ? m : 0;
      mrow[mt] = mm;
      long src = mm;
      const unsigned mu = (unsigned)mm;          // (rows < 2^31: 32-bit divisions, common.h td_split)
      if (p.g_stride > 1) {
        const unsigned per = (unsigned)(p.g_ho * p.g_wo);
        const unsigned f = mu / per, rem = mu - f * per;
        const unsigned yo = rem / (unsigned)p.g_wo, xo = rem - yo * (unsigned)p.g_wo;
        src = ((long)f * p.g_hi + (long)yo * p.g_stride) * p.g_wi + (long)xo * p.g_stride;
      }
      const T* arow = reinterpret_cast<const T*>(p.A) + src * p.lda;
      const T* a0row = p.A0 ? reinterpret_cast<const T*>(p.A0) + src * p.lda0 : nullptr;
      const float* srow = p.a_scale ? p.a_scale + (long)(mu / (unsigned)p.a_scale_rows) * (long)p.K : nullptr;
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        const int k = (ks * 4 + q) * EPC;
        u32x4 v = {0u, 0u, 0u, 0u};
        if (mok[mt] && k < p.K) {
          const T* s = (a0row && k < p.k0) ? a0row + k : arow + k;
          v = *reinterpret_cast<const u32x4*>(s);
          if (srow) {
            float f[EPC];
            Chunk<T>::load(reinterpret_cast<const T*>(&v), f);
#pragma unroll
            for (int e4 = 0; e4 < EPC; e4 += 4) {
              const f32x4 g4 = *reinterpret_cast<const f32x4*>(srow + k + e4);
#pragma unroll
              for (int e = 0; e < 4; ++e) f[e4 + e] *= g4[e];
            }
            Chunk<T>::store(reinterpret_cast<T*>(&v), f);
          }
        }
        xf[mt][ks] = *reinterpret_cast<frag_t*>(&v);
      }
    }
    for (int tp = 0; tp < nts; tp += 2) {            // 32 output channels per pass (tp: tile within the slice)
      const int chp = (nt0 + tp) * 16 + q * 8;
      u32x4 rpre[2][8 / EPC];
      if (p.R && chp < p.N) {                          // residual chunks: issued now, consumed after the MFMAs
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
          for (int h = 0; h < 8 / EPC; ++h)
            rpre[mt][h] = *reinterpret_cast<const u32x4*>(reinterpret_cast<const T*>(p.R) + mrow[mt] * p.ldr + chp + h * EPC);
      }
      f32x4 acc[2][2];
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) acc[t][mt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        const frag_t w0 = wl[((tp + 0) * KS + ks) * 64 + lane];
        const frag_t w1 = wl[((tp + 1) * KS + ks) * 64 + lane];
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
          acc[0][mt] = mma<T>(w0, xf[mt][ks], acc[0][mt]);
          acc[1][mt] = mma<T>(w1, xf[mt][ks], acc[1][mt]);
        }
      }
      const int ch = (nt0 + tp) * 16 + q * 8;         // first of this lane's 8 logical channels
      if (ch < p.N) {
        float sc[8], sh[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) { sc[e] = ssc[tp * 16 + q * 8 + e]; sh[e] = ssh[tp * 16 + q * 8 + e]; }
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
          if (!mok[mt]) continue;
          float v[8];
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            v[r] = acc[0][mt][r] * sc[r] + sh[r];
            v[4 + r] = acc[1][mt][r] * sc[4 + r] + sh[4 + r];
          }
          const long m = mrow[mt];
          if (p.R) {
            float rv[EPC];
#pragma unroll
            for (int h = 0; h < 8 / EPC; ++h) {
              Chunk<T>::load(reinterpret_cast<const T*>(&rpre[mt][h]), rv);
#pragma unroll
              for (int e = 0; e < EPC; ++e) v[h * EPC + e] += rv[e];
            }
          }
          if (p.act == TDEED_ACT_RELU) {
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = fmaxf(v[e], 0.f);
          } else if (p.act == TDEED_ACT_GELU) {
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = gelu_erf(v[e]);
          }
          T* cp = reinterpret_cast<T*>(p.C) + m * p.ldc + ch;
          T* cp2 = (p.C2 && ch < p.n2) ? reinterpret_cast<T*>(p.C2) + m * p.ldc2 + ch : nullptr;
#pragma unroll
          for (int h = 0; h < 8 / EPC; ++h) {
            float o[EPC];
#pragma unroll
            for (int e = 0; e < EPC; ++e) o[e] = v[h * EPC + e];
            Chunk<T>::store(cp + h * EPC, o);
            if (cp2) Chunk<T>::store(cp2 + h * EPC, o);
          }
        }
      }
    }
  }
}

// LDS budget of a weight slice when the whole matrix does not fit 64 KB (TDEED_WS_CAP_KB: 150 by default = the most a
// workgroup can have; with equal slices a 320 x 320 layer takes two slices of 104 KB)
static size_t ws_lds_cap() {
  static const size_t cap = [] {
    const char* e = getenv("TDEED_WS_CAP_KB");
    long kb = e ? atol(e) : 150;
    if (kb < 64) kb = 64;
    if (kb > 150) kb = 150;
    return (size_t)kb * 1024;
  }();
  return cap;
}
#define WS_LDS_CAP ws_lds_cap()
static bool ws_ks_ok(int KS) {
  return KS == 1 || KS == 2 || KS == 3 || KS == 4 || KS == 5 || KS == 6 || KS == 8 || KS == 10 || KS == 12;
}
// n-tiles per block slice: everything when the whole W fits 64 KB (several blocks per CU), otherwise the widest
// even slice that fits WS_LDS_CAP (one block per CU; the wide s4 layers)
static int ws_slice_tiles(int NT, int KS) {
  const size_t per_tile = (size_t)KS * 64 * 16 + 16 * 2 * sizeof(float);
  if ((size_t)NT * per_tile <= 64 * 1024) return NT;
  int nts = (int)(WS_LDS_CAP / per_tile) & ~1;
  if (nts > NT) nts = NT;
  if (nts >= 2) {                                   // equal slices (every workgroup of the grid gets the same share of W)
    const int nsl = (NT + nts - 1) / nts;
    nts = ((NT + nsl - 1) / nsl + 1) & ~1;
  }
  return nts;
}

template <typename T>
static int launch_gemm_ws(GemmWsP& p, hipStream_t st) {
  constexpr int EPC = Chunk<T>::N;
  const int KS = (p.K + 4 * EPC - 1) / (4 * EPC);
  if (!ws_ks_ok(KS)) { tdeed_set_error("gemm_ws: unsupported K=%d", p.K); return TDEED_ERR_ARG; }
  p.NTS = ws_slice_tiles(p.NT, KS);
  if (p.NTS < 2) { tdeed_set_error("gemm_ws: K=%d too deep for an LDS-resident weight slice", p.K); return TDEED_ERR_ARG; }
  const int nsl = (p.NT + p.NTS - 1) / p.NTS;
  const size_t smem = (size_t)p.NTS * KS * 64 * 16 + (size_t)p.NTS * 16 * 2 * sizeof(float);
  const bool wide8 = smem > 64 * 1024 && sizeof(T) == 2;
  const long nchunks = ((long)p.M + (wide8 ? 255 : 127)) / (wide8 ? 256 : 128);
  // persistent blocks, ALL resident: wide slices one per CU; otherwise as many per CU as LDS allows (at most 4).  Round 6: the
  // grid used to be 1024 whatever the weight's size -- a 152 x 152 weight (52.5 KB) leaves three workgroups per CU = 768
  // slots, and the 256 workgroups that did not fit started their share only when a first-round workgroup had finished its own
  long per_cu = smem > 64 * 1024 ? 1 : (long)((160 * 1024) / (smem ? smem : 1));
  if (per_cu > 4) per_cu = 4;
  if (per_cu < 1) per_cu = 1;
  long gx = 256 * per_cu / nsl;
  if (gx > nchunks) gx = nchunks;
  if (gx < 1) gx = 1;
  static TdDevOnce attr_set[16];
#define WS_CASE(k)                                                                                              \
  case k:                                                                                                       \
    if (smem > 64 * 1024 && !attr_set[k].get()) {                                                                     \
      if (hipFuncSetAttribute((const void*)gemm_ws_kernel<T, k, true>, hipFuncAttributeMaxDynamicSharedMemorySize, \
                              WS_LDS_CAP) != hipSuccess) {                                                      \
        tdeed_set_error("gemm_ws: hipFuncSetAttribute failed");                                                 \
        return TDEED_ERR_RUNTIME;                                                                               \
      }                                                                                                         \
      attr_set[k].set();                                                                                        \
    }                                                                                                           \
    if (wide8) {                                                                                                \
      if constexpr (sizeof(T) == 2 && (k == 10 || k == 12)) {                                                   \
        static TdDevOnce a8;                                                                                 \
        if (!a8.get()) { (void)hipFuncSetAttribute((const void*)gemm_ws_kernel<T, k, true, 8>,                        \
                                             hipFuncAttributeMaxDynamicSharedMemorySize, WS_LDS_CAP); a8.set(); } \
        hipLaunchKernelGGL((gemm_ws_kernel<T, k, true, 8>), dim3((unsigned)gx, nsl), dim3(512), smem, st, p);   \
        break;                                                                                                  \
      }                                                                                                         \
    }                                                                                                           \
    hipLaunchKernelGGL((gemm_ws_kernel<T, k, true>), dim3((unsigned)gx, nsl), dim3(256), smem, st, p);          \
    break;
  switch (KS) {
    WS_CASE(1) WS_CASE(2) WS_CASE(3) WS_CASE(4) WS_CASE(5) WS_CASE(6) WS_CASE(8) WS_CASE(10) WS_CASE(12)
    default: break;
  }
#undef WS_CASE
  TD_LAUNCH_CHECK("gemm_ws");
  return TDEED_OK;
}

extern "C" int tdeed_gemm_ws_fits(int K, int N, int dtype) {
  const int epc = dtype == TDEED_F32 ? 4 : 8;
  const int KS = (K + 4 * epc - 1) / (4 * epc);
  const int NT = (N + 31) / 32 * 2;
  if (!ws_ks_ok(KS)) return 0;
  const int nts = ws_slice_tiles(NT, KS);
  const size_t per_tile = (size_t)KS * 64 * 16 + 16 * 2 * sizeof(float);
  if ((size_t)NT * per_tile <= 64 * 1024) return 1;               // whole W in LDS, several workgroups per CU
  if (dtype == TDEED_F32 && (size_t)NT * per_tile <= 96 * 1024) return 1;   // parity mode: whole W, one workgroup per CU
  if (dtype == TDEED_BF16 && nts >= 4 && N <= 1024) return 2;     // one workgroup per CU: a slice (or all) of a wide W
  return 0;
}

extern "C" int tdeed_gemm_ws_fwd(const void* A, long lda, const void* A0, long lda0, int k0,
                                 const float* a_scale, int a_scale_rows, int M, int K, int N,
                                 const void* Wfrag, const float* scale, const float* shift, const void* R,
                                 long ldr, int act, void* C, long ldc, int gather_stride, int gather_hi,
                                 int gather_wi, int gather_ho, int gather_wo, void* C2, long ldc2, int n2, int dtype,
                                 void* stream) {
  TD_CHECK(A && Wfrag && C, "gemm_ws: null pointer");
  TD_CHECK(!C2 || (n2 > 0 && n2 % 8 == 0 && n2 <= N && ldc2 % 8 == 0 && ldc2 >= n2),
           "gemm_ws: bad second output n2=%d ldc2=%ld", n2, ldc2);
  TD_CHECK(M > 0 && K > 0 && N > 0, "gemm_ws: bad sizes M=%d K=%d N=%d", M, K, N);
  TD_CHECK(dtype == TDEED_F32 || dtype == TDEED_BF16, "gemm_ws: bad dtype %d", dtype);
  TD_CHECK(K % 8 == 0 && N % 8 == 0 && lda % 8 == 0 && ldc % 8 == 0, "gemm_ws: K, N, lda, ldc must be multiples of 8");
  TD_CHECK(!A0 || (k0 % 8 == 0 && lda0 % 8 == 0 && k0 <= K), "gemm_ws: bad splice");
  TD_CHECK(!R || ldr % 8 == 0, "gemm_ws: bad ldr");
  TD_CHECK(act >= 0 && act <= 2, "gemm_ws: bad act %d", act);
  TD_CHECK(tdeed_gemm_ws_fits(K, N, dtype), "gemm_ws: K=%d N=%d does not fit the weight-stationary kernel", K, N);
  if (gather_stride > 1)
    TD_CHECK(gather_hi > 0 && gather_wi > 0 && gather_ho > 0 && gather_wo > 0 && M % (gather_ho * gather_wo) == 0,
             "gemm_ws: bad gather geometry");
  GemmWsP p{};
  p.A = A; p.lda = lda; p.A0 = A0; p.lda0 = lda0; p.k0 = A0 ? k0 : 0;
  p.a_scale = a_scale; p.a_scale_rows = a_scale_rows > 0 ? a_scale_rows : 1;
  p.M = M; p.K = K; p.N = N; p.Wf = Wfrag; p.scale = scale; p.shift = shift;
  p.R = R; p.ldr = ldr; p.act = act; p.C = C; p.ldc = ldc;
  p.g_stride = gather_stride; p.g_hi = gather_hi; p.g_wi = gather_wi; p.g_ho = gather_ho; p.g_wo = gather_wo;
  p.NT = (N + 31) / 32 * 2;
  p.NTS = p.NT;
  p.C2 = C2; p.ldc2 = ldc2; p.n2 = C2 ? n2 : 0;
  hipStream_t st = (hipStream_t)stream;
  return dtype == TDEED_F32 ? launch_gemm_ws<float>(p, st) : launch_gemm_ws<bf16_t>(p, st);
}

// =============================================================================================
// Register-stationary contraction for the 320-wide layers of RegNetY-800MF (conv1 / conv3 of the eight s3 blocks: M =
// 156 800 .. 313 600 rows, K = N = 320).  The tiled kernel re-reads A once per 64-column tile (five times, from L2: 355 -
// 440 TFLOP/s), the sliced weight-stationary one twice from memory; here W (200 KB in bf16) lives in REGISTERS -- ten waves,
// each holding the 20 A-operand fragments of its two 16-channel tiles -- and the activations cross the chip once: 64-row
// tiles of A go global -> registers -> LDS (double buffered, row stride = 96 mod 128 bytes: conflict-free fragment reads),
// every wave reads the whole tile as B fragments (one read feeds two MFMAs) and owns 32 output channels of its 64 rows.
// SE gates of the (at most two) frames of a tile sit in a small LDS table that is filled one tile ahead; the gate-shift
// splice, BatchNorm, residual, ReLU and the compact second output are those of gemm_ws_kernel (same weight packing).
constexpr int RS_KS = 10, RS_NW = 10, RS_THR = RS_NW * 64, RS_ROWS = 64, RS_LD = RS_KS * 64 + 96;   // 736 B per row
constexpr int RS_TILE = RS_ROWS * RS_LD;
constexpr int RS_CPT = RS_ROWS * RS_KS * 4 / RS_THR;          // 16-byte chunks per thread per tile (4)
constexpr int RS_KSR = 8;                                     // k-steps of W held in registers; the last RS_KS - RS_KSR come
                                                              // from a wave-private LDS copy (16 registers for 40 % more LDS
                                                              // reads: the loop stays bound by the MFMA pipe, and the
                                                              // residual pieces of a tile fit without spilling)
constexpr int RS_WTAIL = RS_NW * 2 * (RS_KS - RS_KSR) * 64 * 16;      // 40 KB

// SE / RES / OUT2 are compile-time: a run-time branch inside the tile loop makes the wait-count pass drain every
// outstanding load at its join, i.e. wait for the NEXT tile's rows in the middle of this tile's MFMAs.
// STATS (training forward: the raw conv output's BatchNorm statistics): every lane sums the rounded values it stores and their
// squares per channel over ALL tiles of its workgroup; one fold over the 16 pixel lanes at the end leaves one partial row
// colpart[blockIdx.x][2][N] per workgroup -- the kernel is persistent, so the statistics cost 16 FMAs per 16-row tile per lane.
template <bool SE, bool RES, bool OUT2, bool STATS = false>
__global__ __launch_bounds__(RS_THR, 1) void gemm_rs_kernel(const GemmWsP p) {
  typedef bf16_t T;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* tiles = smem;                                        // [2][64][RS_LD]
  bf16x8* wtail = reinterpret_cast<bf16x8*>(smem + 2 * RS_TILE);      // [NW][2][KS - KSR][64] fragments, wave-private
  float* gt = reinterpret_cast<float*>(smem + 2 * RS_TILE + RS_WTAIL);    // [2][2][K] SE gates of the tile's two frames
  float* bnt = gt + 4 * RS_KS * 32;                                   // [2][N] scale, shift
  const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int px = lane & 15, q = lane >> 4;
  const int KP = RS_KS * 32;
  // this wave's weights: tile pair (2 wv, 2 wv + 1) = logical channels [32 wv, 32 wv + 32)
  bf16x8 wf[2][RS_KSR];
  bf16x8* wt = wtail + wv * 2 * (RS_KS - RS_KSR) * 64 + lane;
#pragma unroll
  for (int h = 0; h < 2; ++h)
#pragma unroll
    for (int ks = 0; ks < RS_KS; ++ks) {
      const bf16x8 f = reinterpret_cast<const bf16x8*>(p.Wf)[((long)(2 * wv + h) * RS_KS + ks) * 64 + lane];
      if (ks < RS_KSR) wf[h][ks] = f;
      else wt[(h * (RS_KS - RS_KSR) + ks - RS_KSR) * 64] = f;
    }
  for (int i = tid; i < 2 * p.N; i += RS_THR) {
    const int n = i % p.N;
    bnt[i] = i < p.N ? (p.scale ? p.scale[n] : 1.0f) : (p.shift ? p.shift[n] : 0.0f);
  }
  const long ntiles = ((long)p.M + RS_ROWS - 1) / RS_ROWS;
  // staging: RS_THR = 16 rows x 40 pieces, so a thread serves ONE 16-byte piece (ck) of rows r0, r0 + 16, r0 + 32, r0 + 48
  const int r0 = tid / (RS_KS * 4), ck = tid - r0 * (RS_KS * 4);
  const int kk = ck * 8;
  const bool kok = kk < p.K;
  const bool spl = p.A0 && kk < p.k0;
  const T* abase = spl ? reinterpret_cast<const T*>(p.A0) + kk : reinterpret_cast<const T*>(p.A) + (kok ? kk : 0);
  const long ald = spl ? p.lda0 : p.lda;
  u32x4 sv[RS_CPT];
  float gv = 0.f;                                                     // one gate of the next tile's table per thread
  const int gf = tid / KP, gk = tid - gf * KP;                        // (frame slot, k) of that gate: 2 KP = RS_THR threads
  auto gload = [&](long t) {
    const long m0 = t * RS_ROWS + r0;
#pragma unroll
    for (int j = 0; j < RS_CPT; ++j) {
      const long m = m0 + 16 * j;
      const bool ok = kok && m < p.M;
      const u32x4 v = *reinterpret_cast<const u32x4*>(abase + (ok ? m : 0) * ald);
      const u32x4 z = {0u, 0u, 0u, 0u};
      sv[j] = ok ? v : z;
    }
    if constexpr (SE) {
      const long f0 = (long)((unsigned)(t * RS_ROWS) / (unsigned)p.a_scale_rows);       // (rows < 2^31: 32-bit divisions)
      const long flast = (long)((unsigned)(p.M - 1) / (unsigned)p.a_scale_rows);
      gv = p.a_scale[min(f0 + gf, flast) * (long)p.K + min(gk, p.K - 1)];
    }
  };
  auto lstore = [&](int buf, long t) {
    unsigned char* base = tiles + buf * RS_TILE + r0 * RS_LD + ck * 16;
    if constexpr (SE) {
      float* g = gt + buf * 2 * KP;
      g[tid] = gv;                                                    // table of tile t (both frames), then scale the chunks
      __syncthreads();
      const long f0 = (long)((unsigned)(t * RS_ROWS) / (unsigned)p.a_scale_rows);
#pragma unroll
      for (int j = 0; j < RS_CPT; ++j) {
        const long m = min(t * RS_ROWS + r0 + 16 * j, (long)p.M - 1);
        const float* gr = g + (int)((long)((unsigned)m / (unsigned)p.a_scale_rows) - f0) * KP + kk;
        const f32x4 g0 = *reinterpret_cast<const f32x4*>(gr), g1 = *reinterpret_cast<const f32x4*>(gr + 4);
        bf16x8 x8 = *reinterpret_cast<const bf16x8*>(&sv[j]);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          x8[e] = (bf16_t)((float)x8[e] * g0[e]);
          x8[4 + e] = (bf16_t)((float)x8[4 + e] * g1[e]);
        }
        sv[j] = *reinterpret_cast<const u32x4*>(&x8);
      }
    }
#pragma unroll
    for (int j = 0; j < RS_CPT; ++j) *reinterpret_cast<u32x4*>(base + j * 16 * RS_LD) = sv[j];
  };
  const int ch = 32 * wv + 8 * q;                                     // this lane's 8 logical output channels
  long t = blockIdx.x;
  if (t < ntiles) {
    gload(t);
    lstore(0, t);
  }
  __syncthreads();
  int buf = 0;
  const float lo = p.act == TDEED_ACT_RELU ? 0.f : -3.0e38f;           // ReLU (or nothing) without a branch
  const int chc = min(ch, p.N - 8);
  [[maybe_unused]] float st1[STATS ? 8 : 1], st2[STATS ? 8 : 1];
  if constexpr (STATS) {
#pragma unroll
    for (int e = 0; e < 8; ++e) st1[e] = st2[e] = 0.f;
  }
  for (; t < ntiles; t += gridDim.x, buf ^= 1) {
    const long tn = t + gridDim.x;
    // residual pieces of this tile first, THEN the next tile's rows: vmcnt counts in order, so a wait for a residual piece
    // never has to drain the prefetch behind it
    // residual pieces of the tile in flight per lane: all four, or two where the compact second output needs the registers
    constexpr int RS_RPRE = (SE && RES && OUT2) ? 2 : 4;
    u32x4 rpre[RES ? RS_RPRE : 1];
    auto rload = [&](int mt) {
      const long m = min(t * RS_ROWS + mt * 16 + px, (long)p.M - 1);
      rpre[mt % RS_RPRE] = *reinterpret_cast<const u32x4*>(reinterpret_cast<const T*>(p.R) + m * p.ldr + chc);
    };
    if constexpr (RES) {
#pragma unroll
      for (int mt = 0; mt < RS_RPRE; ++mt) rload(mt);
    }
    if (tn < ntiles) gload(tn);                                       // next tile travels while this one is multiplied
    const unsigned char* base = tiles + buf * RS_TILE;
    // (the STATS form stores the raw contraction: no BatchNorm fold, and 16 registers for the column sums instead)
    const float* bq0 = STATS ? bnt : bnt + chc;
    // one 16-row tile at a time: two accumulators (the wave's two channel tiles) alternate on the MFMA pipe
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) {
      f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
      const long m = t * RS_ROWS + mt * 16 + px;
      const unsigned char* ar = base + (mt * 16 + px) * RS_LD + 16 * q;
#pragma unroll
      for (int ks = 0; ks < RS_KS; ++ks) {
        const bf16x8 a = *reinterpret_cast<const bf16x8*>(ar + 64 * ks);
        const bf16x8 w0 = ks < RS_KSR ? wf[0][ks < RS_KSR ? ks : 0] : wt[(ks - RS_KSR) * 64];
        const bf16x8 w1 = ks < RS_KSR ? wf[1][ks < RS_KSR ? ks : 0] : wt[((RS_KS - RS_KSR) + ks - RS_KSR) * 64];
        acc0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w0, a, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w1, a, acc1, 0, 0, 0);
      }
      // the BatchNorm fold of the lane's 8 channels comes from LDS again for every 16-row tile (an opaque pointer keeps the
      // compiler from holding its 16 registers across the MFMA loops, where the budget of 168 is needed for W and the prefetches)
      const float* bq = bq0;
      if constexpr (SE || RES) asm volatile("" : "+v"(bq));
      const f32x4 s0 = *reinterpret_cast<const f32x4*>(bq), s1 = *reinterpret_cast<const f32x4*>(bq + 4);
      const f32x4 h0 = *reinterpret_cast<const f32x4*>(bq + p.N), h1 = *reinterpret_cast<const f32x4*>(bq + p.N + 4);
      float v[8];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        v[r] = STATS ? acc0[r] : acc0[r] * s0[r] + h0[r];
        v[4 + r] = STATS ? acc1[r] : acc1[r] * s1[r] + h1[r];
      }
      if constexpr (RES) {
        float rv[8];
        Chunk<T>::load(reinterpret_cast<const T*>(&rpre[mt % RS_RPRE]), rv);
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] += rv[e];
        if (mt + RS_RPRE < 4) rload(mt + RS_RPRE);
      }
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = fmaxf(v[e], lo);
      if (ch < p.N && m < p.M) {
        Chunk<T>::store(reinterpret_cast<T*>(p.C) + m * p.ldc + ch, v);
        if constexpr (OUT2) {
          if (ch < p.n2) Chunk<T>::store(reinterpret_cast<T*>(p.C2) + m * p.ldc2 + ch, v);
        }
        if constexpr (STATS) {
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            const float r = round_to<T>(v[e]);
            st1[e] += r;
            st2[e] = fmaf(r, r, st2[e]);
          }
        }
      }
    }
    if (tn < ntiles) lstore(buf ^ 1, tn);     // (every wave is past its reads of buffer buf ^ 1: they ended before the last barrier)
    __syncthreads();
  }
  if constexpr (STATS) {
    // fold over the 16 pixel lanes of each quarter (lanes l, l ^ 1, l ^ 2, l ^ 4, l ^ 8 share q), fixed order
#pragma unroll
    for (int e = 0; e < 8; ++e) {
#pragma unroll
      for (int o = 1; o < 16; o <<= 1) {
        st1[e] += __shfl_xor(st1[e], o, 64);
        st2[e] += __shfl_xor(st2[e], o, 64);
      }
    }
    if (px == 0 && ch < p.N) {
      float* dst = p.colpart + (long)blockIdx.x * 2 * p.N + ch;
      *reinterpret_cast<f32x4*>(dst) = (f32x4){st1[0], st1[1], st1[2], st1[3]};
      *reinterpret_cast<f32x4*>(dst + 4) = (f32x4){st1[4], st1[5], st1[6], st1[7]};
      *reinterpret_cast<f32x4*>(dst + p.N) = (f32x4){st2[0], st2[1], st2[2], st2[3]};
      *reinterpret_cast<f32x4*>(dst + p.N + 4) = (f32x4){st2[4], st2[5], st2[6], st2[7]};
    }
  }
}

// Input-gradient contraction of a training bottleneck's conv1 (tdeed_gemm_dgrad) on the register-stationary scheme for
// K = N = 320 over many rows (the s3 identity blocks of RegNetY-800MF): C = ((A @ W^T) + R) * [mask > 0], columns [0, n2) of the
// raw product to C2 with only the residual left in C there (the gate-shift columns), and the gradient sink's column sums of
// what is stored -- r and r * (bz - bmean) -- one partial row bpart[blockIdx.x][3][N] per persistent workgroup (row 2: zeros;
// sinks with a second statistics map stay on the tiled kernel).  The epilogue operands of a 16-row tile (shortcut gradient,
// mask, statistics map: 3 x 16 bytes per lane) are requested one tile ahead; the means live in LDS and are read through an
// opaque pointer per tile (register budget 168, see gemm_rs_kernel).  231 vs 300 us per call at M = 313 600.
struct GemmRsBwdP {
  const void* A; long lda; int M; const void* Wf;
  const void* R; long ldr; void* C; long ldc; void* C2; long ldc2; int n2;
  const void* mask; long ldmask; const void* bz; long ldbz; const float* bmean; float* bpart;
};
__global__ __launch_bounds__(RS_THR, 1) void gemm_rs_bwd_kernel(const GemmRsBwdP p) {
  typedef bf16_t T;
  constexpr int N = 320;
  // one k-step fewer of W in registers than the forward kernel (7 of 10: 60 KB of wave-private LDS copies, 157 KB in all):
  // the epilogue's operands and column sums need the 8 registers
  constexpr int KSR = 7, WTAIL = RS_NW * 2 * (RS_KS - KSR) * 64 * 16;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* tiles = smem;                                        // [2][64][RS_LD]
  bf16x8* wtail = reinterpret_cast<bf16x8*>(smem + 2 * RS_TILE);      // [NW][2][KS - KSR][64] fragments, wave-private
  float* bmt = reinterpret_cast<float*>(smem + 2 * RS_TILE + WTAIL);   // [N] means of the statistics map
  const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int px = lane & 15, q = lane >> 4;
  bf16x8 wf[2][KSR];
  bf16x8* wt = wtail + wv * 2 * (RS_KS - KSR) * 64 + lane;
#pragma unroll
  for (int h = 0; h < 2; ++h)
#pragma unroll
    for (int ks = 0; ks < RS_KS; ++ks) {
      const bf16x8 f = reinterpret_cast<const bf16x8*>(p.Wf)[((long)(2 * wv + h) * RS_KS + ks) * 64 + lane];
      if (ks < KSR) wf[h][ks] = f;
      else wt[(h * (RS_KS - KSR) + ks - KSR) * 64] = f;
    }
  for (int i = tid; i < N; i += RS_THR) bmt[i] = p.bpart ? p.bmean[i] : 0.f;
  const long ntiles = ((long)p.M + RS_ROWS - 1) / RS_ROWS;
  const int r0 = tid / (RS_KS * 4), ck = tid - r0 * (RS_KS * 4);
  const T* abase = reinterpret_cast<const T*>(p.A) + ck * 8;
  u32x4 sv[RS_CPT];
  auto gload = [&](long t) {
    const long m0 = t * RS_ROWS + r0;
#pragma unroll
    for (int j = 0; j < RS_CPT; ++j) {
      const long m = m0 + 16 * j;
      const bool ok = m < p.M;
      const u32x4 v = *reinterpret_cast<const u32x4*>(abase + (ok ? m : 0) * p.lda);
      const u32x4 z = {0u, 0u, 0u, 0u};
      sv[j] = ok ? v : z;
    }
  };
  auto lstore = [&](int buf) {
    unsigned char* base = tiles + buf * RS_TILE + r0 * RS_LD + ck * 16;
#pragma unroll
    for (int j = 0; j < RS_CPT; ++j) *reinterpret_cast<u32x4*>(base + j * 16 * RS_LD) = sv[j];
  };
  const int ch = 32 * wv + 8 * q;                                     // this lane's 8 logical output channels
  long t = blockIdx.x;
  if (t < ntiles) {
    gload(t);
    lstore(0);
  }
  __syncthreads();
  int buf = 0;
  float st1[8], st2[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) st1[e] = st2[e] = 0.f;
  constexpr int NPRE = 1;                                             // 16-row tiles whose epilogue operands are in flight (2: 9 VGPRs spill, 257 vs 231 us per call)
  for (; t < ntiles; t += gridDim.x, buf ^= 1) {
    const long tn = t + gridDim.x;
    u32x4 er[NPRE], em[NPRE], ez[NPRE];
    auto eload = [&](int mt) {
      const long m = min(t * RS_ROWS + mt * 16 + px, (long)p.M - 1);
      er[mt % NPRE] = *reinterpret_cast<const u32x4*>(reinterpret_cast<const T*>(p.R) + m * p.ldr + ch);
      em[mt % NPRE] = *reinterpret_cast<const u32x4*>(reinterpret_cast<const T*>(p.mask) + m * p.ldmask + ch);
      if (p.bpart) ez[mt % NPRE] = *reinterpret_cast<const u32x4*>(reinterpret_cast<const T*>(p.bz) + m * p.ldbz + ch);
    };
#pragma unroll
    for (int mt = 0; mt < NPRE; ++mt) eload(mt);
    if (tn < ntiles) gload(tn);                                       // next tile travels while this one is multiplied
    const unsigned char* base = tiles + buf * RS_TILE;
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) {
      f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
      const long m = t * RS_ROWS + mt * 16 + px;
      const unsigned char* ar = base + (mt * 16 + px) * RS_LD + 16 * q;
#pragma unroll
      for (int ks = 0; ks < RS_KS; ++ks) {
        const bf16x8 a = *reinterpret_cast<const bf16x8*>(ar + 64 * ks);
        const bf16x8 w0 = ks < KSR ? wf[0][ks < KSR ? ks : 0] : wt[(ks - KSR) * 64];
        const bf16x8 w1 = ks < KSR ? wf[1][ks < KSR ? ks : 0] : wt[((RS_KS - KSR) + ks - KSR) * 64];
        acc0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w0, a, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w1, a, acc1, 0, 0, 0);
      }
      const bool mok = m < p.M;
      const bool gs = p.C2 && ch < p.n2;             // gate-shift columns: the module's gradient leaves, C keeps the residual
      if (gs && mok) {
        float v[8];
#pragma unroll
        for (int r = 0; r < 4; ++r) { v[r] = acc0[r]; v[4 + r] = acc1[r]; }
        Chunk<T>::store(reinterpret_cast<T*>(p.C2) + m * p.ldc2 + ch, v);
      }
      // element pairs straight out of the packed words (bf16 -> fp32 is a shift): nothing but the pair in flight is live
      const u32x4 rw = er[mt % NPRE], mw = em[mt % NPRE], zw = ez[mt % NPRE];
      const float* bp = bmt + ch;
      asm volatile("" : "+v"(bp));
      u32x4 ow;
#pragma unroll
      for (int w = 0; w < 4; ++w) {
        const float a0 = gs ? 0.f : (w < 2 ? acc0[2 * w] : acc1[2 * w - 4]);
        const float a1 = gs ? 0.f : (w < 2 ? acc0[2 * w + 1] : acc1[2 * w - 3]);
        const float r0 = __uint_as_float(rw[w] << 16), r1 = __uint_as_float(rw[w] & 0xffff0000u);
        const float m0 = __uint_as_float(mw[w] << 16), m1 = __uint_as_float(mw[w] & 0xffff0000u);
        const bf16_t o0 = (bf16_t)(m0 > 0.f ? a0 + r0 : 0.f), o1 = (bf16_t)(m1 > 0.f ? a1 + r1 : 0.f);
        const unsigned short u0 = __builtin_bit_cast(unsigned short, o0), u1 = __builtin_bit_cast(unsigned short, o1);
        ow[w] = (unsigned)u0 | ((unsigned)u1 << 16);
        if (p.bpart && mok) {
          const float q0 = (float)o0, q1 = (float)o1;
          const float z0 = __uint_as_float(zw[w] << 16), z1 = __uint_as_float(zw[w] & 0xffff0000u);
          st1[2 * w] += q0;
          st1[2 * w + 1] += q1;
          st2[2 * w] = fmaf(q0, z0 - bp[2 * w], st2[2 * w]);
          st2[2 * w + 1] = fmaf(q1, z1 - bp[2 * w + 1], st2[2 * w + 1]);
        }
      }
      if (mok) *reinterpret_cast<u32x4*>(reinterpret_cast<T*>(p.C) + m * p.ldc + ch) = ow;
      if (mt + NPRE < 4) eload(mt + NPRE);
    }
    if (tn < ntiles) lstore(buf ^ 1);
    __syncthreads();
  }
  if (p.bpart) {
#pragma unroll
    for (int e = 0; e < 8; ++e) {
#pragma unroll
      for (int o = 1; o < 16; o <<= 1) {
        st1[e] += __shfl_xor(st1[e], o, 64);
        st2[e] += __shfl_xor(st2[e], o, 64);
      }
    }
    if (px == 0) {
      float* dst = p.bpart + (long)blockIdx.x * 3 * N + ch;
      *reinterpret_cast<f32x4*>(dst) = (f32x4){st1[0], st1[1], st1[2], st1[3]};
      *reinterpret_cast<f32x4*>(dst + 4) = (f32x4){st1[4], st1[5], st1[6], st1[7]};
      *reinterpret_cast<f32x4*>(dst + N) = (f32x4){st2[0], st2[1], st2[2], st2[3]};
      *reinterpret_cast<f32x4*>(dst + N + 4) = (f32x4){st2[4], st2[5], st2[6], st2[7]};
      *reinterpret_cast<f32x4*>(dst + 2 * N) = (f32x4){0.f, 0.f, 0.f, 0.f};
      *reinterpret_cast<f32x4*>(dst + 2 * N + 4) = (f32x4){0.f, 0.f, 0.f, 0.f};
    }
  }
}

extern "C" int tdeed_gemm_rs_fits(int M, int K, int N) { return (K == 320 && N == 320 && M > 0) ? 1 : 0; }

// same contract as tdeed_gemm_ws_fwd (weights from engine.pack_ws_weights, bf16) without the stride-2 row gather
extern "C" int tdeed_gemm_rs_fwd(const void* A, long lda, const void* A0, long lda0, int k0, const float* a_scale,
                                 int a_scale_rows, int M, int K, int N, const void* Wfrag, const float* scale,
                                 const float* shift, const void* R, long ldr, int act, void* C, long ldc, void* C2, long ldc2,
                                 int n2, void* stream) {
  TD_CHECK(A && Wfrag && C, "gemm_rs: null pointer");
  TD_CHECK(tdeed_gemm_rs_fits(M, K, N), "gemm_rs: M=%d K=%d N=%d unsupported (K = N = 320)", M, K, N);
  TD_CHECK(lda % 8 == 0 && ldc % 8 == 0 && (!R || ldr % 8 == 0), "gemm_rs: row strides must be multiples of 8");
  TD_CHECK(!A0 || (k0 % 8 == 0 && lda0 % 8 == 0 && k0 <= K), "gemm_rs: bad splice");
  TD_CHECK(!C2 || (n2 > 0 && n2 % 8 == 0 && n2 <= N && ldc2 % 8 == 0 && ldc2 >= n2), "gemm_rs: bad second output");
  TD_CHECK(!a_scale || (a_scale_rows >= RS_ROWS), "gemm_rs: a tile of %d rows must span at most two frames (rows per frame %d)",
           RS_ROWS, a_scale_rows);
  GemmWsP p{};
  p.A = A; p.lda = lda; p.A0 = A0; p.lda0 = lda0; p.k0 = A0 ? k0 : 0;
  p.a_scale = a_scale; p.a_scale_rows = a_scale_rows > 0 ? a_scale_rows : 1;
  p.M = M; p.K = K; p.N = N; p.Wf = Wfrag; p.scale = scale; p.shift = shift;
  p.R = R; p.ldr = ldr; p.act = act; p.C = C; p.ldc = ldc;
  p.g_stride = 1; p.g_hi = p.g_wi = p.g_ho = p.g_wo = 0;
  p.NT = 2 * RS_NW; p.NTS = p.NT;
  p.C2 = C2; p.ldc2 = ldc2; p.n2 = C2 ? n2 : 0;
  const size_t smem = (size_t)2 * RS_TILE + RS_WTAIL + (size_t)4 * RS_KS * 32 * sizeof(float) + (size_t)2 * N * sizeof(float);
  TD_CHECK(act == TDEED_ACT_NONE || act == TDEED_ACT_RELU, "gemm_rs: ReLU or no activation only");
  long grid = ((long)M + RS_ROWS - 1) / RS_ROWS;
  if (grid > 256) grid = 256;
  static TdDevOnce attr[8];
#define TD_RS(Sv, Rv, Ov)                                                                                              \
  do {                                                                                                                 \
    const int ix = (Sv ? 4 : 0) + (Rv ? 2 : 0) + (Ov ? 1 : 0);                                                         \
    if (!attr[ix].get()) {                                                                                               \
      if (hipFuncSetAttribute((const void*)gemm_rs_kernel<Sv, Rv, Ov>, hipFuncAttributeMaxDynamicSharedMemorySize,     \
                              160 * 1024) != hipSuccess) {                                                             \
        tdeed_set_error("gemm_rs: hipFuncSetAttribute failed");                                                        \
        return TDEED_ERR_RUNTIME;                                                                                      \
      }                                                                                                                \
      attr[ix].set();                                                                                                 \
    }                                                                                                                  \
    hipLaunchKernelGGL((gemm_rs_kernel<Sv, Rv, Ov>), dim3((unsigned)grid), dim3(RS_THR), smem, (hipStream_t)stream, p); \
  } while (0)
  const bool se = a_scale != nullptr, rs = R != nullptr, o2 = C2 != nullptr;
  if (se) { if (rs) { if (o2) TD_RS(true, true, true); else TD_RS(true, true, false); }
            else { if (o2) TD_RS(true, false, true); else TD_RS(true, false, false); } }
  else { if (rs) { if (o2) TD_RS(false, true, true); else TD_RS(false, true, false); }
         else { if (o2) TD_RS(false, false, true); else TD_RS(false, false, false); } }
#undef TD_RS
  TD_LAUNCH_CHECK("gemm_rs");
  return TDEED_OK;
}

// Training forward of the same shape: raw output (no BatchNorm fold, no activation) + its column statistics, one partial row
// per workgroup: colpart fp32 [tdeed_gemm_rs_grid(M)][2][N] (sums | sums of squares of the stored, rounded values) -- what
// tdeed_gemm_fwd(colpart) leaves per 128-row tile, here per persistent workgroup.
extern "C" int tdeed_gemm_rs_grid(int M) {
  long grid = ((long)M + RS_ROWS - 1) / RS_ROWS;
  return (int)(grid > 256 ? 256 : grid);
}
extern "C" int tdeed_gemm_rs_stats_fwd(const void* A, long lda, const void* A0, long lda0, int k0, int M, int K, int N,
                                       const void* Wfrag, void* C, long ldc, float* colpart, void* stream) {
  TD_CHECK(A && Wfrag && C && colpart, "gemm_rs_stats: null pointer");
  TD_CHECK(tdeed_gemm_rs_fits(M, K, N), "gemm_rs_stats: M=%d K=%d N=%d unsupported (K = N = 320)", M, K, N);
  TD_CHECK(lda % 8 == 0 && ldc % 8 == 0, "gemm_rs_stats: row strides must be multiples of 8");
  TD_CHECK(!A0 || (k0 % 8 == 0 && lda0 % 8 == 0 && k0 <= K), "gemm_rs_stats: bad splice");
  GemmWsP p{};
  p.A = A; p.lda = lda; p.A0 = A0; p.lda0 = lda0; p.k0 = A0 ? k0 : 0;
  p.a_scale_rows = 1;
  p.M = M; p.K = K; p.N = N; p.Wf = Wfrag;
  p.act = TDEED_ACT_NONE; p.C = C; p.ldc = ldc;
  p.g_stride = 1;
  p.NT = 2 * RS_NW; p.NTS = p.NT;
  p.colpart = colpart;
  const size_t smem = (size_t)2 * RS_TILE + RS_WTAIL + (size_t)4 * RS_KS * 32 * sizeof(float) + (size_t)2 * N * sizeof(float);
  static TdDevOnce attr;
  if (!attr.get()) {
    if (hipFuncSetAttribute((const void*)gemm_rs_kernel<false, false, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize,
                            160 * 1024) != hipSuccess) {
      tdeed_set_error("gemm_rs_stats: hipFuncSetAttribute failed");
      return TDEED_ERR_RUNTIME;
    }
    attr.set();
  }
  hipLaunchKernelGGL((gemm_rs_kernel<false, false, false, true>), dim3((unsigned)tdeed_gemm_rs_grid(M)), dim3(RS_THR), smem,
                     (hipStream_t)stream, p);
  TD_LAUNCH_CHECK("gemm_rs_stats");
  return TDEED_OK;
}

// tdeed_gemm_dgrad for K = N = 320 on the register-stationary scheme (gemm_rs_bwd_kernel): Wfrag = pack_ws_weights of the
// [N][K] matrix the product is taken with (W^T of the layer); R (the shortcut gradient, [M][N] rows) and mask are required, no
// stride-2 residual and no second statistics map (those calls stay on tdeed_gemm_dgrad); bpart fp32 [tdeed_gemm_rs_grid(M)][3][N].
extern "C" int tdeed_gemm_dgrad_rs(const void* A, long lda, int M, int K, int N, const void* Wfrag, const void* R, long ldr,
                                   void* C, long ldc, void* C2, long ldc2, int n2, const void* mask, long ldmask,
                                   const void* bz, long ldbz, const float* bmean, float* bpart, void* stream) {
  TD_CHECK(A && Wfrag && R && C && mask, "gemm_dgrad_rs: null pointer");
  TD_CHECK(tdeed_gemm_rs_fits(M, K, N), "gemm_dgrad_rs: M=%d K=%d N=%d unsupported (K = N = 320)", M, K, N);
  TD_CHECK(lda % 8 == 0 && ldc % 8 == 0 && ldr % 8 == 0 && ldmask % 8 == 0, "gemm_dgrad_rs: row strides must be multiples of 8");
  TD_CHECK(!C2 || (n2 > 0 && n2 % 8 == 0 && n2 <= N && ldc2 % 8 == 0 && ldc2 >= n2), "gemm_dgrad_rs: bad second output");
  TD_CHECK(!bpart || (bz && bmean && ldbz % 8 == 0), "gemm_dgrad_rs: statistics operands missing");
  GemmRsBwdP p{};
  p.A = A; p.lda = lda; p.M = M; p.Wf = Wfrag; p.R = R; p.ldr = ldr; p.C = C; p.ldc = ldc;
  p.C2 = C2; p.ldc2 = ldc2; p.n2 = C2 ? n2 : 0;
  p.mask = mask; p.ldmask = ldmask; p.bz = bz; p.ldbz = ldbz; p.bmean = bmean; p.bpart = bpart;
  const size_t smem = (size_t)2 * RS_TILE + (size_t)RS_NW * 2 * (RS_KS - 7) * 64 * 16 + (size_t)N * sizeof(float);
  static TdDevOnce attr;
  if (!attr.get()) {
    if (hipFuncSetAttribute((const void*)gemm_rs_bwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) !=
        hipSuccess) {
      tdeed_set_error("gemm_dgrad_rs: hipFuncSetAttribute failed");
      return TDEED_ERR_RUNTIME;
    }
    attr.set();
  }
  hipLaunchKernelGGL(gemm_rs_bwd_kernel, dim3((unsigned)tdeed_gemm_rs_grid(M)), dim3(RS_THR), smem, (hipStream_t)stream, p);
  TD_LAUNCH_CHECK("gemm_dgrad_rs");
  return TDEED_OK;
}


// =============================================================================================
// Split-K contraction for the short sequences of the SGP encoder-decoder (M = B*T of a few hundred rows, K up to
// 6C): with so few rows a tiled kernel has ~40 workgroups that each walk K in 20+ dependent global->LDS round
// trips (30-43 us for 0.9 GFLOP).  Here K is cut into chunks of <=192: a workgroup issues ALL loads of its
// 64 x 64 x 192 brick at once (12 x 16 B per lane), runs 6 MFMA k-steps out of LDS and writes an fp32 partial;
// a second launch sums the partials and applies scale / shift / residual / activation.  bf16 operands.
constexpr int SK_KC = 192, SK_RS = SK_KC * 2 + 16;     // K chunk, LDS row stride (bytes; the skew spreads banks)

__global__ __launch_bounds__(256) void gemm_splitk_kernel(const bf16_t* __restrict__ A, long lda,
                                                          const bf16_t* __restrict__ W, long ldw, int M, int N, int K,
                                                          int n_tiles, float* __restrict__ part) {
  __shared__ __attribute__((aligned(16))) unsigned char sa[64 * SK_RS];
  __shared__ __attribute__((aligned(16))) unsigned char sw[64 * SK_RS];
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, pl = lane & 15, q = lane >> 4;
  const int tile = xcd_logical_id(blockIdx.x, gridDim.x);
  const int n0 = (tile % n_tiles) * 64, m0 = (tile / n_tiles) * 64;
  const int z = blockIdx.y, k0 = z * SK_KC;
  constexpr int PPR = SK_KC / 8;                       // 16-byte pieces per row
  // branch-free: every lane loads from a clamped (always valid) address so all 12 loads are in flight together;
  // out-of-range pieces are zeroed on the way into LDS
  u32x4 va[6], vw[6];
#pragma unroll
  for (int j = 0; j < 6; ++j) {
    const int i = tid + 256 * j;
    const int row = i / PPR, pc = i - row * PPR;
    const int k = min(k0 + pc * 8, K - 8);
    va[j] = *reinterpret_cast<const u32x4*>(A + (long)min(m0 + row, M - 1) * lda + k);
    vw[j] = *reinterpret_cast<const u32x4*>(W + (long)min(n0 + row, N - 1) * ldw + k);
  }
#pragma unroll
  for (int j = 0; j < 6; ++j) {
    const int i = tid + 256 * j;
    const int row = i / PPR, pc = i - row * PPR;
    const bool kin = k0 + pc * 8 < K;
    const u32x4 zero = {0u, 0u, 0u, 0u};
    *reinterpret_cast<u32x4*>(sa + row * SK_RS + pc * 16) = (kin && m0 + row < M) ? va[j] : zero;
    *reinterpret_cast<u32x4*>(sw + row * SK_RS + pc * 16) = (kin && n0 + row < N) ? vw[j] : zero;
  }
  __syncthreads();
  f32x4 acc[4];
#pragma unroll
  for (int mt = 0; mt < 4; ++mt) acc[mt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int ks = 0; ks < SK_KC / 32; ++ks) {
    const bf16x8 wf = *reinterpret_cast<const bf16x8*>(sw + (wv * 16 + pl) * SK_RS + (ks * 32 + q * 8) * 2);
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) {
      const bf16x8 af = *reinterpret_cast<const bf16x8*>(sa + (mt * 16 + pl) * SK_RS + (ks * 32 + q * 8) * 2);
      acc[mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf, af, acc[mt], 0, 0, 0);
    }
  }
  const int n = n0 + wv * 16 + 4 * q;
  if (n < N) {
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) {
      const int m = m0 + mt * 16 + pl;
      if (m < M) *reinterpret_cast<f32x4*>(part + ((long)z * M + m) * N + n) = acc[mt];
    }
  }
}

__global__ __launch_bounds__(256) void gemm_splitk_reduce_kernel(const float* __restrict__ part, int S, int M, int N,
                                                                 const float* __restrict__ scale,
                                                                 const float* __restrict__ shift,
                                                                 const bf16_t* __restrict__ R, long ldr, int act,
                                                                 bf16_t* __restrict__ C, long ldc) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  const int n4 = N >> 2;
  if (i >= (long)M * n4) return;
  const int m = (int)(i / n4), n = (int)(i - (long)m * n4) * 4;
  f32x4 v = *reinterpret_cast<const f32x4*>(part + (long)m * N + n);
  // eight partials in flight per thread (round 6: one `load; s_waitcnt vmcnt(0); add` per split was 16 - 24 dependent round
  // trips at K = 4C .. 6C); the additions keep their order
  for (int s0 = 1; s0 < S; s0 += 8) {
    f32x4 t[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) t[k] = *reinterpret_cast<const f32x4*>(part + ((long)min(s0 + k, S - 1) * M + m) * N + n);
    TD_ISSUE_FENCE();
#pragma unroll
    for (int k = 0; k < 8; ++k)
      if (s0 + k < S) { v[0] += t[k][0]; v[1] += t[k][1]; v[2] += t[k][2]; v[3] += t[k][3]; }
  }
  bf16x4 rr;
  if (R) rr = *reinterpret_cast<const bf16x4*>(R + (long)m * ldr + n);
  bf16x4 o;
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    float x = v[e];
    if (scale) x *= scale[n + e];
    if (shift) x += shift[n + e];
    if (R) x += (float)rr[e];
    if (act == TDEED_ACT_RELU) x = fmaxf(x, 0.f);
    else if (act == TDEED_ACT_GELU) x = gelu_erf(x);
    o[e] = (bf16_t)x;
  }
  *reinterpret_cast<bf16x4*>(C + (long)m * ldc + n) = o;
}

extern "C" int tdeed_gemm_splitk_splits(int K) { return (K + SK_KC - 1) / SK_KC; }

// the partial products alone (workspace [tdeed_gemm_splitk_splits(K)][M][N] fp32): for callers that fold them in a kernel of
// their own (tdeed_sgp_fold_cols: bias + GELU + the per-channel sums the next GroupNorm needs)
extern "C" int tdeed_gemm_splitk_partials(const void* A, long lda, int M, int K, int N, const void* W, long ldw,
                                          float* workspace, void* stream) {
  TD_CHECK(A && W && workspace, "gemm_splitk_partials: null pointer");
  TD_CHECK(M > 0 && K > 0 && N > 0, "gemm_splitk_partials: bad sizes M=%d K=%d N=%d", M, K, N);
  TD_CHECK(K % 8 == 0 && N % 8 == 0 && lda % 8 == 0 && ldw % 8 == 0,
           "gemm_splitk_partials: K=%d N=%d and the row strides must be multiples of 8", K, N);
  const int S = tdeed_gemm_splitk_splits(K);
  TD_CHECK(S <= 65535, "gemm_splitk_partials: K=%d too large", K);
  const int n_tiles = (N + 63) / 64, m_tiles = (M + 63) / 64;
  hipLaunchKernelGGL(gemm_splitk_kernel, dim3(n_tiles * m_tiles, S), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)A,
                     lda, (const bf16_t*)W, ldw, M, N, K, n_tiles, workspace);
  TD_LAUNCH_CHECK("gemm_splitk_partials");
  return TDEED_OK;
}

extern "C" int tdeed_gemm_splitk_fwd(const void* A, long lda, int M, int K, int N, const void* W, long ldw,
                                     const float* scale, const float* shift, const void* R, long ldr, int act,
                                     void* C, long ldc, float* workspace, void* stream) {
  TD_CHECK(A && W && C && workspace, "gemm_splitk: null pointer");
  TD_CHECK(M > 0 && K > 0 && N > 0, "gemm_splitk: bad sizes M=%d K=%d N=%d", M, K, N);
  TD_CHECK(K % 8 == 0 && N % 8 == 0 && lda % 8 == 0 && ldw % 8 == 0 && ldc % 8 == 0 && (!R || ldr % 8 == 0),
           "gemm_splitk: K=%d N=%d and the row strides must be multiples of 8", K, N);
  TD_CHECK(act >= 0 && act <= 2, "gemm_splitk: bad act %d", act);
  const int S = tdeed_gemm_splitk_splits(K);
  TD_CHECK(S <= 65535, "gemm_splitk: K=%d too large", K);
  const int n_tiles = (N + 63) / 64, m_tiles = (M + 63) / 64;
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(gemm_splitk_kernel, dim3(n_tiles * m_tiles, S), dim3(256), 0, st, (const bf16_t*)A, lda,
                     (const bf16_t*)W, ldw, M, N, K, n_tiles, workspace);
  TD_LAUNCH_CHECK("gemm_splitk");
  const long items = (long)M * (N / 4);
  hipLaunchKernelGGL(gemm_splitk_reduce_kernel, dim3((unsigned)((items + 255) / 256)), dim3(256), 0, st, workspace, S,
                     M, N, scale, shift, (const bf16_t*)R, ldr, act, (bf16_t*)C, ldc);
  TD_LAUNCH_CHECK("gemm_splitk_reduce");
  return TDEED_OK;
}
