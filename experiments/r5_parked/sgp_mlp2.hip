// GroupNorm + MLP + residual of an SGP block / mixer for narrow feature dimensions (C <= 384: RegNetY-200MF, C = 368),
// decomposed for a chip of 256 CUs at a few hundred rows (reference: /root/reference/model/modules.py:186, 316:
// out = y + mlp(gn(y)), mlp = Conv1d(C, 4C, 1) -> GELU -> Conv1d(4C, C, 1); GroupNorm(16, C) modules.py:115, 216).
//
// Why another form than sgp_mlp_kernel (sgp_fused.hip): at R = 800 rows that kernel runs 13 row tiles x 4 hidden
// chunks = 52 workgroups, each streaming 542 KB of weights through ONE CU's load path (~25 GB/s per CU: 24 us) while
// 200 CUs idle.  Here a workgroup owns 16*MT rows and ONE SLICE of 128 hidden units (one 16-unit tile per wave):
//
//     Hs  = GELU(W1[slice] . GN(y rows) + b1[slice])        K = C      (12 k-steps, weights prefetched whole)
//     P_s = W2[:, slice] . Hs                               K = 128    (4 k-steps, weights prefetched whole)
//
// so a workgroup streams 192 KB of weights, ALL of whose loads are issued before anything else happens (one exposed
// memory latency per workgroup instead of a ring of them), and R/64 x 12 = 156 workgroups share the weight stream.
// The fc2 products of the 12 slices are fp32 partials [S][R][C]; sgp_fold_rows_kernel (one wave per row) sums them in
// a fixed order, adds b2 and the residual, stores the bf16 row -- and, having the whole row in one wave, leaves the
// LayerNorm statistics (mean, rstd over C) of that output row for the next block's front kernel, which then no longer
// re-reads its clip's (T x C) slab to derive them.  No float atomics: results are bit-reproducible.
#include "common.h"
#include <stdlib.h>

namespace {

constexpr int M2_NW = 8;            // waves per workgroup
constexpr int M2_HS = 128;          // hidden units per slice: one 16-unit tile per wave
constexpr int M2_KS1 = 12;          // k-steps of fc1 (C padded to 384)
constexpr int M2_KS2 = M2_HS / 32;  // k-steps of fc2
constexpr int M2_NT2 = 3;           // output tiles per wave in fc2: 8 waves x 3 x 16 = 384 >= C
constexpr int M2_KP = M2_KS1 * 32;
constexpr int M2_LD = M2_KP + 8;    // A-tile row stride (elements): rows shift by one 16-byte bank slot
constexpr int M2_LDH = M2_HS + 8;
constexpr int M2_MAXCL = 4;         // clips one row tile may touch

__device__ __forceinline__ float gelu_fast2(float x) {      // erf by Abramowitz-Stegun 7.1.26, |err| <= 1.5e-7
  const float z = fabsf(x) * 0.70710678118654752440f;
  const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, z, 1.0f));
  const float poly = t * fmaf(t, fmaf(t, fmaf(t, fmaf(t, 1.061405429f, -1.453152027f), 1.421413741f), -0.284496736f), 0.254829592f);
  const float e = 1.0f - poly * __builtin_amdgcn_exp2f(-1.44269504088896341f * z * z);
  return 0.5f * x * (1.0f + copysignf(e, x));
}

// STAMP: diagnostic build only (tools/stamp_sgp_mlp2.py): wave 0 of every workgroup leaves s_memtime / s_memrealtime
// stamps of its phases in `stamps` [workgroup][8]; never instantiated by the product path
#define M2_STAMP(i)                                                                                     \
  if constexpr (STAMP) {                                                                                \
    if (tid == 0) stamps[(long)blockIdx.x * 8 + (i)] = __builtin_amdgcn_s_memtime(); \
  }

// sum over each row of 16 lanes, every lane gets it: four DPP moves (quad swaps, half-row mirror, row mirror) instead of
// four ds_bpermute round trips through the LDS crossbar (~100 cycles each when nothing else covers them)
__device__ __forceinline__ float row16_sum(float v) {
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));   // lane ^ 1
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true));   // lane ^ 2
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xF, 0xF, true));  // 7 - lane in the half
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x140, 0xF, 0xF, true));  // 15 - lane in the row
  return v;
}

template <int MT, bool STAMP = false>
__global__ __launch_bounds__(M2_NW * 64, 2) void sgp_mlp2_kernel(
    const bf16_t* __restrict__ y, int R, int T_len, int C, int G, const float* __restrict__ gn_w,
    const float* __restrict__ gn_b, float eps, const bf16_t* __restrict__ W1, const float* __restrict__ b1,
    const bf16_t* __restrict__ W2, float* __restrict__ partial, const float* __restrict__ chsum, int nslice,
    unsigned long long* __restrict__ stamps = nullptr, int dbg = 0) {
  constexpr int ROWS = 16 * MT;
  constexpr int NTHR = M2_NW * 64;
  extern __shared__ __attribute__((aligned(16))) unsigned char smraw[];
  bf16_t* At = reinterpret_cast<bf16_t*>(smraw);                    // [ROWS][M2_LD]  GN(y) rows, bf16, K pad zero
  bf16_t* Ht = At + ROWS * M2_LD;                                   // [ROWS][M2_LDH] GELU(fc1) of this slice
  float* gstat = reinterpret_cast<float*>(Ht + ROWS * M2_LDH);      // [M2_MAXCL][32][2] (mean, rstd)
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, lr = lane & 15, lq = lane >> 4;
  // workgroup -> (slice, row tile), row tiles fastest: consecutive ids (dealt round-robin over the 8 XCDs) give every XCD
  // the same number of workgroups.  (Pinning each slice's row tiles to one XCD -- its 192 KB of weights then leave the
  // memory side once -- was measured: -3 % cycles per workgroup at 64-row tiles, but 12 slices over 8 XCDs put 34 workgroups
  // on four XCDs of 32 CUs at 48-row tiles and the launch took 20 % longer.)
  const int nrt = (R + ROWS - 1) / ROWS;
  const int s = blockIdx.x / nrt;
  if (s >= nslice) return;
  const int r0 = (blockIdx.x - s * nrt) * ROWS;
  const int c_lo = r0 / T_len, c_hi = min(R - 1, r0 + ROWS - 1) / T_len;
  const int cg = C / G;
  const int nck = C / 8;
  const int nto = C / 16;
  if constexpr (STAMP) {
    if (tid == 0) stamps[(long)blockIdx.x * 8 + 7] = __builtin_amdgcn_s_memrealtime();
  }
  M2_STAMP(0)

  // ---- loads, oldest-needed first (vmcnt retires in issue order): per-channel sums for the GroupNorm statistics, the
  // GroupNorm affine of this thread's channel chunk, the row tile, the fc1 weight tile.  (The fc2 tiles are requested
  // behind the A-tile staging: their latency then hides behind fc1 and their registers are free for fc1's LDS reads.)
  // (1) GroupNorm partial sums: 16 lanes per group walk the group's channels, <= M2_MAXCL clips per tile
  constexpr int MAXG = 2;                                           // channels per lane and group: cg <= 32
  float cs[M2_MAXCL][MAXG][2];
  const int ncl = c_hi - c_lo + 1;
  const bool gthr = tid < G * 16;
  const int gg = min(gthr ? tid >> 4 : 0, G - 1), gj = tid & 15;
#pragma unroll
  for (int ci = 0; ci < M2_MAXCL; ++ci)
#pragma unroll
    for (int u = 0; u < MAXG; ++u) {
      const int cl = min(gj + 16 * u, cg - 1);
      const f32x2 v = *reinterpret_cast<const f32x2*>(chsum + ((long)min(c_lo + ci, c_hi) * C + gg * cg + cl) * 2);
      cs[ci][u][0] = v[0];
      cs[ci][u][1] = v[1];
    }
  // (2) staging roles: thread = (channel chunk ck, row lane rl); rows r0 + rl, + NRL, ...: the chunk's GroupNorm affine is
  // loaded once, the (clip, group) statistics are re-read from LDS only when the clip changes
  const int NRL = NTHR / nck;                                       // row lanes (>= 10 for C <= 384)
  const int ck = tid % nck, rl = tid / nck;
  const bool sthr = rl < NRL;
  constexpr int MAXIT = (ROWS + 9) / 10;
  f32x4 gw[2], gb[2];
  gw[0] = *reinterpret_cast<const f32x4*>(gn_w + ck * 8);
  gw[1] = *reinterpret_cast<const f32x4*>(gn_w + ck * 8 + 4);
  gb[0] = *reinterpret_cast<const f32x4*>(gn_b + ck * 8);
  gb[1] = *reinterpret_cast<const f32x4*>(gn_b + ck * 8 + 4);
  bf16x8 yv[MAXIT];
  const long rbase = (STAMP && (dbg & 2)) ? 0 : r0;
#pragma unroll
  for (int it = 0; it < MAXIT; ++it) {
    const int row = min(rl + it * NRL, ROWS - 1);
    const long r = min(rbase + row, (long)R - 1);
    yv[it] = *reinterpret_cast<const bf16x8*>(y + r * C + ck * 8);
  }
  // (3) fc1 weight tile of this wave (slice s), pre-packed in MFMA A-operand fragment order: one wave-load = 1 KB
  const int sw = (STAMP && (dbg & 1)) ? 0 : s;
  bf16x8 w1[M2_KS1];
  {
    const bf16_t* p1 = W1 + (((long)sw * M2_NW + wid) * M2_KS1 * 64 + lane) * 8;
#pragma unroll
    for (int ks = 0; ks < M2_KS1; ++ks) w1[ks] = *reinterpret_cast<const bf16x8*>(p1 + ks * 512);
  }
  const f32x4 bias1 = *reinterpret_cast<const f32x4*>(b1 + s * M2_HS + wid * 16 + lq * 4);
  TD_ISSUE_FENCE();
  M2_STAMP(1)

  // ---- GroupNorm statistics of the clips this tile touches (fixed-order butterfly over the group's channels)
#pragma unroll
  for (int ci = 0; ci < M2_MAXCL; ++ci) {
    float a = 0.f, bq = 0.f;
#pragma unroll
    for (int u = 0; u < MAXG; ++u)
      if (gj + 16 * u < cg) {
        a += cs[ci][u][0];
        bq += cs[ci][u][1];
      }
    a = row16_sum(a);
    bq = row16_sum(bq);
    if (gthr && gj == 0 && ci < ncl) {
      const float n = (float)(cg * T_len);
      const float mean = a / n;
      const float var = fmaxf(bq / n - mean * mean, 0.f);
      gstat[(ci * 32 + gg) * 2] = mean;
      gstat[(ci * 32 + gg) * 2 + 1] = 1.0f / sqrtf(var + eps);
    }
  }
  __syncthreads();
  M2_STAMP(2)

  // ---- A = GN(y rows) as bf16 into LDS (K pad columns zero): GN(y) = y * sc + sh, sc = rstd_g * w_c, sh = b_c - mean_g * sc
  {
    const IDiv dT(T_len), dcg(cg);
    const int g0 = dcg.div(ck * 8);
    const int split = (g0 + 1) * cg - ck * 8;                       // chunk elements >= split belong to group g0 + 1
    const int g1 = min(g0 + 1, G - 1);
    float sc[8], sh[8];
    int cur = -1;
#pragma unroll
    for (int it = 0; it < MAXIT; ++it) {
      const int row = rl + it * NRL;
      if (sthr && row < ROWS) {
        const long r = (long)r0 + row;
        bf16x8 o;
        if (r < R) {
          const int ci = dT.div((int)r) - c_lo;
          if (ci != cur) {
            cur = ci;
            const f32x2 st0 = *reinterpret_cast<const f32x2*>(gstat + (ci * 32 + g0) * 2);
            const f32x2 st1 = *reinterpret_cast<const f32x2*>(gstat + (ci * 32 + g1) * 2);
#pragma unroll
            for (int e = 0; e < 8; ++e) {                           // groups hold >= 8 channels: a chunk spans at most two
              const bool hi = e >= split;
              const float mean = hi ? st1[0] : st0[0], rstd = hi ? st1[1] : st0[1];
              sc[e] = rstd * gw[e >> 2][e & 3];
              sh[e] = fmaf(-mean, sc[e], gb[e >> 2][e & 3]);
            }
          }
#pragma unroll
          for (int e = 0; e < 8; ++e) o[e] = (bf16_t)fmaf((float)yv[it][e], sc[e], sh[e]);
        } else {
#pragma unroll
          for (int e = 0; e < 8; ++e) o[e] = (bf16_t)0.f;
        }
        *reinterpret_cast<bf16x8*>(At + row * M2_LD + ck * 8) = o;
      }
    }
    const int padc = (M2_LD - C) / 8;                               // 16-byte chunks of padding per row
    for (int i = tid; i < ROWS * padc; i += NTHR) {
      const int row = i / padc, pk = i - row * padc;
      bf16x8 zz;
#pragma unroll
      for (int e = 0; e < 8; ++e) zz[e] = (bf16_t)0.f;
      *reinterpret_cast<bf16x8*>(At + row * M2_LD + C + pk * 8) = zz;
    }
  }
  // (4) fc2 weight tiles of this wave (3 output tiles x 4 k-steps): requested now, needed after fc1
  bf16x8 w2[M2_NT2][M2_KS2];
#pragma unroll
  for (int nt = 0; nt < M2_NT2; ++nt) {
    const int ot = min(wid * M2_NT2 + nt, nto - 1);
    const bf16_t* p2 = W2 + ((((long)sw * nto + ot) * M2_KS2) * 64 + lane) * 8;
#pragma unroll
    for (int ks = 0; ks < M2_KS2; ++ks) w2[nt][ks] = *reinterpret_cast<const bf16x8*>(p2 + ks * 512);
  }
  __syncthreads();
  M2_STAMP(3)

  // ---- fc1: this wave's 16 hidden units x ROWS rows (weights = MFMA A operand, activation rows = B operand: lane l
  // ends up with units 4*(l>>4) .. +3 of activation row l&15).  LDS reads go out a batch of k-steps ahead of the MFMAs
  // that consume them: at 2 waves per SIMD nothing else hides their latency.
  f32x4 acc1[MT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) acc1[mt] = f32x4{0.f, 0.f, 0.f, 0.f};
  constexpr int KB = MT <= 2 ? 4 : 2;                               // k-steps per batch: KB * MT fragments per buffer
  constexpr int NB = M2_KS1 / KB;
  bf16x8 xf[2][KB][MT];
  auto read_batch = [&](bf16x8 (&dst)[KB][MT], int kb) {
#pragma unroll
    for (int k = 0; k < KB; ++k)
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
        dst[k][mt] = *reinterpret_cast<const bf16x8*>(At + (mt * 16 + lr) * M2_LD + (kb * KB + k) * 32 + lq * 8);
  };
  read_batch(xf[0], 0);
#pragma unroll
  for (int kb = 0; kb < NB; ++kb) {
    if (kb + 1 < NB) read_batch(xf[(kb + 1) & 1], kb + 1);
    __builtin_amdgcn_sched_barrier(0);                              // the next batch's reads stay ahead of this batch's MFMAs
#pragma unroll
    for (int k = 0; k < KB; ++k)
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
        acc1[mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w1[kb * KB + k], xf[kb & 1][k][mt], acc1[mt], 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
  }
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
    bf16x4 o;
#pragma unroll
    for (int j = 0; j < 4; ++j) o[j] = (bf16_t)gelu_fast2(acc1[mt][j] + bias1[j]);
    *reinterpret_cast<bf16x4*>(Ht + (mt * 16 + lr) * M2_LDH + wid * 16 + lq * 4) = o;
  }
  __syncthreads();
  M2_STAMP(4)

  // ---- fc2 partial of this slice: out features of this wave's tiles += W2[:, slice] . Hs
  f32x4 acc2[M2_NT2][MT];
#pragma unroll
  for (int nt = 0; nt < M2_NT2; ++nt)
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) acc2[nt][mt] = f32x4{0.f, 0.f, 0.f, 0.f};
  {
    bf16x8 hf[M2_KS2][MT];
#pragma unroll
    for (int ks = 0; ks < M2_KS2; ++ks)
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
        hf[ks][mt] = *reinterpret_cast<const bf16x8*>(Ht + (mt * 16 + lr) * M2_LDH + ks * 32 + lq * 8);
#pragma unroll
    for (int ks = 0; ks < M2_KS2; ++ks)
#pragma unroll
      for (int nt = 0; nt < M2_NT2; ++nt)
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
          acc2[nt][mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w2[nt][ks], hf[ks][mt], acc2[nt][mt], 0, 0, 0);
  }
#pragma unroll
  for (int nt = 0; nt < M2_NT2; ++nt) {
    const int n0 = (wid * M2_NT2 + nt) * 16 + lq * 4;
    if (n0 < C) {
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
        const long r = (long)r0 + mt * 16 + lr;
        if (r < R) *reinterpret_cast<f32x4*>(partial + ((long)s * R + r) * C + n0) = acc2[nt][mt];
      }
    }
    if (nt == 0) { M2_STAMP(5) }
  }
  if constexpr (STAMP) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    M2_STAMP(6)
  }
}

// out[r][:] = bf16(sum_s partial[s][r][:] (fixed order) + b2 + y[r][:]); rowstat[r] = (mean, rstd) over C of the stored
// (rounded) row -- the LayerNorm statistics the next block's front kernel needs (modules.py:353-357: biased variance,
// eps inside the sqrt).  One wave per row, 16-byte accesses.
__global__ __launch_bounds__(256) void sgp_fold_rows_kernel(const float* __restrict__ partial, int S, int R, int C,
                                                            const float* __restrict__ b2, const bf16_t* __restrict__ y,
                                                            bf16_t* __restrict__ out, float* __restrict__ rowstat,
                                                            float eps) {
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int r = blockIdx.x * 4 + wid;
  if (r >= R) return;
  const long RC = (long)R * C;
  float s1 = 0.f, s2 = 0.f;
  for (int c0 = lane * 4; c0 < C; c0 += 256) {
    const bf16x4 yr = *reinterpret_cast<const bf16x4*>(y + (long)r * C + c0);
    const f32x4 bb = *reinterpret_cast<const f32x4*>(b2 + c0);
    f32x4 a = {(float)yr[0] + bb[0], (float)yr[1] + bb[1], (float)yr[2] + bb[2], (float)yr[3] + bb[3]};
    const float* p = partial + (long)r * C + c0;
    f32x4 pv[16];
#pragma unroll
    for (int s = 0; s < 16; ++s) pv[s] = *reinterpret_cast<const f32x4*>(p + (long)min(s, S - 1) * RC);
#pragma unroll
    for (int s = 0; s < 16; ++s)
      if (s < S) { a[0] += pv[s][0]; a[1] += pv[s][1]; a[2] += pv[s][2]; a[3] += pv[s][3]; }
    for (int s = 16; s < S; ++s) {
      const f32x4 q = *reinterpret_cast<const f32x4*>(p + (long)s * RC);
      a[0] += q[0]; a[1] += q[1]; a[2] += q[2]; a[3] += q[3];
    }
    bf16x4 o = {(bf16_t)a[0], (bf16_t)a[1], (bf16_t)a[2], (bf16_t)a[3]};
    *reinterpret_cast<bf16x4*>(out + (long)r * C + c0) = o;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float v = (float)o[e];
      s1 += v;
      s2 = fmaf(v, v, s2);
    }
  }
  if (rowstat) {
    s1 = wave_sum(s1);
    s2 = wave_sum(s2);
    if (lane == 0) {
      const float m = s1 / (float)C;
      const float var = fmaxf(s2 / (float)C - m * m, 0.f);
      rowstat[(long)r * 2] = m;
      rowstat[(long)r * 2 + 1] = 1.0f / sqrtf(var + eps);
    }
  }
}

// The same fold with the AdaptiveMaxPool1d that follows an encoder block (modules.py:64, 75-77) folded in: one workgroup
// per POOLED row i of a clip, one wave per block-output row of its window [floor(i T/O), ceil((i+1) T/O)) (<= 3 rows).
// Wave k folds row lo + k exactly like sgp_fold_rows_kernel, stores it with its LayerNorm statistics if this window
// owns it (row t belongs to the window whose start is the last one <= t: overlapping windows of odd lengths fold a shared
// row twice and store it once), and leaves the stored (rounded) values in LDS; wave 0 then takes the element-wise maximum
// as the pooled row and its statistics.  Removes the max-pool launch and its pass over the map.
__global__ __launch_bounds__(192) void sgp_fold_rows_pool_kernel(const float* __restrict__ partial, int S, int B, int T_in,
                                                                 int T_out, int C, const float* __restrict__ b2,
                                                                 const bf16_t* __restrict__ y, bf16_t* __restrict__ out,
                                                                 float* __restrict__ rowstat, bf16_t* __restrict__ pooled,
                                                                 float* __restrict__ rowstat_p, float eps) {
  extern __shared__ __attribute__((aligned(16))) unsigned char fsm[];
  bf16_t* rows = reinterpret_cast<bf16_t*>(fsm);                    // [3][C] stored values of the window's rows
  const int lane = threadIdx.x & 63, k = threadIdx.x >> 6;
  const int b = blockIdx.x / T_out, i = blockIdx.x - b * T_out;
  const int lo = (int)(((long)i * T_in) / T_out);
  const int hi = (int)((((long)(i + 1)) * T_in + T_out - 1) / T_out);
  const int lo_next = i + 1 < T_out ? (int)(((long)(i + 1) * T_in) / T_out) : T_in;
  const int nwin = hi - lo;
  const long RC = (long)B * T_in * C;
  if (k < nwin) {
    const long r = (long)b * T_in + lo + k;
    const bool own = lo + k < lo_next;
    float s1 = 0.f, s2 = 0.f;
    for (int c0 = lane * 4; c0 < C; c0 += 256) {
      const bf16x4 yr = *reinterpret_cast<const bf16x4*>(y + r * C + c0);
      const f32x4 bb = *reinterpret_cast<const f32x4*>(b2 + c0);
      f32x4 a = {(float)yr[0] + bb[0], (float)yr[1] + bb[1], (float)yr[2] + bb[2], (float)yr[3] + bb[3]};
      const float* p = partial + r * C + c0;
      f32x4 pv[16];
#pragma unroll
      for (int s = 0; s < 16; ++s) pv[s] = *reinterpret_cast<const f32x4*>(p + (long)min(s, S - 1) * RC);
#pragma unroll
      for (int s = 0; s < 16; ++s)
        if (s < S) { a[0] += pv[s][0]; a[1] += pv[s][1]; a[2] += pv[s][2]; a[3] += pv[s][3]; }
      for (int s = 16; s < S; ++s) {
        const f32x4 q = *reinterpret_cast<const f32x4*>(p + (long)s * RC);
        a[0] += q[0]; a[1] += q[1]; a[2] += q[2]; a[3] += q[3];
      }
      const bf16x4 o = {(bf16_t)a[0], (bf16_t)a[1], (bf16_t)a[2], (bf16_t)a[3]};
      if (own) *reinterpret_cast<bf16x4*>(out + r * C + c0) = o;
      *reinterpret_cast<bf16x4*>(rows + k * C + c0) = o;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float v = (float)o[e];
        s1 += v;
        s2 = fmaf(v, v, s2);
      }
    }
    if (own && rowstat) {
      s1 = wave_sum(s1);
      s2 = wave_sum(s2);
      if (lane == 0) {
        const float m = s1 / (float)C;
        rowstat[r * 2] = m;
        rowstat[r * 2 + 1] = 1.0f / sqrtf(fmaxf(s2 / (float)C - m * m, 0.f) + eps);
      }
    }
  }
  __syncthreads();
  if (k == 0) {
    float p1 = 0.f, p2 = 0.f;
    for (int c0 = lane * 4; c0 < C; c0 += 256) {
      bf16x4 m = *reinterpret_cast<const bf16x4*>(rows + c0);
      for (int j = 1; j < nwin; ++j) {
        const bf16x4 v = *reinterpret_cast<const bf16x4*>(rows + j * C + c0);
#pragma unroll
        for (int e = 0; e < 4; ++e) m[e] = (float)v[e] > (float)m[e] ? v[e] : m[e];
      }
      *reinterpret_cast<bf16x4*>(pooled + ((long)b * T_out + i) * C + c0) = m;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float v = (float)m[e];
        p1 += v;
        p2 = fmaf(v, v, p2);
      }
    }
    if (rowstat_p) {
      p1 = wave_sum(p1);
      p2 = wave_sum(p2);
      if (lane == 0) {
        const float m = p1 / (float)C;
        rowstat_p[((long)b * T_out + i) * 2] = m;
        rowstat_p[((long)b * T_out + i) * 2 + 1] = 1.0f / sqrtf(fmaxf(p2 / (float)C - m * m, 0.f) + eps);
      }
    }
  }
}

// out[b][t][c] = bf16(act(sum_s partial[s][b*T + t][c] + bias[c])) for one clip x 16 channels per workgroup, and the
// per-channel sum / sum of squares over T of the stored values -> chsum [B][C][2] (what the GroupNorm of the MLP half
// reads): the fold of the mixer's concat_fc contraction (modules.py:308-309: Conv1d(6C, C, 1) -> GELU).
__global__ __launch_bounds__(256) void sgp_fold_cols_kernel(const float* __restrict__ partial, int S, int T_len, int C,
                                                            long RC, const float* __restrict__ bias, int act,
                                                            bf16_t* __restrict__ out, float* __restrict__ chsum) {
  __shared__ float red[64][16][2];
  const int b = blockIdx.x, c0 = blockIdx.y * 16;
  const int cq = threadIdx.x & 3, tl = threadIdx.x >> 2;
  const int c = min(c0 + cq * 4, C - 4);
  const bool cok = c0 + cq * 4 < C;
  const f32x4 bb = *reinterpret_cast<const f32x4*>(bias + c);
  float s1[4] = {0.f, 0.f, 0.f, 0.f}, s2[4] = {0.f, 0.f, 0.f, 0.f};
  for (int t = tl; t < T_len; t += 64) {
    const long off = ((long)b * T_len + t) * C + c;
    f32x4 pv[16];
#pragma unroll
    for (int s = 0; s < 16; ++s) pv[s] = *reinterpret_cast<const f32x4*>(partial + (long)min(s, S - 1) * RC + off);
    f32x4 a = bb;
#pragma unroll
    for (int s = 0; s < 16; ++s)
      if (s < S) { a[0] += pv[s][0]; a[1] += pv[s][1]; a[2] += pv[s][2]; a[3] += pv[s][3]; }
    for (int s = 16; s < S; ++s) {
      const f32x4 p = *reinterpret_cast<const f32x4*>(partial + (long)s * RC + off);
      a[0] += p[0]; a[1] += p[1]; a[2] += p[2]; a[3] += p[3];
    }
    bf16x4 o;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      float x = a[e];
      if (act == TDEED_ACT_RELU) x = fmaxf(x, 0.f);
      else if (act == TDEED_ACT_GELU) x = gelu_erf(x);
      o[e] = (bf16_t)x;
      const float v = (float)o[e];
      s1[e] += v;
      s2[e] = fmaf(v, v, s2[e]);
    }
    if (cok) *reinterpret_cast<bf16x4*>(out + off) = o;
  }
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    red[tl][cq * 4 + e][0] = s1[e];
    red[tl][cq * 4 + e][1] = s2[e];
  }
  __syncthreads();
  if (threadIdx.x < 32) {
    const int ch = threadIdx.x >> 1, k = threadIdx.x & 1;
    float a = 0.f;
    for (int i = 0; i < 64; ++i) a += red[i][ch][k];
    if (c0 + ch < C) chsum[((long)b * C + c0 + ch) * 2 + k] = a;
  }
}

// 1-D grid: slices x row tiles
int mlp2_grid(int R, int rows, int S) { return S * ((R + rows - 1) / rows); }

size_t mlp2_smem(int rows, int C) {
  (void)C;
  return (size_t)rows * (M2_LD + M2_LDH) * 2 + (size_t)M2_MAXCL * 32 * 2 * sizeof(float);
}

// rows per workgroup: the smallest of 32 / 48 / 64 whose grid (row tiles x slices) still fits the chip's 256 CUs at one
// workgroup each (per-workgroup time grows with the rows: GroupNorm + GELU are vector-ALU work, the A tile LDS traffic),
// as long as a tile then touches at most M2_MAXCL clips
int mlp2_rows(int R, int T, int S) {
  static const int force_rows = getenv("TDEED_SGP_MLP2_ROWS") ? atoi(getenv("TDEED_SGP_MLP2_ROWS")) : 0;
  auto ok = [&](int rows) { return (rows - 1) / T + 2 <= M2_MAXCL; };
  if ((force_rows == 32 || force_rows == 48 || force_rows == 64) && ok(force_rows)) return force_rows;
  for (int rows = 32; rows <= 64; rows += 16)
    if (ok(rows) && cdiv(R, rows) * S <= 256) return rows;
  return ok(64) ? 64 : (ok(48) ? 48 : 32);
}

}  // namespace

// partial: [S][B*T][C] fp32 (tdeed_gemm_splitk_partials' workspace); out (B,T,C) bf16; chsum [B][C][2] fp32
extern "C" int tdeed_sgp_fold_cols(const float* partial, int S, int B, int T, int C, const float* bias, int act, void* out,
                                   float* chsum, void* stream) {
  TD_CHECK(partial && bias && out && chsum, "sgp_fold_cols: null pointer");
  TD_CHECK(S > 0 && B > 0 && T > 0 && C >= 4 && C % 4 == 0 && act >= 0 && act <= 2, "sgp_fold_cols: bad arguments");
  hipLaunchKernelGGL(sgp_fold_cols_kernel, dim3(B, cdiv(C, 16)), dim3(256), 0, (hipStream_t)stream, partial, S, T, C,
                     (long)B * T * C, bias, act, (bf16_t*)out, chsum);
  TD_LAUNCH_CHECK("sgp_fold_cols");
  return TDEED_OK;
}

// diagnostic: the 64-row kernel with phase stamps (stamps: [R/64 * S][8] u64; [0..6] s_memtime at start / loads issued /
// statistics done / A tile staged / fc1 done / first partial store issued / stores landed, [7] s_memrealtime at start)
extern "C" int tdeed_sgp_mlp2_stamped(const void* y, int R, int T, int C, int G, const float* gn_w, const float* gn_b,
                                      float eps, const void* W1p, const float* b1p, const void* W2p, float* partial,
                                      const float* chsum, unsigned long long* stamps, int rows, int dbg, void* stream) {
  TD_CHECK(y && W1p && W2p && partial && chsum && stamps, "sgp_mlp2_stamped: null pointer");
  TD_CHECK(tdeed_sgp_mlp2_fits(R, T, C, G), "sgp_mlp2_stamped: geometry not served");
  const int S = (4 * C + M2_HS - 1) / M2_HS;
  hipError_t e = hipFuncSetAttribute((const void*)sgp_mlp2_kernel<4, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  if (e == hipSuccess) e = hipFuncSetAttribute((const void*)sgp_mlp2_kernel<2, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  if (e != hipSuccess) { tdeed_set_error("sgp_mlp2_stamped: %s", hipGetErrorString(e)); return TDEED_ERR_RUNTIME; }
  if (rows == 64)
    hipLaunchKernelGGL((sgp_mlp2_kernel<4, true>), dim3(mlp2_grid(R, 64, S)), dim3(M2_NW * 64), mlp2_smem(64, C), (hipStream_t)stream,
                       (const bf16_t*)y, R, T, C, G, gn_w, gn_b, eps, (const bf16_t*)W1p, b1p, (const bf16_t*)W2p, partial, chsum, S, stamps, dbg);
  else
    hipLaunchKernelGGL((sgp_mlp2_kernel<2, true>), dim3(mlp2_grid(R, 32, S)), dim3(M2_NW * 64), mlp2_smem(32, C), (hipStream_t)stream,
                       (const bf16_t*)y, R, T, C, G, gn_w, gn_b, eps, (const bf16_t*)W1p, b1p, (const bf16_t*)W2p, partial, chsum, S, stamps, dbg);
  TD_LAUNCH_CHECK("sgp_mlp2_stamped");
  return TDEED_OK;
}

// hidden slices of the narrow form: ceil(4C / 128)
extern "C" int tdeed_sgp_mlp2_slices(int C) { return (4 * C + M2_HS - 1) / M2_HS; }

// 1 when sgp_mlp2 serves the geometry: bf16, C a multiple of 16 with C <= 384, G groups of <= 32 channels, producer
// supplies per-channel sums (chsum)
extern "C" int tdeed_sgp_mlp2_fits(int R, int T, int C, int G) {
  if (C % 16 != 0 || C > M2_KP || C < 64 || G <= 0 || G > 32 || C % G != 0 || C / G > 32 || C / G < 8 || T <= 0 || R % T != 0)
    return 0;
  if (tdeed_sgp_mlp2_slices(C) > 16) return 0;
  if ((32 - 1) / T + 2 > M2_MAXCL) return 0;                         // even a 32-row tile would touch too many clips
  return 1;
}

// y (R,C) bf16 rows of whole clips of T rows; chsum [R/T][C][2] per-channel (sum, sum of squares) over each clip's rows of
// y (written by tdeed_sgp_front_fwd); W1p / b1p / W2p: engine.pack_mlp2_frags; partial: fp32 scratch
// [tdeed_sgp_mlp2_slices(C)][R][C]; rowstat: optional [R][2] output (LayerNorm mean, rstd of every output row).
extern "C" int tdeed_sgp_mlp2_fwd(const void* y, int R, int T, int C, int G, const float* gn_w, const float* gn_b, float eps,
                                  const void* W1p, const float* b1p, const void* W2p, const float* b2, void* out,
                                  float* partial, const float* chsum, float* rowstat, float ln_eps, int T_pool,
                                  void* pooled, float* rowstat_pool, void* stream) {
  TD_CHECK(y && gn_w && gn_b && W1p && b1p && W2p && b2 && out && partial && chsum, "sgp_mlp2: null pointer");
  TD_CHECK(tdeed_sgp_mlp2_fits(R, T, C, G), "sgp_mlp2: geometry R=%d T=%d C=%d G=%d not served", R, T, C, G);
  const int S = tdeed_sgp_mlp2_slices(C);
  hipStream_t st = (hipStream_t)stream;
  static TdDevOnce attr_set;
  if (!attr_set.get()) {
    hipError_t e = hipFuncSetAttribute((const void*)sgp_mlp2_kernel<4>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e == hipSuccess) e = hipFuncSetAttribute((const void*)sgp_mlp2_kernel<3>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e == hipSuccess) e = hipFuncSetAttribute((const void*)sgp_mlp2_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) { tdeed_set_error("sgp_mlp2: hipFuncSetAttribute: %s", hipGetErrorString(e)); return TDEED_ERR_RUNTIME; }
    attr_set.set();
  }
  // 64-row tiles while they fill the chip (R / 64 x S workgroups); 32-row tiles for the short pyramid levels
  const int rows = mlp2_rows(R, T, S);
  if (rows == 64)
    hipLaunchKernelGGL((sgp_mlp2_kernel<4>), dim3(mlp2_grid(R, 64, S)), dim3(M2_NW * 64), mlp2_smem(64, C), st, (const bf16_t*)y, R,
                       T, C, G, gn_w, gn_b, eps, (const bf16_t*)W1p, b1p, (const bf16_t*)W2p, partial, chsum, S);
  else if (rows == 48)
    hipLaunchKernelGGL((sgp_mlp2_kernel<3>), dim3(mlp2_grid(R, 48, S)), dim3(M2_NW * 64), mlp2_smem(48, C), st, (const bf16_t*)y, R,
                       T, C, G, gn_w, gn_b, eps, (const bf16_t*)W1p, b1p, (const bf16_t*)W2p, partial, chsum, S);
  else
    hipLaunchKernelGGL((sgp_mlp2_kernel<2>), dim3(mlp2_grid(R, 32, S)), dim3(M2_NW * 64), mlp2_smem(32, C), st, (const bf16_t*)y, R,
                       T, C, G, gn_w, gn_b, eps, (const bf16_t*)W1p, b1p, (const bf16_t*)W2p, partial, chsum, S);
  if (T_pool > 0) {
    // windows of AdaptiveMaxPool1d(T_pool) over T rows hold at most 3 rows for T_pool >= T / 2 (the pyramid halves, rounding up)
    TD_CHECK(pooled && T_pool <= T && 2 * T_pool >= T, "sgp_mlp2: pooled output needs T/2 <= T_pool <= T (T=%d, T_pool=%d)", T, T_pool);
    const int Bn = R / T;
    hipLaunchKernelGGL(sgp_fold_rows_pool_kernel, dim3(Bn * T_pool), dim3(192), (size_t)3 * C * sizeof(bf16_t), st, partial, S,
                       Bn, T, T_pool, C, b2, (const bf16_t*)y, (bf16_t*)out, rowstat, (bf16_t*)pooled, rowstat_pool, ln_eps);
  } else
    hipLaunchKernelGGL(sgp_fold_rows_kernel, dim3(cdiv(R, 4)), dim3(256), 0, st, partial, S, R, C, b2, (const bf16_t*)y,
                       (bf16_t*)out, rowstat, ln_eps);
  TD_LAUNCH_CHECK("sgp_mlp2");
  return TDEED_OK;
}
