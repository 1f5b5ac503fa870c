// SE excitation (timm SqueezeExcite fc1 / ReLU / fc2 / sigmoid of a RegNetY bottleneck) computed INSIDE the
// conv3 contraction that consumes the gates: a workgroup derives the gates of the (few) frames its rows belong to from the
// squeeze sums conv2 left ([N][n_parts][C] fp32) and keeps them in LDS, so no se_gate launch sits between conv2 and conv3.
// Arithmetic = se_gate_mfma_kernel (conv.hip): the frames are MFMA columns, the weights arrive as A-operand fragments
// (engine.pack_se_mfma), the fp32 means / hidden units are split hi + lo into two bf16 MFMAs.
#pragma once
#include "common.h"

struct SeP {
  const float* pooled;      // [N][n_parts][C] squeeze SUMS; null = no fused excitation
  int n_parts;
  float inv_cnt;            // 1 / (Ho * Wo)
  int C, R;
  const bf16x8* w1f;        // [ceil(R/16)][ceil(C/32)][64]
  const float* b1;
  const bf16x8* w2f;        // [ceil(C/16)][ceil(R/32)][64]
  const float* b2;
  float* gate_out;          // optional [N][C]: the gates as a tensor too (tests / taps); every workgroup that computes a
                            // frame's gates writes the same values
};

// bytes of LDS scratch: means hi / lo [16][PS1] and hidden hi / lo [16][PS2] in bf16 (+8 elements per row spread the banks)
__host__ __device__ inline int se_excite_scratch_bytes(int C, int R) {
  const int KS1 = (C + 31) >> 5, KS2 = (R + 31) >> 5;
  return 2 * 16 * (KS1 * 32 + 8) * 2 + 2 * 16 * (KS2 * 32 + 8) * 2;
}

// gates of frames [f_first, f_first + nf) (nf <= 16, frames beyond f_last clamp to it) -> gtab[f][ldg] (LDS, fp32).
// All NW * 64 threads of the workgroup call it; it ends with a barrier (gtab and scratch are then free to read / reuse).
// se.pooled may point into LDS (generic address).
// stamps (diagnostic, tools/bench_bneck.py): thread 0 leaves clock64() behind the means / hidden / gate stages in stamps[0..2]
// PRE: the fc1 fragments of this wave's first hidden tile were requested by the caller (*w1pre), earlier than this call.
// The barriers fence LDS only: the weight fragments requested here stay in flight across them (a __syncthreads() drains
// vmcnt, i.e. waited for the fc2 fragments two stages before they are used).
#define SE_LDS_BARRIER()                                               \
  do {                                                                 \
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");    \
    __builtin_amdgcn_s_barrier();                                      \
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");    \
  } while (0)
template <int KS1M, int KS2M, bool EARLY = true, int NW = 4, bool PRE = false>
__device__ __forceinline__ void se_excite_lds(const SeP& se, long f_first, int nf, long f_last, float* gtab, int ldg,
                                              unsigned char* scratch, long long* stamps = nullptr,
                                              const bf16x8 (*w1pre)[KS1M] = nullptr) {
  const int C = se.C, R = se.R;
  const int KS1 = (C + 31) >> 5, RT = (R + 15) >> 4, CT = (C + 15) >> 4, KS2 = (R + 31) >> 5;
  const int PS1 = KS1 * 32 + 8, PS2 = KS2 * 32 + 8;
  bf16_t* Phi = reinterpret_cast<bf16_t*>(scratch);             // [16][PS1]
  bf16_t* Plo = Phi + 16 * PS1;
  bf16_t* Hhi = Plo + 16 * PS1;                                 // [16][PS2]
  bf16_t* Hlo = Hhi + 16 * PS2;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, pl = lane & 15, q = lane >> 4;
  // fc1 fragments of this wave's first hidden tile: requested before anything else
  bf16x8 w1r[KS1M];
  if constexpr (PRE) {
#pragma unroll
    for (int ks = 0; ks < KS1M; ++ks) w1r[ks] = (*w1pre)[ks];
  } else if constexpr (EARLY) {
    const int tc = min(wv, RT - 1);
#pragma unroll
    for (int ks = 0; ks < KS1M; ++ks) w1r[ks] = se.w1f[((long)tc * KS1 + min(ks, KS1 - 1)) * 64 + lane];
  }
  // fc2 fragments of ALL this wave's channel tiles, requested now as well (EARLY): the gate phase otherwise walks its tiles
  // through one L2 round trip each
  constexpr int MAXT2 = (24 + NW - 1) / NW;             // C <= 384: at most 24 channel tiles
  bf16x8 w2r[EARLY ? MAXT2 : 1][KS2M];
  if constexpr (EARLY) {
#pragma unroll
    for (int j = 0; j < MAXT2; ++j) {
      const int tc = min(wv + j * NW, CT - 1);
#pragma unroll
      for (int ks = 0; ks < KS2M; ++ks) w2r[j][ks] = se.w2f[((long)tc * KS2 + min(ks, KS2 - 1)) * 64 + lane];
    }
  }
  // ... and the biases of this lane's rows (a load behind the MFMAs would be one more exposed round trip per tile)
  constexpr int MAXT1 = (6 + NW - 1) / NW;               // R <= 96: at most 6 hidden tiles
  float b1v[MAXT1][4], b2v[MAXT2][4];
#pragma unroll
  for (int j = 0; j < MAXT1; ++j)
#pragma unroll
    for (int e = 0; e < 4; ++e) b1v[j][e] = se.b1[min((wv + j * NW) * 16 + 4 * q + e, R - 1)];
#pragma unroll
  for (int j = 0; j < MAXT2; ++j)
#pragma unroll
    for (int e = 0; e < 4; ++e) b2v[j][e] = se.b2[min((wv + j * NW) * 16 + 4 * q + e, C - 1)];
  // ---- squeeze sums -> means -> hi / lo (pad columns and unused frame columns are zero)
  const int c4n = KS1 * 8;
  for (int i = tid; i < 16 * c4n; i += NW * 64) {
    const int f = i / c4n, c = (i - f * c4n) * 4;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    if (f < nf && c < C) {
      const float* src = se.pooled + (min(f_first + f, f_last) * se.n_parts) * (long)C + c;
      for (int p = 0; p < se.n_parts; ++p) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(src + (long)p * C);
        acc[0] += v[0]; acc[1] += v[1]; acc[2] += v[2]; acc[3] += v[3];
      }
    }
    bf16x4 hi, lo;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float m = acc[e] * se.inv_cnt;
      hi[e] = (bf16_t)m;
      lo[e] = (bf16_t)(m - (float)hi[e]);
    }
    TD_LDS_CHECK((f * PS1 + c) * 2, 8, 16 * PS1 * 2);
    *reinterpret_cast<bf16x4*>(Phi + f * PS1 + c) = hi;
    *reinterpret_cast<bf16x4*>(Plo + f * PS1 + c) = lo;
  }
  for (int i = tid; i < 2 * 16 * PS2 / 8; i += NW * 64) reinterpret_cast<u32x4*>(Hhi)[i] = (u32x4){0u, 0u, 0u, 0u};
  SE_LDS_BARRIER();
  if (stamps && tid == 0) stamps[0] = clock64();
  // ---- hidden units: tiles wv, wv + NW, ...
#pragma unroll
  for (int j1 = 0; j1 < MAXT1; ++j1) {
    const int tile = wv + j1 * NW;
    if (tile >= RT) break;
    if (!(EARLY || PRE) || tile != wv) {
#pragma unroll
      for (int ks = 0; ks < KS1M; ++ks) w1r[ks] = se.w1f[((long)tile * KS1 + min(ks, KS1 - 1)) * 64 + lane];
    }
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < KS1M; ++ks)
      if (ks < KS1) {
        const bf16x8 bh = *reinterpret_cast<const bf16x8*>(Phi + pl * PS1 + ks * 32 + q * 8);
        const bf16x8 bl = *reinterpret_cast<const bf16x8*>(Plo + pl * PS1 + ks * 32 + q * 8);
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w1r[ks], bh, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w1r[ks], bl, acc, 0, 0, 0);
      }
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int r = tile * 16 + 4 * q + e;
      const float hv = r < R ? fmaxf(acc[e] + b1v[j1][e], 0.f) : 0.f;
      const bf16_t hi = (bf16_t)hv;
      Hhi[pl * PS2 + r] = hi;
      Hlo[pl * PS2 + r] = (bf16_t)(hv - (float)hi);
    }
  }
  SE_LDS_BARRIER();
  if (stamps && tid == 0) stamps[1] = clock64();
  // ---- gates: channel tiles wv, wv + NW, ...
#pragma unroll
  for (int j = 0; j < MAXT2; ++j) {
    const int tile = wv + j * NW;
    if (tile >= CT) break;
    bf16x8 wt[KS2M];
#pragma unroll
    for (int ks = 0; ks < KS2M; ++ks) {
      if constexpr (EARLY) wt[ks] = w2r[j][ks];
      else wt[ks] = se.w2f[((long)tile * KS2 + min(ks, KS2 - 1)) * 64 + lane];
    }
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < KS2M; ++ks)
      if (ks < KS2) {
        const bf16x8 bh = *reinterpret_cast<const bf16x8*>(Hhi + pl * PS2 + ks * 32 + q * 8);
        const bf16x8 bl = *reinterpret_cast<const bf16x8*>(Hlo + pl * PS2 + ks * 32 + q * 8);
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wt[ks], bh, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wt[ks], bl, acc, 0, 0, 0);
      }
    const int c0 = tile * 16 + 4 * q;
    if (c0 < C && pl < nf) {
      f32x4 g;
#pragma unroll
      for (int e = 0; e < 4; ++e) g[e] = sigmoid_fast_(acc[e] + b2v[j][e]);
      TD_DEV_ASSERT(pl < 16 && c0 + 4 <= ldg);
      *reinterpret_cast<f32x4*>(gtab + pl * ldg + c0) = g;
      if (se.gate_out && f_first + pl <= f_last) *reinterpret_cast<f32x4*>(se.gate_out + (f_first + pl) * (long)C + c0) = g;
    }
  }
  SE_LDS_BARRIER();
  if (stamps && tid == 0) stamps[2] = clock64();
}
