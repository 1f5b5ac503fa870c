// The dense contractions of the SGP encoder-decoder (reference: /root/reference/model/modules.py:134-138 mlp =
// Conv1d(C,4C,1) -> GELU -> Conv1d(4C,C,1); 186, 316 out = y + mlp(gn(y)); 245-246, 307-308 concat_fc + GELU) as ONE
// pipelined small-M kernel with the surrounding element-wise work in its prologue / epilogue:
//
//   MODE 0  H   = GELU(GroupNorm16(y) . W1^T + b1)             bf16 [R][4C]      (GroupNorm statistics from per-channel sums)
//   MODE 1  out = y + H . W2^T + b2                            + per-row LayerNorm partial sums, + AdaptiveMaxPool1d (T = 2 T')
//   MODE 2  mo  = GELU(cat . Wc^T + bc)                        + per-channel GroupNorm partial sums
//
// Why this form (round 5).  The stage runs a few hundred rows through ~19 MB of weights: every launch is bound by the bytes
// ONE CU must pull through its own load path (~64 B/clk) and by the memory round trips in front of its first MFMA, not by
// HBM or MFMA rates.  Rounds 2/3 split the hidden dimension over workgroups and folded fp32 partials [S][R][C] in a second
// launch (sgp_mlp2 + sgp_fold_*): 390 MB of measured traffic per forward for 25 MB of algorithmic bytes, and at C = 768 the
// partials (118 MB per MLP) ruled the form out altogether.  Here every output element is produced by ONE workgroup over the
// full K: no partials, no fold launch; the hidden tensor H (2.4 MB at cfg2) is the only intermediate that touches memory.
//
//   * workgroup = 4 waves; tile = 16 MT rows of ONE clip (row tiles never straddle clips: one set of GroupNorm statistics,
//     one pooled-row parity, per-clip channel sums) x 64 NT output features; wave w owns NT feature tiles of 16;
//   * weights are the MFMA A operand, pre-packed in fragment order [feature tile][k-step][lane][8] and read straight from
//     L2 into registers (each wave streams only its own tiles: no LDS traffic, 1 KB per wave instruction);
//   * activation rows are the B operand: 128-column chunks staged global -> registers (-> GroupNorm affine) -> LDS, read back
//     as 16-byte fragments; LDS row stride 288 B = 2 slots mod 16: the 16 rows x 4 k-quarters of a fragment read fall on 16
//     distinct 16-byte bank slots;
//   * a ring of three chunks: while chunk c is multiplied, chunks c+1 and c+2 (weights AND rows) are in flight -- ~96 KB per
//     workgroup, two workgroups per CU.  The barrier fences LDS only, the global loads stay in flight across it.  K is padded
//     to whole super-iterations of 3 chunks (zero weights), so the steady state has no conditional load; K = C (12 k-steps at
//     C = 368) is exactly one super-iteration: every load of the launch is issued before the first MFMA.
#include "common.h"
#include <stdio.h>
#include <stdlib.h>

namespace {

constexpr int SG_KC = 4;              // k-steps (of 32) per chunk
constexpr int SG_CK = SG_KC * 32;     // columns per chunk
constexpr int SG_NBUF = 3;            // chunks in the ring
constexpr int SG_LDB = SG_CK * 2 + 32;   // LDS row stride in bytes (288): 18 slots = 2 mod 16
constexpr int SG_PPR = SG_CK / 8;     // 16-byte pieces per row and chunk

struct SgpGemmP {
  const void* A; long lda;            // [B*T][K] activations (TA)
  const bf16x8* W; int KSP;           // packed fragments [N/16][KSP][64], KSP = k-steps padded to a multiple of 12
  const float* bias;                  // [N]
  void* out; long ldo;                // [B*T][N]
  int B, T, N, K, NJ, nct, ct_major;
  const float* chsum; int chs_parts;  // MODE 0: [parts][B][K][2] per-channel (sum, sum of squares) over the clip's rows
  const float* gn_w; const float* gn_b; int G; float eps;
  const void* resid; long ldr;        // MODE 1: residual rows (TO)
  float* rowstat_part;                // MODE 1: [nct][B*T][2] (sum, sum of squares) of the stored row over this tile's features
  void* pooled; float* rowstat_pool_part; int T_out;      // MODE 1 with T == 2 T_out: pooled rows [B*T_out][N] + their sums
  float* chs_out;                     // MODE 2: [NJ][B][N][2] per-channel sums of the stored rows of this row tile
  bf16_t* out16;                      // MODE 2 (optional): a bf16 copy of out, the next contraction's operand in wide models
};

#define SG_LDS_BARRIER()                                               \
  do {                                                                 \
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");    \
    __builtin_amdgcn_s_barrier();                                      \
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");    \
  } while (0)

__device__ __forceinline__ float sg_gelu(float x) {      // erf by Abramowitz-Stegun 7.1.26, |err| <= 1.5e-7
  const float z = fabsf(x) * 0.70710678118654752440f;
  const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, z, 1.0f));
  const float poly = t * fmaf(t, fmaf(t, fmaf(t, fmaf(t, 1.061405429f, -1.453152027f), 1.421413741f), -0.284496736f), 0.254829592f);
  const float e = 1.0f - poly * __builtin_amdgcn_exp2f(-1.44269504088896341f * z * z);
  return 0.5f * x * (1.0f + copysignf(e, x));
}

template <typename T> struct SgRow;          // one 8-column piece of an activation row in registers
template <> struct SgRow<bf16_t> {
  bf16x8 v;
  __device__ __forceinline__ void load(const bf16_t* p) { v = *reinterpret_cast<const bf16x8*>(p); }
  __device__ __forceinline__ float get(int e) const { return (float)v[e]; }
};
template <> struct SgRow<float> {
  f32x4 a, b;
  __device__ __forceinline__ void load(const float* p) {
    a = *reinterpret_cast<const f32x4*>(p);
    b = *reinterpret_cast<const f32x4*>(p + 4);
  }
  __device__ __forceinline__ float get(int e) const { return e < 4 ? a[e] : b[e - 4]; }
};

template <typename T> struct SgOut;          // 4 consecutive features of one row
template <> struct SgOut<float> {
  static __device__ __forceinline__ f32x4 load(const float* p) { return *reinterpret_cast<const f32x4*>(p); }
  static __device__ __forceinline__ float rnd(float v) { return v; }
  static __device__ __forceinline__ void store(float* p, const f32x4& v) { *reinterpret_cast<f32x4*>(p) = v; }
};
template <> struct SgOut<bf16_t> {
  static __device__ __forceinline__ f32x4 load(const bf16_t* p) {
    const bf16x4 t = *reinterpret_cast<const bf16x4*>(p);
    return f32x4{(float)t[0], (float)t[1], (float)t[2], (float)t[3]};
  }
  static __device__ __forceinline__ float rnd(float v) { return (float)(bf16_t)v; }
  static __device__ __forceinline__ void store(bf16_t* p, const f32x4& v) {
    const bf16x4 t = {(bf16_t)v[0], (bf16_t)v[1], (bf16_t)v[2], (bf16_t)v[3]};
    *reinterpret_cast<bf16x4*>(p) = t;
  }
};

// sum over the four 16-lane rows of a wave (lanes sharing l & 15), every lane gets it
__device__ __forceinline__ float sg_sum_q(float v) {
  v += __shfl_xor(v, 16, 64);
  v += __shfl_xor(v, 32, 64);
  return v;
}
// sum over the 16 lanes of a row (lanes sharing l >> 4)
__device__ __forceinline__ float sg_sum_r(float v) { return td_row16_sum(v); }      // (DPP: the xor butterfly's pairing, same bits)

// the same sum by DPP moves (lane ^ 1, lane ^ 2, the other quad of the half row, the other half row): the pairing of the
// xor butterfly above, so the same bits, without its LDS crossbar round trips
__device__ __forceinline__ float sg_row16_sum(float v) {
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true));
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xF, 0xF, true));
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x140, 0xF, 0xF, true));
  return v;
}

template <int MT, int NT, int MODE, typename TA, typename TO>
__global__ __launch_bounds__(256, 2) void sgp_gemm_kernel(const SgpGemmP p) {
  constexpr int BM = 16 * MT;
  constexpr int NP = BM * SG_PPR / 256;                 // row pieces per thread and chunk (MT = 1: 1, 2: 2, 4: 4)
  static_assert(BM * SG_PPR % 256 == 0, "tile rows");
  extern __shared__ __attribute__((aligned(16))) unsigned char sg_smem[];
  unsigned char* abuf = sg_smem;                                              // [NBUF][BM][SG_LDB]
  float* gtab = reinterpret_cast<float*>(sg_smem + SG_NBUF * BM * SG_LDB);    // MODE 0: [2][Kp] GroupNorm scale / shift
  float* red = gtab + (MODE == 0 ? 2 * p.KSP * 32 : 0);                       // reduction scratch
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, lr = lane & 15, lq = lane >> 4;
  // ---- which tile: logical ids run along one XCD; ct_major = 1: that XCD keeps a few feature tiles' weights and sees every
  // row tile (MODE 0: the rows are small), 0: it keeps a few row tiles and streams every weight (MODE 1 / 2: K = 4C .. 6C rows)
  const int nrt = p.B * p.NJ;
  const int L = (int)xcd_logical_id(blockIdx.x, (long)nrt * p.nct);
  const int rt = p.ct_major ? L % nrt : L / p.nct;
  const int ct = p.ct_major ? L / nrt : L % p.nct;
  const int b = rt / p.NJ, j = rt - b * p.NJ;
  const int t0 = j * BM;                                 // first row of the tile inside its clip
  const int nrows = min(BM, p.T - t0);
  const long row0 = (long)b * p.T + t0;
  const int NFT = p.N >> 4;
  const int ft0 = (ct * 4 + wid) * NT;                   // this wave's first feature tile
  const int nchunks = p.KSP / SG_KC;

  // ---- loads of a chunk: this thread's NP row pieces, this wave's KC x NT weight fragments
  const TA* Ab = reinterpret_cast<const TA*>(p.A);
  int prow[NP], pcol[NP];
#pragma unroll
  for (int i = 0; i < NP; ++i) {
    const int q = tid + i * 256;
    prow[i] = q / SG_PPR;
    pcol[i] = (q - prow[i] * SG_PPR) * 8;
  }
  SgRow<TA> areg[SG_NBUF][NP];
  bf16x8 wreg[SG_NBUF][SG_KC][NT];
  auto issue = [&](int slot, int chunk) {
#pragma unroll
    for (int i = 0; i < NP; ++i) {
      const int col = min(chunk * SG_CK + pcol[i], p.K - 8);                  // columns past K are zeroed at commit
      areg[slot][i].load(Ab + (row0 + min(prow[i], nrows - 1)) * p.lda + col);
    }
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      const bf16x8* wp = p.W + ((long)min(ft0 + nt, NFT - 1) * p.KSP + chunk * SG_KC) * 64 + lane;
#pragma unroll
      for (int ks = 0; ks < SG_KC; ++ks) wreg[slot][ks][nt] = wp[ks * 64];
    }
  };
  issue(0, 0);
  issue(1, 1);
  f32x4 bias[NT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) bias[nt] = *reinterpret_cast<const f32x4*>(p.bias + min(ft0 + nt, NFT - 1) * 16 + lq * 4);

  // ---- MODE 0: GroupNorm scale / shift of this clip per input channel, in ONE barrier (round 6).  Sixteen lanes per group
  // (sixteen groups per pass of the 256 threads): a lane requests the per-part sums of ITS channels of the group (jj, jj + 16,
  // ...) and their affine straight from global memory -- everything in one round trip, beside the first two chunks -- adds
  // them in the order the round-5 prologue did (a channel's parts in order, the lane's channels in order, then the 16-lane
  // tree: DPP moves with the pairing of the xor butterfly, i.e. the same bits), derives the group's mean / rstd in registers
  // and writes the table entries of its own channels.  The round-5 form went channel sums -> LDS -> barrier -> group
  // reduce -> LDS -> barrier -> table -> barrier and its loads were four dependent loops (2.7 us of a 6.6 us workgroup).
  if constexpr (MODE == 0) {
    const int Kp = p.KSP * 32;
    const int cg = p.K / p.G;
    const int gl = tid >> 4, jj = tid & 15;
    // (3 channels per lane cover groups of up to 48 channels in one pass, 2 parts per batch: the registers of a deeper batch
    //  -- and of requesting the first batch ahead of the chunks -- took the (2, 1) form from 151 to 199 VGPRs, i.e. from three
    //  workgroups per CU to two: 736 workgroups then ran in two rounds and the launch lost 3 us at T = 100, measured)
    constexpr int NCL = 3, PB = 2;
    for (int c = p.K + tid; c < Kp; c += 256) {           // the k pad: 0 * x + 0
      gtab[c] = 0.f;
      gtab[Kp + c] = 0.f;
    }
    for (int g0 = 0; g0 < p.G; g0 += 16) {
      const int g = min(g0 + gl, p.G - 1);
      const bool gok = g0 + gl < p.G;
      float s = 0.f, q = 0.f, gw_[NCL], gb_[NCL];
#pragma unroll
      for (int u = 0; u < NCL; ++u) {
        const int ch = g * cg + min(jj + 16 * u, cg - 1);
        const bool uok = 16 * u < cg;
        gw_[u] = uok ? p.gn_w[ch] : 0.f;
        gb_[u] = uok ? p.gn_b[ch] : 0.f;
      }
      for (int c0 = 0; c0 < cg; c0 += 16 * NCL) {
        float s_[NCL], q_[NCL];
#pragma unroll
        for (int u = 0; u < NCL; ++u) s_[u] = q_[u] = 0.f;
        for (int p0 = 0; p0 < p.chs_parts; p0 += PB) {
          f32x2 v[NCL][PB];
#pragma unroll
          for (int u = 0; u < NCL; ++u)
#pragma unroll
            for (int k = 0; k < PB; ++k) {
              const int ch = g * cg + min(c0 + jj + 16 * u, cg - 1), pt = min(p0 + k, p.chs_parts - 1);
              if (c0 + 16 * u < cg && p0 + k < p.chs_parts)
                v[u][k] = *reinterpret_cast<const f32x2*>(p.chsum + (((long)pt * p.B + b) * p.K + ch) * 2);
              else v[u][k] = f32x2{0.f, 0.f};
            }
          TD_ISSUE_FENCE();
#pragma unroll
          for (int u = 0; u < NCL; ++u)
#pragma unroll
            for (int k = 0; k < PB; ++k)
              if (p0 + k < p.chs_parts) {
                s_[u] += v[u][k][0];
                q_[u] += v[u][k][1];
              }
        }
#pragma unroll
        for (int u = 0; u < NCL; ++u)
          if (c0 + jj + 16 * u < cg) {
            s += s_[u];
            q += q_[u];
          }
      }
      s = sg_row16_sum(s);
      q = sg_row16_sum(q);
      const float n = (float)cg * (float)p.T;
      const float mean = s / n;
      const float var = fmaxf(q / n - mean * mean, 0.f);
      const float rstd = 1.0f / sqrtf(var + p.eps);
      for (int c0 = 0; c0 < cg; c0 += 16 * NCL) {
#pragma unroll
        for (int u = 0; u < NCL; ++u) {
          const int c = c0 + jj + 16 * u;
          if (c < cg && gok) {
            const int ch = g * cg + c;
            const float w_ = c0 == 0 ? gw_[u] : p.gn_w[ch], b_ = c0 == 0 ? gb_[u] : p.gn_b[ch];
            const float sc = rstd * w_;
            gtab[ch] = sc;
            gtab[Kp + ch] = fmaf(-mean, sc, b_);
          }
        }
      }
    }
    __syncthreads();
  }

  f32x4 acc[MT][NT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = f32x4{0.f, 0.f, 0.f, 0.f};

  int cur = 0;
  auto commit = [&](int slot) {
    unsigned char* ab = abuf + slot * BM * SG_LDB;
#pragma unroll
    for (int i = 0; i < NP; ++i) {
      const int col = cur * SG_CK + pcol[i];
      bf16x8 o;
      if constexpr (MODE == 0) {
        const int Kp = p.KSP * 32;
        const f32x4 s0 = *reinterpret_cast<const f32x4*>(gtab + col), s1 = *reinterpret_cast<const f32x4*>(gtab + col + 4);
        const f32x4 h0 = *reinterpret_cast<const f32x4*>(gtab + Kp + col), h1 = *reinterpret_cast<const f32x4*>(gtab + Kp + col + 4);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const float sc = e < 4 ? s0[e] : s1[e - 4], sh = e < 4 ? h0[e] : h1[e - 4];
          o[e] = (bf16_t)fmaf(areg[slot][i].get(e), sc, sh);          // columns past K: 0 * x + 0
        }
      } else {
        const bool ok = col < p.K;
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = ok ? (bf16_t)areg[slot][i].get(e) : (bf16_t)0.f;
      }
      *reinterpret_cast<bf16x8*>(ab + prow[i] * SG_LDB + pcol[i] * 2) = o;
    }
  };
  auto compute = [&](int slot) {
    const unsigned char* ab = abuf + slot * BM * SG_LDB + lr * SG_LDB + lq * 16;
#pragma unroll
    for (int ks = 0; ks < SG_KC; ++ks) {
      bf16x8 xf[MT];
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) xf[mt] = *reinterpret_cast<const bf16x8*>(ab + mt * 16 * SG_LDB + ks * 64);
#pragma unroll
      for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
          acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wreg[slot][ks][nt], xf[mt], acc[mt][nt], 0, 0, 0);
    }
  };
  // stage: commit this chunk's rows to LDS, put chunk + 2 in flight, multiply
#define SG_STAGE(S, ISSUE)                      \
  commit(S);                                    \
  if (ISSUE) issue((S + 2) % SG_NBUF, cur + 2); \
  SG_LDS_BARRIER();                             \
  compute(S);                                   \
  ++cur;
  const int nsuper = nchunks / SG_NBUF;
  for (int sc = 0; sc + 1 < nsuper; ++sc) {
    SG_STAGE(0, true)
    SG_STAGE(1, true)
    SG_STAGE(2, true)
  }
  // residual rows of MODE 1: requested here, one super-iteration before they are needed
  f32x4 res[MT][NT];
  if constexpr (MODE == 1) {
    const TO* R = reinterpret_cast<const TO*>(p.resid);
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        const long r = row0 + min(mt * 16 + lr, nrows - 1);
        res[mt][nt] = SgOut<TO>::load(R + r * p.ldr + min(ft0 + nt, NFT - 1) * 16 + lq * 4);
      }
  }
  SG_STAGE(0, true)
  SG_STAGE(1, false)
  SG_STAGE(2, false)
#undef SG_STAGE

  // ---- epilogue.  Lane (lr, lq) holds features f0 + 4 lq .. + 3 of row mt * 16 + lr for each of its feature tiles.
  TO* O = reinterpret_cast<TO*>(p.out);
  float rs1[MT], rs2[MT];              // MODE 1: row sums over this wave's features
  float ps1[MT], ps2[MT];              // MODE 1 + pool
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) rs1[mt] = rs2[mt] = ps1[mt] = ps2[mt] = 0.f;
  const bool do_pool = MODE == 1 && p.pooled != nullptr;
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) {
    const bool fok = ft0 + nt < NFT;
    const int f = (ft0 + nt) * 16 + lq * 4;
    float cs1[4] = {0.f, 0.f, 0.f, 0.f}, cs2[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      const int tr = mt * 16 + lr;
      const bool rok = tr < nrows;
      f32x4 v;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float x = acc[mt][nt][e] + bias[nt][e];
        if constexpr (MODE == 1) x += res[mt][nt][e];
        else x = sg_gelu(x);
        v[e] = (MODE == 0) ? x : SgOut<TO>::rnd(x);
      }
      if (rok && fok) {
        if constexpr (MODE == 0) SgOut<bf16_t>::store(reinterpret_cast<bf16_t*>(p.out) + (row0 + tr) * p.ldo + f, v);
        else SgOut<TO>::store(O + (row0 + tr) * p.ldo + f, v);
        if constexpr (MODE == 2) {
          if (p.out16) SgOut<bf16_t>::store(p.out16 + (row0 + tr) * p.ldo + f, v);
        }
      }
      if constexpr (MODE == 1) {
        if (rok && fok) {
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            rs1[mt] += v[e];
            rs2[mt] = fmaf(v[e], v[e], rs2[mt]);
          }
        }
        if (do_pool) {
          // AdaptiveMaxPool1d with T = 2 T_out: rows (2i, 2i + 1) sit in lanes lr, lr ^ 1 of the same tile (t0 is even)
          f32x4 m;
#pragma unroll
          for (int e = 0; e < 4; ++e) m[e] = fmaxf(v[e], __shfl_xor(v[e], 1, 64));
          if (rok && fok && !(lr & 1)) {
            SgOut<TO>::store(reinterpret_cast<TO*>(p.pooled) + ((long)b * p.T_out + ((t0 + tr) >> 1)) * p.ldo + f, m);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              ps1[mt] += m[e];
              ps2[mt] = fmaf(m[e], m[e], ps2[mt]);
            }
          }
        }
      }
      if constexpr (MODE == 2) {
        if (rok) {
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            cs1[e] += v[e];
            cs2[e] = fmaf(v[e], v[e], cs2[e]);
          }
        }
      }
    }
    if constexpr (MODE == 2) {
      // per-channel sums over the rows of this tile: the 16 lanes of a row group hold 16 different rows of the same 4 features
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float a = sg_sum_r(cs1[e]), q = sg_sum_r(cs2[e]);
        if (lr == 0 && fok) {
          const f32x2 o = {a, q};
          *reinterpret_cast<f32x2*>(p.chs_out + (((long)j * p.B + b) * p.N + f + e) * 2) = o;
        }
      }
    }
  }
  if constexpr (MODE == 1) {
    // row sums: over the four lane rows of the wave, then over the four waves (fixed order) -> one partial per column tile
    float* r1 = red;                      // [4][BM][2]
    float* r2 = red + 4 * BM * 2;         // [4][BM][2] pooled (even rows used)
    __syncthreads();                      // (the GroupNorm tables / nothing else lives in `red` here, but waves may still read LDS tiles)
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      const float a = sg_sum_q(rs1[mt]), q = sg_sum_q(rs2[mt]);
      if (lq == 0) {
        r1[(wid * BM + mt * 16 + lr) * 2] = a;
        r1[(wid * BM + mt * 16 + lr) * 2 + 1] = q;
      }
      if (do_pool) {
        const float pa = sg_sum_q(ps1[mt]), pq = sg_sum_q(ps2[mt]);
        if (lq == 0) {
          r2[(wid * BM + mt * 16 + lr) * 2] = pa;
          r2[(wid * BM + mt * 16 + lr) * 2 + 1] = pq;
        }
      }
    }
    __syncthreads();
    if (tid < nrows && p.rowstat_part) {
      float a = 0.f, q = 0.f;
#pragma unroll
      for (int w = 0; w < 4; ++w) {
        a += r1[(w * BM + tid) * 2];
        q += r1[(w * BM + tid) * 2 + 1];
      }
      const f32x2 o = {a, q};
      *reinterpret_cast<f32x2*>(p.rowstat_part + ((long)ct * p.B * p.T + row0 + tid) * 2) = o;
    }
    if (do_pool && tid < nrows && !(tid & 1) && p.rowstat_pool_part) {
      float a = 0.f, q = 0.f;
#pragma unroll
      for (int w = 0; w < 4; ++w) {
        a += r2[(w * BM + tid) * 2];
        q += r2[(w * BM + tid) * 2 + 1];
      }
      const f32x2 o = {a, q};
      *reinterpret_cast<f32x2*>(p.rowstat_pool_part + ((long)ct * p.B * p.T_out + (long)b * p.T_out + ((t0 + tid) >> 1)) * 2) = o;
    }
  }
}

inline size_t sg_smem_bytes(int BM, int mode, int KSP) {
  size_t s = (size_t)SG_NBUF * BM * SG_LDB;
  if (mode == 0) s += (size_t)2 * KSP * 32 * sizeof(float) + (64 + (size_t)2 * KSP * 32) * sizeof(float);
  else s += (size_t)2 * 4 * BM * 2 * sizeof(float);
  return s;
}

template <int MT, int NT, int MODE, typename TA, typename TO>
int sg_launch(const SgpGemmP& p, hipStream_t st) {
  const size_t smem = sg_smem_bytes(16 * MT, MODE, p.KSP);
  static TdDevOnce once;
  if (!once.get()) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(sgp_gemm_kernel<MT, NT, MODE, TA, TO>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) {
      tdeed_set_error("sgp_gemm: cannot raise the dynamic LDS limit");
      return TDEED_ERR_LAUNCH;
    }
    once.set();
  }
  TD_CHECK(smem <= 160 * 1024, "sgp_gemm: %zu bytes of LDS", smem);
  hipLaunchKernelGGL((sgp_gemm_kernel<MT, NT, MODE, TA, TO>), dim3((unsigned)(p.B * p.NJ * p.nct)), dim3(256), smem, st, p);
  TD_LAUNCH_CHECK("sgp_gemm");
  return TDEED_OK;
}

template <int MODE, typename TA, typename TO>
int sg_dispatch(const SgpGemmP& p, int MT, int NT, hipStream_t st) {
  if (MT == 4 && NT == 2) return sg_launch<4, 2, MODE, TA, TO>(p, st);
  if (MT == 4 && NT == 1) return sg_launch<4, 1, MODE, TA, TO>(p, st);
  if (MT == 2 && NT == 2) return sg_launch<2, 2, MODE, TA, TO>(p, st);
  if (MT == 2 && NT == 1) return sg_launch<2, 1, MODE, TA, TO>(p, st);
  if (MT == 1 && NT == 2) return sg_launch<1, 2, MODE, TA, TO>(p, st);
  if (MT == 1 && NT == 1) return sg_launch<1, 1, MODE, TA, TO>(p, st);
  tdeed_set_error("sgp_gemm: no tile form MT=%d NT=%d", MT, NT);
  return TDEED_ERR_ARG;
}

}  // namespace

// k-steps the packed weight of a K-wide contraction must hold (zero padded): whole super-iterations of the chunk ring
extern "C" int tdeed_sgp_gemm_ksteps(int K) { return ((K + 31) / 32 + 11) / 12 * 12; }

// row tiles per clip / column tiles of a tile form (the shapes of rowstat_part, chs_out)
extern "C" int tdeed_sgp_gemm_row_tiles(int T, int MT) { return (T + 16 * MT - 1) / (16 * MT); }
extern "C" int tdeed_sgp_gemm_col_tiles(int N, int NT) { return (N + 64 * NT - 1) / (64 * NT); }

// The tile form the launcher picks for (mode, B, T, N, K): mode 0 / 3 = GroupNorm + fc1 + GELU on bf16 / fp32 rows, 1 = fc2 +
// residual, 2 = concat_fc.  Returns MT * 16 + NT.  Rules from tools/bench_sgp_gemm.py on MI355X (kernel-trace durations):
// small tiles win wherever the launch is latency bound (cfg2: every launch), because three or four resident workgroups per CU
// hide each other's round trips; the wide models' T = 100 levels are bound by bytes per CU and take the tile with the fewest.
// TDEED_SGP_FORM_<mode> = "MT,NT" overrides (experiments).
extern "C" int tdeed_sgp_gemm_form(int mode, int B, int T, int N, int K) {
  static int ov[4] = {-1, -1, -1, -1};
  static bool init = false;
  if (!init) {
    for (int m = 0; m < 4; ++m) {
      char nm[32];
      snprintf(nm, sizeof nm, "TDEED_SGP_FORM_%d", m);
      const char* e = getenv(nm);
      int a = 0, b = 0;
      if (e && sscanf(e, "%d,%d", &a, &b) == 2 && (a == 1 || a == 2 || a == 4) && (b == 1 || b == 2)) ov[m] = a * 16 + b;
    }
    init = true;
  }
  if (mode >= 0 && mode < 4 && ov[mode] >= 0 && !(mode == 3 && ov[mode] == 4 * 16 + 2)) return ov[mode];
  const long R = (long)B * T;
  const bool wide = (mode == 0 || mode == 3 ? K : N) > 384;
  int MT = 2, NT = 1;
  if (mode == 0 || mode == 3) {
    if (wide) { MT = (mode == 0 && T >= 50) ? 4 : 2; NT = 2; }
    else { MT = 2; NT = 1; }
  } else if (mode == 1) {
    MT = (!wide && T < 64) ? 1 : 2; NT = 1;
  } else {
    if (wide) { MT = R >= 1200 ? 4 : 2; NT = 1; }
    else { MT = T < 64 ? 1 : 2; NT = 1; }
  }
  return MT * 16 + NT;
}

// MODE 0: H = GELU(GroupNorm(y) . W^T + b); y [B*T][K] (dtype_a), chsum [parts][B][K][2], H bf16 [B*T][N]
extern "C" int tdeed_sgp_gemm_gn_gelu(const void* y, int B, int T, int K, const float* chsum, int chs_parts, const float* gn_w,
                                      const float* gn_b, int G, float eps, const void* Wp, const float* bias, int N, void* H,
                                      int form, int dtype_a, void* stream) {
  TD_CHECK(y && chsum && gn_w && gn_b && Wp && bias && H, "sgp_gemm_gn_gelu: null pointer");
  TD_CHECK(B > 0 && T > 0 && K % 8 == 0 && K <= 1024 && N % 16 == 0 && G > 0 && K % G == 0 && G <= 32 && chs_parts > 0,
           "sgp_gemm_gn_gelu: bad sizes");
  SgpGemmP p = {};
  const int MT = form >> 4, NT = form & 15;
  p.A = y; p.lda = K; p.W = (const bf16x8*)Wp; p.KSP = tdeed_sgp_gemm_ksteps(K); p.bias = bias; p.out = H; p.ldo = N;
  p.B = B; p.T = T; p.N = N; p.K = K; p.NJ = tdeed_sgp_gemm_row_tiles(T, MT); p.nct = tdeed_sgp_gemm_col_tiles(N, NT);
  p.ct_major = 1;
  p.chsum = chsum; p.chs_parts = chs_parts; p.gn_w = gn_w; p.gn_b = gn_b; p.G = G; p.eps = eps;
  hipStream_t st = (hipStream_t)stream;
  if (dtype_a == TDEED_BF16) return sg_dispatch<0, bf16_t, bf16_t>(p, MT, NT, st);
  if (dtype_a == TDEED_F32) return sg_dispatch<0, float, bf16_t>(p, MT, NT, st);
  tdeed_set_error("sgp_gemm_gn_gelu: bad dtype %d", dtype_a);
  return TDEED_ERR_ARG;
}

// MODE 1: out = resid + H . W^T + b; H bf16 [B*T][K]; out / resid / pooled in dtype_o; rowstat_part [nct][B*T][2];
// pooled (optional, T == 2 T_out) [B*T_out][N] with rowstat_pool_part [nct][B*T_out][2]
extern "C" int tdeed_sgp_gemm_residual(const void* H, int B, int T, int K, const void* Wp, const float* bias, int N,
                                       const void* resid, void* out, float* rowstat_part, void* pooled,
                                       float* rowstat_pool_part, int T_out, int form, int dtype_o, void* stream) {
  TD_CHECK(H && Wp && bias && resid && out, "sgp_gemm_residual: null pointer");
  TD_CHECK(B > 0 && T > 0 && K % 8 == 0 && N % 16 == 0, "sgp_gemm_residual: bad sizes");
  TD_CHECK(!pooled || (T == 2 * T_out && rowstat_pool_part), "sgp_gemm_residual: the fused max-pool needs T == 2 T_out");
  SgpGemmP p = {};
  const int MT = form >> 4, NT = form & 15;
  p.A = H; p.lda = K; p.W = (const bf16x8*)Wp; p.KSP = tdeed_sgp_gemm_ksteps(K); p.bias = bias; p.out = out; p.ldo = N;
  p.B = B; p.T = T; p.N = N; p.K = K; p.NJ = tdeed_sgp_gemm_row_tiles(T, MT); p.nct = tdeed_sgp_gemm_col_tiles(N, NT);
  p.ct_major = 0;
  p.resid = resid; p.ldr = N; p.rowstat_part = rowstat_part; p.pooled = pooled; p.rowstat_pool_part = rowstat_pool_part;
  p.T_out = T_out;
  hipStream_t st = (hipStream_t)stream;
  if (dtype_o == TDEED_BF16) return sg_dispatch<1, bf16_t, bf16_t>(p, MT, NT, st);
  if (dtype_o == TDEED_F32) return sg_dispatch<1, bf16_t, float>(p, MT, NT, st);
  tdeed_set_error("sgp_gemm_residual: bad dtype %d", dtype_o);
  return TDEED_ERR_ARG;
}

// MODE 2: out = GELU(A . W^T + b); A bf16 [B*T][K]; out in dtype_o; chs_out [NJ][B][N][2]; out16: optional bf16 copy of out
extern "C" int tdeed_sgp_gemm_gelu_chsum(const void* A, int B, int T, int K, const void* Wp, const float* bias, int N,
                                         void* out, float* chs_out, void* out16, int form, int dtype_o, void* stream) {
  TD_CHECK(A && Wp && bias && out && chs_out, "sgp_gemm_gelu_chsum: null pointer");
  TD_CHECK(B > 0 && T > 0 && K % 8 == 0 && N % 16 == 0, "sgp_gemm_gelu_chsum: bad sizes");
  SgpGemmP p = {};
  const int MT = form >> 4, NT = form & 15;
  p.A = A; p.lda = K; p.W = (const bf16x8*)Wp; p.KSP = tdeed_sgp_gemm_ksteps(K); p.bias = bias; p.out = out; p.ldo = N;
  p.B = B; p.T = T; p.N = N; p.K = K; p.NJ = tdeed_sgp_gemm_row_tiles(T, MT); p.nct = tdeed_sgp_gemm_col_tiles(N, NT);
  p.ct_major = 0;
  p.chs_out = chs_out;
  p.out16 = (bf16_t*)out16;
  hipStream_t st = (hipStream_t)stream;
  if (dtype_o == TDEED_BF16) return sg_dispatch<2, bf16_t, bf16_t>(p, MT, NT, st);
  if (dtype_o == TDEED_F32) return sg_dispatch<2, bf16_t, float>(p, MT, NT, st);
  tdeed_set_error("sgp_gemm_gelu_chsum: bad dtype %d", dtype_o);
  return TDEED_ERR_ARG;
}
