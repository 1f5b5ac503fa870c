// Parked in round 5 (superseded by t-deed_amd/csrc/sgp_gemm.hip): the round-2 fused GroupNorm + fc1 + GELU + fc2 launch that
// lived at the end of csrc/sgp_fused.hip.  Not compiled.
#include "../../t-deed_amd/csrc/common.h"

// =========================================================================== GroupNorm + MLP + residual (bf16, MFMA)
// out[r][:] = y[r][:] + W2 . GELU(W1 . GN16(y)[r][:] + b1) + b2        (modules.py:186, 316; mlp = Conv1d(C,4C,1), GELU,
// Conv1d(4C,C,1)).  One workgroup (8 waves, two per SIMD) owns ROWS = 16*MT whole rows and the hidden chunks
// {blockIdx.y, blockIdx.y + S, ...} of the 4 chunks of C hidden units: the normalised rows sit in LDS (bf16, the MFMA B
// operand), a hidden chunk is produced into a second LDS tile and consumed from there, weights stream from L2 straight
// into MFMA A-operand fragments (pre-packed in fragment order, see loadw) through a register ring, the fc2 accumulators stay in registers.  With S = 1 the epilogue adds b2 and the
// residual and writes bf16 rows; with S > 1 (the stage is latency bound at a few hundred rows: more workgroups, each
// streaming 1/S of the weights) it writes fp32 partials that sgp_mlp_fold_kernel sums in a fixed order.
// Accumulator layout (v_mfma_f32_16x16x32_bf16, weights as A): lane l holds output features 4*(l>>4) .. +3 for
// activation row l&15.
constexpr int MLP_MAXCL = 8;      // clips one row tile may touch
constexpr int MLP_NW = 8;         // waves per workgroup

// GELU(erf) with erf from Abramowitz-Stegun 7.1.26 (|error| <= 1.5e-7: far below bf16 resolution): ~15 VALU instructions
// instead of erff's ~45; a 64-row tile evaluates 94k of them on four SIMDs.
__device__ __forceinline__ float gelu_fast(float x) {
  const float z = fabsf(x) * 0.70710678118654752440f;
  const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, z, 1.0f));
  const float poly = t * fmaf(t, fmaf(t, fmaf(t, fmaf(t, 1.061405429f, -1.453152027f), 1.421413741f), -0.284496736f), 0.254829592f);
  const float e = 1.0f - poly * __builtin_amdgcn_exp2f(-1.44269504088896341f * z * z);
  return 0.5f * x * (1.0f + copysignf(e, x));
}

template <int MT, int NT>
__global__ __launch_bounds__(MLP_NW * 64, MLP_NW / 4) void sgp_mlp_kernel(
    const bf16_t* __restrict__ y, int R, int T_len, int C, int G, const float* __restrict__ gn_w,
    const float* __restrict__ gn_b, float eps, const bf16_t* __restrict__ W1, const float* __restrict__ b1,
    const bf16_t* __restrict__ W2, const float* __restrict__ b2, bf16_t* __restrict__ out, float* __restrict__ partial,
    const float* __restrict__ chsum) {
  constexpr int ROWS = 16 * MT;
  constexpr int NTHR = MLP_NW * 64;
  extern __shared__ __attribute__((aligned(16))) unsigned char smraw[];
  const int KP = (C + 31) / 32 * 32;          // K padded to whole MFMA steps
  const int LD = KP + 8;                      // row stride in elements: rows shift by one 16-B bank slot
  bf16_t* At = reinterpret_cast<bf16_t*>(smraw);
  bf16_t* Ht = At + ROWS * LD;
  float* gstat = reinterpret_cast<float*>(Ht + ROWS * LD);        // [MLP_MAXCL][G][2] (mean, rstd)
  float* part = reinterpret_cast<float*>(Ht);                     // GN reduction scratch aliases the hidden tile
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, lr = lane & 15, lq = lane >> 4;
  const int r0 = blockIdx.x * ROWS;
  const int S = gridDim.y;
  const int c_lo = r0 / T_len, c_hi = min(R - 1, r0 + ROWS - 1) / T_len;
  const int cg = C / G;
  const int nck = C / 8;

  // ---- GroupNorm statistics of the clips this tile touches: per-channel sum / sum of squares come from the producer
  // (chsum [clips][C][2], written by sgp_front) or, for other producers, from one pass over the clips' slabs here;
  // the fold over a group's channels is a fixed-order butterfly either way.
  auto fold_groups = [&](const float* chs, int ci) {
    if (tid < G * 16) {                        // 16 lanes per group
      const int g = tid >> 4, j = tid & 15;
      float a = 0.f, bq = 0.f;
      for (int cl = j; cl < cg; cl += 16) {
        a += chs[(g * cg + cl) * 2];
        bq += chs[(g * cg + cl) * 2 + 1];
      }
#pragma unroll
      for (int o = 8; o > 0; o >>= 1) {
        a += __shfl_xor(a, o, 64);
        bq += __shfl_xor(bq, o, 64);
      }
      if (j == 0) {
        const float n = (float)(cg * T_len);
        const float mean = a / n;
        const float var = fmaxf(bq / n - mean * mean, 0.f);
        gstat[((ci - c_lo) * G + g) * 2] = mean;
        gstat[((ci - c_lo) * G + g) * 2 + 1] = 1.0f / sqrtf(var + eps);
      }
    }
  };
  if (chsum) {
    for (int ci = c_lo; ci <= c_hi; ++ci) fold_groups(chsum + (long)ci * C * 2, ci);
    __syncthreads();
  } else {
    const int TL = min(NTHR / nck, 8);         // threads per channel chunk along t (8 x C x 2 floats of scratch fit Ht)
    const int ck = tid % nck, tl = tid / nck;
    float* chs = part + TL * C * 2;            // [C][2] per-channel totals
    for (int ci = c_lo; ci <= c_hi; ++ci) {
      const bf16_t* slab = y + (long)ci * T_len * C;
      if (tl < TL) {
        float s[8], q[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) s[e] = q[e] = 0.f;
        for (int tb = tl; tb < T_len; tb += 8 * TL) {             // 8 rows per round, all loads issued before the sums
          bf16x8 v[8];
#pragma unroll
          for (int u = 0; u < 8; ++u) v[u] = *reinterpret_cast<const bf16x8*>(slab + (long)min(tb + u * TL, T_len - 1) * C + ck * 8);
          TD_ISSUE_FENCE();
#pragma unroll
          for (int u = 0; u < 8; ++u)
            if (tb + u * TL < T_len) {
#pragma unroll
              for (int e = 0; e < 8; ++e) {
                const float f = (float)v[u][e];
                s[e] += f;
                q[e] = fmaf(f, f, q[e]);
              }
            }
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          part[(tl * C + ck * 8 + e) * 2] = s[e];
          part[(tl * C + ck * 8 + e) * 2 + 1] = q[e];
        }
      }
      __syncthreads();
      for (int ch = tid; ch < C; ch += NTHR) {
        float a = 0.f, bq = 0.f;
        for (int j = 0; j < TL; ++j) {
          a += part[(j * C + ch) * 2];
          bq += part[(j * C + ch) * 2 + 1];
        }
        chs[ch * 2] = a;
        chs[ch * 2 + 1] = bq;
      }
      __syncthreads();
      fold_groups(chs, ci);
      __syncthreads();
    }
  }

  // ---- stage A = GN(y rows) as bf16, K pad columns zero.  Two phases: every load of the tile is issued before the
  // first use (one memory round trip for the whole tile instead of one per item).
  {
    constexpr int MAXIT = (ROWS * 96 / (MT == 4 ? 2 : 1) + NTHR - 1) / NTHR;     // 64 rows: C <= 384; 32 rows: C <= 768
    const IDiv dck(nck), dcg(cg), dT(T_len);
    const int nitem = ROWS * nck;
    bf16x8 yv[MAXIT];
    f32x4 wv[MAXIT][2], bv[MAXIT][2];
#pragma unroll
    for (int it = 0; it < MAXIT; ++it) {
      const int i = min(tid + it * NTHR, nitem - 1);
      int row, ck;
      dck.divmod(i, row, ck);
      const long r = min((long)r0 + row, (long)R - 1);
      yv[it] = *reinterpret_cast<const bf16x8*>(y + r * C + ck * 8);
      wv[it][0] = *reinterpret_cast<const f32x4*>(gn_w + ck * 8);
      wv[it][1] = *reinterpret_cast<const f32x4*>(gn_w + ck * 8 + 4);
      bv[it][0] = *reinterpret_cast<const f32x4*>(gn_b + ck * 8);
      bv[it][1] = *reinterpret_cast<const f32x4*>(gn_b + ck * 8 + 4);
    }
    TD_ISSUE_FENCE();
#pragma unroll
    for (int it = 0; it < MAXIT; ++it) {
      const int i = tid + it * NTHR;
      if (i < nitem) {
        int row, ck;
        dck.divmod(i, row, ck);
        const long r = (long)r0 + row;
        bf16x8 o;
        if (r < R) {
          const int ci = dT.div((int)r) - c_lo;
          const int g0 = dcg.div(ck * 8);
          const int split = (g0 + 1) * cg - ck * 8;                // elements >= split belong to the next group
          const float* st = gstat + (ci * G) * 2;
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            int g = g0;
            if (cg >= 8) g += (e >= split) ? 1 : 0; else g = dcg.div(ck * 8 + e);
            o[e] = (bf16_t)(((float)yv[it][e] - st[g * 2]) * st[g * 2 + 1] * wv[it][e >> 2][e & 3] + bv[it][e >> 2][e & 3]);
          }
        } else {
#pragma unroll
          for (int e = 0; e < 8; ++e) o[e] = (bf16_t)0.f;
        }
        *reinterpret_cast<bf16x8*>(At + row * LD + ck * 8) = o;
      }
    }
    const int padc = (LD - C) / 8;              // 16-B chunks of padding per row (K pad + bank-shift pad)
    for (int i = tid; i < ROWS * padc; i += NTHR) {
      const int row = i / padc, ck = i - row * padc;
      bf16x8 zz;
#pragma unroll
      for (int e = 0; e < 8; ++e) zz[e] = (bf16_t)0.f;
      *reinterpret_cast<bf16x8*>(At + row * LD + C + ck * 8) = zz;
    }
  }
  __syncthreads();

  const int KS = KP / 32;
  f32x4 acc2[NT][MT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt)
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) acc2[nt][mt] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int ntile = C / 16;                      // 16-row weight tiles per hidden chunk / of the output
  const bf16_t* wrow[NT];
  // one K sweep: acc[nt][mt] += W[rows of this wave's tiles][k] . X[row][k], weights through a 3-deep register ring
  // weights arrive pre-packed in fragment order ([chunk][tile][k-step][lane][8], K zero-padded: engine.pack_mlp_frags):
  // one wave-load is 1 KB of consecutive bytes.  Row-major weights would make every load 16 row segments of 64 B, and the
  // address path (16 cache lines per instruction, 3 instructions per k-step and wave) then paces the kernel 4x below the
  // MFMA rate.  No select on the loaded value (a select right behind a load is a wait for it): a ring slot past KS
  // re-reads step 0 and is skipped.
  auto loadw = [&](bf16x8 (&w)[NT], int ks) {
    const int ko = (ks < KS ? ks : 0) * 512;
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) w[nt] = *reinterpret_cast<const bf16x8*>(wrow[nt] + ko);
  };
  // RING k-steps of weight fragments are in flight per wave (RING x NT KB; 8 waves): the stream is latency bound
  // (~1 us from L2 under load), so bytes in flight per CU set its rate -- 48 KB (2 steps ahead) gave ~50 GB/s per CU
  constexpr int RING = NT <= 3 ? 6 : 3;         // the 6-tile variant has no registers for more
  auto sweep = [&](const bf16_t* X, f32x4 (&acc)[NT][MT]) {
    bf16x8 w[RING][NT];
    auto fma_step = [&](const bf16x8 (&wf)[NT], int ks) {
      if (ks < KS) {
        bf16x8 xf[MT];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) xf[mt] = *reinterpret_cast<const bf16x8*>(X + (mt * 16 + lr) * LD + ks * 32 + lq * 8);
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
          for (int mt = 0; mt < MT; ++mt) acc[nt][mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[nt], xf[mt], acc[nt][mt], 0, 0, 0);
      }
    };
#pragma unroll
    for (int d = 0; d < RING - 1; ++d) loadw(w[d], d);
    for (int ks = 0; ks < KS; ks += RING) {
#pragma unroll
      for (int d = 0; d < RING; ++d) {
        loadw(w[(d + RING - 1) % RING], ks + d + RING - 1);
        fma_step(w[d], ks + d);
      }
    }
  };

  for (int chunk = blockIdx.y; chunk < 4; chunk += S) {
    // ---------------- fc1: hidden units [chunk*C, chunk*C + C), this wave's tiles wid*NT .. +NT-1
    {
      f32x4 acc1[NT][MT];
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        const int tt = min(wid * NT + nt, ntile - 1);                // tiles past C/16 recompute the last one (results dropped)
        wrow[nt] = W1 + (((long)chunk * ntile + tt) * KS * 64 + lane) * 8;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) acc1[nt][mt] = f32x4{0.f, 0.f, 0.f, 0.f};
      }
      sweep(At, acc1);
      // bias + GELU -> hidden tile [row][unit] (units past C of this chunk are zero: they are K padding of fc2)
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        const int f0 = (wid * NT + nt) * 16 + lq * 4;
        if (f0 < KP) {
          float bias[4];
#pragma unroll
          for (int j = 0; j < 4; ++j) bias[j] = (f0 + j < C) ? b1[(long)chunk * C + f0 + j] : 0.f;
#pragma unroll
          for (int mt = 0; mt < MT; ++mt) {
            bf16x4 o;
#pragma unroll
            for (int j = 0; j < 4; ++j) o[j] = (bf16_t)((f0 + j < C) ? gelu_fast(acc1[nt][mt][j] + bias[j]) : 0.f);
            *reinterpret_cast<bf16x4*>(Ht + (mt * 16 + lr) * LD + f0) = o;
          }
        }
      }
    }
    __syncthreads();
    // ---------------- fc2 partial: out features of this wave += W2[:, chunk*C .. +C) . hidden chunk
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      const int tt = min(wid * NT + nt, ntile - 1);
      wrow[nt] = W2 + (((long)chunk * ntile + tt) * KS * 64 + lane) * 8;
    }
    sweep(Ht, acc2);
    __syncthreads();
  }

  // ---------------- epilogue: S == 1: + b2 + residual y, bf16 rows; S > 1: fp32 partial of this hidden slice
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) {
    const int n0 = (wid * NT + nt) * 16 + lq * 4;
    if (n0 < C) {
      float bias[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) bias[j] = b2[n0 + j];
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
        const long r = (long)r0 + mt * 16 + lr;
        if (r < R) {
          if (S == 1) {
            const bf16x4 yr = *reinterpret_cast<const bf16x4*>(y + r * C + n0);
            bf16x4 o;
#pragma unroll
            for (int j = 0; j < 4; ++j) o[j] = (bf16_t)(acc2[nt][mt][j] + bias[j] + (float)yr[j]);
            *reinterpret_cast<bf16x4*>(out + r * C + n0) = o;
          } else {
            *reinterpret_cast<f32x4*>(partial + ((long)blockIdx.y * R + r) * C + n0) = acc2[nt][mt];
          }
        }
      }
    }
  }
}

// out = sum_s partial[s] (fixed order) + b2 + y, bf16
__global__ __launch_bounds__(256) void sgp_mlp_fold_kernel(const float* __restrict__ partial, int S, long RC, int C,
                                                           const float* __restrict__ b2, const bf16_t* __restrict__ y,
                                                           bf16_t* __restrict__ out) {
  const long i = ((long)blockIdx.x * 256 + threadIdx.x) * 8;
  if (i >= RC) return;
  const int c = (int)(i % C);
  float a[8];
  Chunk<bf16_t>::load(y + i, a);
#pragma unroll
  for (int e = 0; e < 8; ++e) a[e] += b2[c + e];
  for (int s = 0; s < S; ++s) {
    const f32x4 p0 = *reinterpret_cast<const f32x4*>(partial + (long)s * RC + i);
    const f32x4 p1 = *reinterpret_cast<const f32x4*>(partial + (long)s * RC + i + 4);
#pragma unroll
    for (int e = 0; e < 4; ++e) { a[e] += p0[e]; a[4 + e] += p1[e]; }
  }
  Chunk<bf16_t>::store(out + i, a);
}

static size_t mlp_smem(int C, int rows, int G) {
  const int KP = (C + 31) / 32 * 32, LD = KP + 8;
  return (size_t)2 * rows * LD * 2 + (size_t)MLP_MAXCL * G * 2 * sizeof(float);
}

// rows per workgroup for a width: 64 while a wave's 3 feature tiles cover C (8 waves x 3 x 16 = 384), else 32 with 6 tiles
static int mlp_rows(int C) { return C <= 384 ? 64 : 32; }

// hidden-chunk split: the 4 chunks of C hidden units go to `S` workgroups per row tile while the row tiles alone leave
// most of the chip idle
extern "C" int tdeed_sgp_mlp_splits(int R, int C) {
  const int tiles = (R + mlp_rows(C) - 1) / mlp_rows(C);
  return tiles * 4 <= 256 ? 4 : (tiles * 2 <= 256 ? 2 : 1);
}

// 1 when the fused GroupNorm+MLP kernel serves this geometry (bf16; C a multiple of 16 groups and of 8; C <= 768)
extern "C" int tdeed_sgp_mlp_fits(int R, int T, int C, int G) {
  if (C % 16 != 0 || G <= 0 || G > 32 || C % G != 0 || C > 768 || C < 64 || T <= 0 || R % T != 0) return 0;
  const int rows = mlp_rows(C);
  if ((rows - 1) / T + 2 > MLP_MAXCL) return 0;
  const int nck = C / 8;
  const int TL = (MLP_NW * 64) / nck < 8 ? (MLP_NW * 64) / nck : 8;
  if (TL < 1) return 0;
  if ((size_t)(TL + 1) * C * 2 * sizeof(float) > (size_t)rows * ((C + 31) / 32 * 32 + 8) * 2) return 0;   // GN scratch aliases Ht
  return mlp_smem(C, rows, G) <= 160 * 1024 ? 1 : 0;
}

// partial: fp32 scratch of tdeed_sgp_mlp_splits(R, C) * R * C floats (unused when the split is 1)
extern "C" int tdeed_sgp_mlp_fwd(const void* y, int R, int T, int C, int G, const float* gn_w, const float* gn_b, float eps,
                                 const void* W1, const float* b1, const void* W2, const float* b2, void* out, float* partial,
                                 const float* chsum, void* stream) {
  TD_CHECK(y && gn_w && gn_b && W1 && b1 && W2 && b2 && out, "sgp_mlp: null pointer");
  TD_CHECK(tdeed_sgp_mlp_fits(R, T, C, G), "sgp_mlp: geometry R=%d T=%d C=%d G=%d not served", R, T, C, G);
  const int rows = mlp_rows(C);
  const size_t smem = mlp_smem(C, rows, G);
  static const int force_s = getenv("TDEED_SGP_MLP_SPLIT") ? atoi(getenv("TDEED_SGP_MLP_SPLIT")) : 0;
  const int S = (force_s == 1 || force_s == 2 || force_s == 4) ? force_s : tdeed_sgp_mlp_splits(R, C);
  TD_CHECK(S == 1 || partial, "sgp_mlp: split %d needs the partial buffer", S);
  hipStream_t st = (hipStream_t)stream;
  static TdDevOnce attr_set;
  if (!attr_set.get()) {
    hipError_t e = hipFuncSetAttribute((const void*)sgp_mlp_kernel<4, 3>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e == hipSuccess) e = hipFuncSetAttribute((const void*)sgp_mlp_kernel<2, 6>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) { tdeed_set_error("sgp_mlp: hipFuncSetAttribute: %s", hipGetErrorString(e)); return TDEED_ERR_RUNTIME; }
    attr_set.set();
  }
  if (rows == 64)
    hipLaunchKernelGGL((sgp_mlp_kernel<4, 3>), dim3(cdiv(R, 64), S), dim3(MLP_NW * 64), smem, st, (const bf16_t*)y, R, T, C, G,
                       gn_w, gn_b, eps, (const bf16_t*)W1, b1, (const bf16_t*)W2, b2, (bf16_t*)out, partial, chsum);
  else
    hipLaunchKernelGGL((sgp_mlp_kernel<2, 6>), dim3(cdiv(R, 32), S), dim3(MLP_NW * 64), smem, st, (const bf16_t*)y, R, T, C, G,
                       gn_w, gn_b, eps, (const bf16_t*)W1, b1, (const bf16_t*)W2, b2, (bf16_t*)out, partial, chsum);
  if (S > 1) {
    const long RC = (long)R * C;
    hipLaunchKernelGGL(sgp_mlp_fold_kernel, dim3((unsigned)((RC / 8 + 255) / 256)), dim3(256), 0, st, partial, S, RC, C, b2,
                       (const bf16_t*)y, (bf16_t*)out);
  }
  TD_LAUNCH_CHECK("sgp_mlp");
  return TDEED_OK;
}
