// Fused SGP encoder-decoder kernels (reference: /root/reference/model/modules.py:159-188 SGPBlock.forward, 283-318
// SGPMixer.forward, 320-363 LayerNorm), NTC layout.  The stage is latency bound (a few hundred rows, ~1 MB of weights per
// contraction): what costs is the NUMBER of dependent launches, so a block is two launches and a mixer three:
//
//   sgp_front_kernel    x -> y = x + LN(x) + fc*phi + (convw+convkw)*psi          (LayerNorm + all depthwise branches)
//   mixer_front_kernel  z, x_lo -> cat = [out1|out2|out3|out4|LN1(z)|up(LN2(x_lo))] (both LayerNorms, up-sampling, branches)
//   sgp_mlp_kernel      y -> y + fc2(GELU(fc1(GroupNorm16(y))))                    (GroupNorm + both 1x1 convs on MFMA)
//
// Row statistics are recomputed where they are needed instead of being handed from launch to launch: every front
// workgroup (one clip x 16 channels) re-derives the LayerNorm mean / rstd of its clip's rows from the L2-resident
// (T x C) slab (74 KB at T=100, C=368), every MLP workgroup re-derives the GroupNorm statistics of the clips its rows
// belong to.  That costs a couple of microseconds of L2 reads per workgroup and removes the LayerNorm / GroupNorm
// launches and their round trips through HBM.
#include "common.h"
#include "sgp_tile.h"
#include <stdlib.h>

namespace {

// ------------------------------------------------------------------------------------------ LayerNorm row statistics
// mean / rstd over C of rows [0, T) of one clip (row stride ld) -> mu[T], rs[T] in LDS.  Two adjacent lanes share a row
// (each walks half of its 16-byte chunks, all loads independent and in flight together), one shuffle joins them: a
// wave-per-row reduction would be 2 dependent cross-lane reductions per row, i.e. microseconds at one row per wave.
// Biased variance, eps inside the sqrt (modules.py:353-357); sums in one pass (var = E[x^2] - mean^2 in fp32).
template <typename T>
__device__ __forceinline__ void ln_row_stats(const T* __restrict__ slab, long ld, int T_len, int C, float eps, float* mu,
                                             float* rs) {
  constexpr int EPC = Chunk<T>::N;
  const int nch = C / EPC;
  const int half = threadIdx.x & 1;
  const int nthr = blockDim.x;
  for (int t0 = 0; t0 < T_len; t0 += nthr / 2) {
    const int t = t0 + (threadIdx.x >> 1);
    const int tt = min(t, T_len - 1);
    const T* row = slab + (long)tt * ld;
    float s = 0.f, q = 0.f;
#pragma unroll 4
    for (int ck = half; ck < nch; ck += 2) {
      float v[EPC];
      Chunk<T>::load(row + (long)ck * EPC, v);
#pragma unroll
      for (int e = 0; e < EPC; ++e) {
        s += v[e];
        q = fmaf(v[e], v[e], q);
      }
    }
    s += __shfl_xor(s, 1, 64);
    q += __shfl_xor(q, 1, 64);
    if (half == 0 && t < T_len) {
      const float m = s / (float)C;
      const float var = fmaxf(q / (float)C - m * m, 0.f);
      mu[t] = m;
      rs[t] = 1.0f / sqrtf(var + eps);
    }
  }
}

// the same statistics handed over by the producer of the rows: parts == 0: rowstat [rows][2] = (mean, rstd)
// (sgp_fold_rows_kernel, avgpool_posenc); parts > 0: [parts][rows][2] = (sum, sum of squares) of the row over each column
// tile of the contraction that stored it (sgp_gemm.hip MODE 1), summed here in order; pstride = floats per part
__device__ __forceinline__ void ln_row_stats_load(const float* __restrict__ rowstat, int parts, long pstride, int T_len,
                                                  int C, float eps, float* mu, float* rs) {
  for (int t = threadIdx.x; t < T_len; t += blockDim.x) {
    if (parts == 0) {
      const f32x2 v = *reinterpret_cast<const f32x2*>(rowstat + (long)t * 2);
      mu[t] = v[0];
      rs[t] = v[1];
    } else {
      float s = 0.f, q = 0.f;
      for (int pt = 0; pt < parts; ++pt) {
        const f32x2 v = *reinterpret_cast<const f32x2*>(rowstat + pt * pstride + (long)t * 2);
        s += v[0];
        q += v[1];
      }
      const float m = s / (float)C;
      mu[t] = m;
      rs[t] = 1.0f / sqrtf(fmaxf(q / (float)C - m * m, 0.f) + eps);
    }
  }
}

// LayerNorm applied to the interior rows of a staged tile: tile[halo + t][c] = (x - mu[t]) / den[t] * w[c] + b[c]
// -> the sum of the values this thread stored (rows tl, tl + 16, ...), rounded like a stored activation of type R
template <typename R = float>
__device__ __forceinline__ float ln_apply_tile(float* tile, int T_len, int halo, const float* mu, const float* rs, float w,
                                               float b, bool cok) {
  const int c = threadIdx.x & 15, tl = threadIdx.x >> 4;
  float s = 0.f;
  for (int t = tl; t < T_len; t += 16) {
    const float x = tile[(halo + t) * SGP_CH + c];
    const float v = cok ? round_to<R>((x - mu[t]) * rs[t] * w + b) : 0.f;
    tile[(halo + t) * SGP_CH + c] = v;
    s += v;
  }
  return s;
}

}  // namespace

// =========================================================================== SGPBlock front half (LN fused)
template <typename T>
__global__ __launch_bounds__(256) void sgp_front_kernel(const T* __restrict__ x, int T_len, int C, int ks, int up,
                                                        const float* __restrict__ ln_w, const float* __restrict__ ln_b,
                                                        float eps, const float* __restrict__ dw,
                                                        const float* __restrict__ db, T* __restrict__ y,
                                                        float* __restrict__ chsum,
                                                        const float* __restrict__ rowstat, int rs_parts) {
  extern __shared__ float sm[];
  const int halo = up >> 1;
  const int wlen = 2 * ks + up + 2;
  float* tile = sm;                                   // [(T+2h)][16]
  float* res = tile + (T_len + 2 * halo) * SGP_CH;    // [T][16]
  float* wl = res + T_len * SGP_CH;                   // [wlen][16]
  float* red = wl + wlen * SGP_CH;                    // [17][16]
  float* mu = red + 17 * SGP_CH;                      // [T]
  float* rs = mu + T_len;                             // [T]
  const int b = blockIdx.x, c0 = blockIdx.y * SGP_CH;
  const long base = (long)b * T_len * C;
  const int c = threadIdx.x & 15, tl = threadIdx.x >> 4;
  const bool cok = c0 + c < C;
  float tv[SGP_TI][Chunk<T>::N], wv[SGP_WI];
  tile_issue<T>(x + base, C, T_len, c0, C, tv);
  dw_issue(dw, wlen, c0, C, wv);
  const Bias5 bb = bias_issue(db, C, c0 + c, C);
  const float lw = ln_w[min(c0 + c, C - 1)], lb = ln_b[min(c0 + c, C - 1)];
  if (rowstat) ln_row_stats_load(rowstat + (long)b * T_len * 2, rs_parts, (long)gridDim.x * T_len * 2, T_len, C, eps, mu, rs);
  else ln_row_stats<T>(x + base, C, T_len, C, eps, mu, rs);
  tile_commit<T>(tv, T_len, c0, C, tile, halo, res);          // res starts as the raw rows: y = x + (branches) grows in place
  dw_commit(wv, wlen, c0, C, wl);
  __syncthreads();
  red[tl * SGP_CH + c] = ln_apply_tile(tile, T_len, halo, mu, rs, lw, lb, cok);      // + this thread's share of mean_T
  __syncthreads();
  const float mean_c = tile_mean_fold(red, c, T_len);
#define FRONT_FAST(KS_, UP_)                                                                                   \
  branch_runs<KS_, UP_>(tile, wl, bb, cok, c, tl, T_len, halo, mean_c,                                         \
                        [&](int t, const BranchOut& r, float o) { res[t * SGP_CH + c] += r.inst + r.conv_gate + o; });
#define FRONT_GENERIC                                                                                          \
  for (int t = tl; t < T_len; t += 16) {                                                                       \
    BranchOut r = branch_eval(tile, wl, bb, cok, t, c, halo, ks, up, mean_c);                                  \
    res[t * SGP_CH + c] += r.inst + r.conv_gate + tile[(halo + t) * SGP_CH + c];                               \
  }
  SGP_BRANCH_DISPATCH(ks, up, FRONT_FAST, FRONT_GENERIC)
#undef FRONT_FAST
#undef FRONT_GENERIC
  __syncthreads();
  store_tile<T>(res, y + base, C, 0, T_len, c0, C, (const T*)nullptr, 0);
  if (chsum) {
    // per-channel sum and sum of squares over T of the stored y (what GroupNorm of the MLP half reads): the consumer
    // then only folds 16 groups instead of re-reading the clip.  Rounded like the store.
    float s = 0.f, q = 0.f;
    if (cok)
      for (int t = tl; t < T_len; t += 16) {
        const float v = round_to<T>(res[t * SGP_CH + c]);
        s += v;
        q = fmaf(v, v, q);
      }
    float* redq = tile;                 // [16][16] second scratch: the LayerNorm tile is dead behind the barrier above
    red[tl * SGP_CH + c] = s;
    redq[tl * SGP_CH + c] = q;
    __syncthreads();
    if (threadIdx.x < SGP_CH && c0 + (int)threadIdx.x < C) {
      float a = 0.f, bq = 0.f;
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        a += red[i * SGP_CH + threadIdx.x];
        bq += redq[i * SGP_CH + threadIdx.x];
      }
      chsum[((long)b * C + c0 + threadIdx.x) * 2] = a;
      chsum[((long)b * C + c0 + threadIdx.x) * 2 + 1] = bq;
    }
  }
}

static size_t front_smem(int T_len, int ks, int up, int ntiles, int nres, int nstat) {
  const int halo = up / 2, wlen = 2 * ks + up + 2;
  return (size_t)(ntiles * (T_len + 2 * halo) * SGP_CH + nres * T_len * SGP_CH + ntiles * wlen * SGP_CH + 17 * SGP_CH +
                  nstat * 2 * T_len) * sizeof(float);
}

extern "C" int tdeed_sgp_front_fwd(const void* x, int B, int T, int C, int ks, int up, const float* ln_w,
                                   const float* ln_b, float eps, const float* dw, const float* db, void* y,
                                   float* chsum, const float* rowstat, int rowstat_parts, int dtype, void* stream) {
  TD_CHECK(x && ln_w && ln_b && dw && db && y, "sgp_front: null pointer");
  TD_CHECK(rowstat_parts >= 0 && rowstat_parts <= 64, "sgp_front: rowstat_parts");
  TD_CHECK(B > 0 && T > 0 && C % 8 == 0 && ks % 2 == 1 && up % 2 == 1 && up >= ks, "sgp_front: bad sizes");
  TD_CHECK(T <= (dtype == TDEED_BF16 ? 512 : 256) && 2 * ks + up + 2 <= 80,
           "sgp_front: T=%d / ks=%d up=%d beyond the staging registers", T, ks, up);
  TD_CHECK(C <= 4 * 64 * (dtype == TDEED_BF16 ? 8 : 4), "sgp_front: C=%d too wide", C);
  const size_t smem = front_smem(T, ks, up, 1, 1, 1);
  TD_CHECK(smem <= 64 * 1024, "sgp_front: T=%d too long for the LDS window", T);
  dim3 grid(B, cdiv(C, SGP_CH));
  hipStream_t st = (hipStream_t)stream;
  if (dtype == TDEED_F32)
    hipLaunchKernelGGL(sgp_front_kernel<float>, grid, dim3(256), smem, st, (const float*)x, T, C, ks, up, ln_w, ln_b, eps,
                       dw, db, (float*)y, chsum, rowstat, rowstat_parts);
  else if (dtype == TDEED_BF16)
    hipLaunchKernelGGL(sgp_front_kernel<bf16_t>, grid, dim3(256), smem, st, (const bf16_t*)x, T, C, ks, up, ln_w, ln_b,
                       eps, dw, db, (bf16_t*)y, chsum, rowstat, rowstat_parts);
  else { tdeed_set_error("sgp_front: bad dtype %d", dtype); return TDEED_ERR_ARG; }
  TD_LAUNCH_CHECK("sgp_front");
  return TDEED_OK;
}

// =========================================================================== SGPMixer front half (LN1, LN2 fused)
// cat row = [out1 | out2 | out3 | out4 | zn | xu], each C wide (modules.py:302-308).  The two sources are independent until
// the concat: blockIdx.z = 0 handles z (LN1, slabs 4, 0, 2), blockIdx.z = 1 handles x_lo (LN2, up-sampling, slabs 5, 1, 3).
template <typename T, typename TC>
__global__ __launch_bounds__(256) void mixer_front_kernel(const T* __restrict__ z, const T* __restrict__ xlo, int T_hi,
                                                          int T_lo, int C, int ks, int up,
                                                          const float* __restrict__ ln1_w, const float* __restrict__ ln1_b,
                                                          const float* __restrict__ ln2_w, const float* __restrict__ ln2_b,
                                                          float eps, const float* __restrict__ dw1,
                                                          const float* __restrict__ db1, const float* __restrict__ dw2,
                                                          const float* __restrict__ db2, TC* __restrict__ cat,
                                                          const float* __restrict__ rowstat_z, int parts_z,
                                                          const float* __restrict__ rowstat_x, int parts_x) {
  extern __shared__ float sm[];
  const int halo = up >> 1;
  const int wlen = 2 * ks + up + 2;
  const int trows = T_hi + 2 * halo;
  float* tile = sm;                            // the source sequence at T_hi with zero halos (zn or xu)
  float* res = tile + trows * SGP_CH;          // [T_hi][16] (holds xn [T_lo][16] during the up-sampling)
  float* res2 = res + T_hi * SGP_CH;           // [T_hi][16]
  float* wl = res2 + T_hi * SGP_CH;
  float* red = wl + wlen * SGP_CH;             // [17][16]
  float* mu = red + 17 * SGP_CH;               // [T_hi]
  float* rs = mu + T_hi;
  const int b = blockIdx.x, c0 = blockIdx.y * SGP_CH, src = blockIdx.z;
  const long ldc = 6L * C;
  TC* crow = cat + (long)b * T_hi * ldc;
  const int T_src = src == 0 ? T_hi : T_lo;
  const T* sb = src == 0 ? z + (long)b * T_hi * C : xlo + (long)b * T_lo * C;
  const float* dw = src == 0 ? dw1 : dw2;
  const float* db = src == 0 ? db1 : db2;
  const int c = threadIdx.x & 15, tl = threadIdx.x >> 4;
  const bool cok = c0 + c < C;
  const int cc = min(c0 + c, C - 1);
  const float lw = src == 0 ? ln1_w[cc] : ln2_w[cc], lb = src == 0 ? ln1_b[cc] : ln2_b[cc];
  {
    float sv[SGP_TI][Chunk<T>::N], wv[SGP_WI];
    tile_issue<T>(sb, C, T_src, c0, C, sv);
    dw_issue(dw, wlen, c0, C, wv);
    const float* rst = src == 0 ? rowstat_z : rowstat_x;
    if (rst) ln_row_stats_load(rst + (long)b * T_src * 2, src == 0 ? parts_z : parts_x, (long)gridDim.x * T_src * 2, T_src, C,
                               eps, mu, rs);
    else ln_row_stats<T>(sb, C, T_src, C, eps, mu, rs);
    if (src == 0) tile_commit<T>(sv, T_hi, c0, C, tile, halo);
    else tile_commit<T>(sv, T_lo, c0, C, res, 0);          // x_lo at T_lo -> res (no halo)
    dw_commit(wv, wlen, c0, C, wl);
  }
  const Bias5 bb = bias_issue(db, C, c0 + c, C);
  if (src == 1)
    for (int i = threadIdx.x; i < 2 * halo * SGP_CH; i += 256) {
      int r = i / SGP_CH, c_ = i - r * SGP_CH;
      int row = r < halo ? r : (T_hi + r);
      tile[row * SGP_CH + c_] = 0.f;
    }
  __syncthreads();
  // the normalised sequences are tensors in the reference: rounded like stored activations of the stream's type; each thread
  // keeps the sum of the rows it produced (its share of mean_T, folded by every thread behind the next barrier)
  if (src == 0) {
    red[tl * SGP_CH + c] = ln_apply_tile<T>(tile, T_hi, halo, mu, rs, lw, lb, cok);        // zn = LN1(z)
  } else {
    ln_apply_tile<T>(res, T_lo, 0, mu, rs, lw, lb, cok);         // xn = LN2(x_lo), before the up-sampling (modules.py:287-288)
    __syncthreads();
    const float scale = (T_hi > 1) ? (float)(T_lo - 1) / (float)(T_hi - 1) : 0.f;
    float sacc = 0.f;
    for (int t = tl; t < T_hi; t += 16) {
      float v;
      if (T_hi == T_lo) {
        v = res[t * SGP_CH + c];
      } else {
        const float sp = scale * (float)t;
        const int i0 = (int)sp;
        const int i1 = i0 + (i0 < T_lo - 1 ? 1 : 0);
        const float l1 = fminf(fmaxf(sp - (float)i0, 0.f), 1.f);
        const float l0 = 1.f - l1;
        v = l0 * res[i0 * SGP_CH + c] + l1 * res[i1 * SGP_CH + c];
      }
      v = round_to<T>(v);
      tile[(halo + t) * SGP_CH + c] = v;
      sacc += v;
    }
    red[tl * SGP_CH + c] = sacc;
  }
  __syncthreads();
  // slab 4 = zn, slab 5 = xu
  store_tile<TC>(tile + halo * SGP_CH, crow + (long)(4 + src) * C, ldc, 0, T_hi, c0, C, (const TC*)nullptr, 0);
  const float mean_c = tile_mean_fold(red, c, T_hi);
#define MIX_FAST(KS_, UP_)                                                                                     \
  branch_runs<KS_, UP_>(tile, wl, bb, cok, c, tl, T_hi, halo, mean_c, [&](int t, const BranchOut& r, float) {  \
    res[t * SGP_CH + c] = r.conv_gate;                                                                         \
    res2[t * SGP_CH + c] = r.inst;                                                                             \
  });
#define MIX_GENERIC                                                                                            \
  for (int t = tl; t < T_hi; t += 16) {                                                                        \
    BranchOut r = branch_eval(tile, wl, bb, cok, t, c, halo, ks, up, mean_c);                                  \
    res[t * SGP_CH + c] = r.conv_gate;                                                                         \
    res2[t * SGP_CH + c] = r.inst;                                                                             \
  }
  SGP_BRANCH_DISPATCH(ks, up, MIX_FAST, MIX_GENERIC)
#undef MIX_FAST
#undef MIX_GENERIC
  __syncthreads();
  store_tile<TC>(res, crow + (long)src * C, ldc, 0, T_hi, c0, C, (const TC*)nullptr, 0);
  store_tile<TC>(res2, crow + (long)(2 + src) * C, ldc, 0, T_hi, c0, C, (const TC*)nullptr, 0);
}

extern "C" int tdeed_mixer_front_fwd(const void* z, const void* xlo, int B, int T_hi, int T_lo, int C, int ks, int up,
                                     const float* ln1_w, const float* ln1_b, const float* ln2_w, const float* ln2_b,
                                     float eps, const float* dw1, const float* db1, const float* dw2, const float* db2,
                                     void* cat, const float* rowstat_z, int parts_z, const float* rowstat_x, int parts_x,
                                     int dtype, int dtype_cat, void* stream) {
  TD_CHECK(z && xlo && ln1_w && ln1_b && ln2_w && ln2_b && dw1 && db1 && dw2 && db2 && cat, "mixer_front: null pointer");
  TD_CHECK(parts_z >= 0 && parts_z <= 64 && parts_x >= 0 && parts_x <= 64, "mixer_front: rowstat parts");
  TD_CHECK(B > 0 && T_hi >= T_lo && T_lo > 0 && C % 8 == 0 && ks % 2 == 1 && up % 2 == 1 && up >= ks,
           "mixer_front: bad sizes");
  TD_CHECK(T_hi <= (dtype == TDEED_BF16 ? 512 : 256) && 2 * ks + up + 2 <= 80,
           "mixer_front: T=%d / ks=%d up=%d beyond the staging registers", T_hi, ks, up);
  TD_CHECK(C <= 4 * 64 * (dtype == TDEED_BF16 ? 8 : 4), "mixer_front: C=%d too wide", C);
  const size_t smem = front_smem(T_hi, ks, up, 1, 2, 1);
  TD_CHECK(smem <= 64 * 1024, "mixer_front: T=%d too long for the LDS window", T_hi);
  dim3 grid(B, cdiv(C, SGP_CH), 2);
  hipStream_t st = (hipStream_t)stream;
  if (dtype == TDEED_F32 && dtype_cat == TDEED_F32)
    hipLaunchKernelGGL((mixer_front_kernel<float, float>), grid, dim3(256), smem, st, (const float*)z, (const float*)xlo, T_hi,
                       T_lo, C, ks, up, ln1_w, ln1_b, ln2_w, ln2_b, eps, dw1, db1, dw2, db2, (float*)cat, rowstat_z, parts_z,
                       rowstat_x, parts_x);
  else if (dtype == TDEED_F32 && dtype_cat == TDEED_BF16)
    // fp32 residual stream, bf16 contraction operand (the throughput mode of round 5): the six slabs are rounded once, at the store
    hipLaunchKernelGGL((mixer_front_kernel<float, bf16_t>), grid, dim3(256), smem, st, (const float*)z, (const float*)xlo, T_hi,
                       T_lo, C, ks, up, ln1_w, ln1_b, ln2_w, ln2_b, eps, dw1, db1, dw2, db2, (bf16_t*)cat, rowstat_z, parts_z,
                       rowstat_x, parts_x);
  else if (dtype == TDEED_BF16 && dtype_cat == TDEED_BF16)
    hipLaunchKernelGGL((mixer_front_kernel<bf16_t, bf16_t>), grid, dim3(256), smem, st, (const bf16_t*)z, (const bf16_t*)xlo,
                       T_hi, T_lo, C, ks, up, ln1_w, ln1_b, ln2_w, ln2_b, eps, dw1, db1, dw2, db2, (bf16_t*)cat, rowstat_z,
                       parts_z, rowstat_x, parts_x);
  else { tdeed_set_error("mixer_front: bad dtypes %d / %d", dtype, dtype_cat); return TDEED_ERR_ARG; }
  TD_LAUNCH_CHECK("mixer_front");
  return TDEED_OK;
}

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
