// Fused SGP encoder-decoder kernels (reference: /root/reference/model/modules.py:159-188 SGPBlock.forward, 283-318
// SGPMixer.forward, 320-363 LayerNorm), NTC layout.  The stage is latency bound (a few hundred rows, ~1 MB of weights per
// contraction): what costs is the NUMBER of dependent launches, so a block is two launches and a mixer three:
//
//   sgp_front_kernel    x -> y = x + LN(x) + fc*phi + (convw+convkw)*psi          (LayerNorm + all depthwise branches)
//   mixer_front_kernel  z, x_lo -> cat = [out1|out2|out3|out4|LN1(z)|up(LN2(x_lo))] (both LayerNorms, up-sampling, branches)
//   (the contractions -- GroupNorm + fc1 + GELU, fc2 + residual, concat_fc -- are csrc/sgp_gemm.hip since round 5; the
//   round-2 / round-3 fused MLP kernels are parked under experiments/r5_parked/)
//
// Row statistics are recomputed where they are needed instead of being handed from launch to launch: every front
// workgroup (one clip x 16 channels) re-derives the LayerNorm mean / rstd of its clip's rows from the L2-resident
// (T x C) slab (74 KB at T=100, C=368), every MLP workgroup re-derives the GroupNorm statistics of the clips its rows
// belong to.  That costs a couple of microseconds of L2 reads per workgroup and removes the LayerNorm / GroupNorm
// launches and their round trips through HBM.
#include "common.h"
#include "sgp_tile.h"
#include <stdlib.h>

static long long* g_sgp_front_dbg = nullptr;           // diagnostic: per-workgroup phase time stamps (wall_clock64, 10 ns ticks)
extern "C" int tdeed_sgp_front_set_debug(void* buf) { g_sgp_front_dbg = (long long*)buf; return TDEED_OK; }
#define SF_STAMP(i) do { if (dbg && threadIdx.x == 0) dbg[(long)(blockIdx.y * gridDim.x + blockIdx.x) * 16 + (i)] = wall_clock64(); } while (0)

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
  // a row's parts are requested TOGETHER, in batches of PB (round 6: as `for (pt) { load; add; }` hipcc emitted
  // `global_load; s_waitcnt vmcnt(0)` per part -- 3 to 12 dependent L2 round trips in front of every front launch, and the
  // vmcnt(0) also drained the tile / weight loads issued before); the additions keep their order
  constexpr int PB = 6;
  for (int t = threadIdx.x; t < T_len; t += blockDim.x) {
    if (parts == 0) {
      const f32x2 v = *reinterpret_cast<const f32x2*>(rowstat + (long)t * 2);
      mu[t] = v[0];
      rs[t] = v[1];
    } else {
      float s = 0.f, q = 0.f;
      for (int p0 = 0; p0 < parts; p0 += PB) {
        f32x2 v[PB];
#pragma unroll
        for (int k = 0; k < PB; ++k) v[k] = *reinterpret_cast<const f32x2*>(rowstat + min(p0 + k, parts - 1) * pstride + (long)t * 2);
        TD_ISSUE_FENCE();
#pragma unroll
        for (int k = 0; k < PB; ++k)
          if (p0 + k < parts) {
            s += v[k][0];
            q += v[k][1];
          }
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
                                                        const float* __restrict__ rowstat, int rs_parts,
                                                        bf16_t* __restrict__ y16, long long* __restrict__ dbg) {
  extern __shared__ float sm[];
  SF_STAMP(0);
  const int halo = up >> 1;
  const int wlen = 2 * ks + up + 2;
  float* tile = sm;                                   // [(T+2h)][16]
  float* res = tile + (T_len + 2 * halo) * SGP_CH;    // [T][16]
  float* wl = res + T_len * SGP_CH;                   // [wlen][16]
  float* red = wl + wlen * SGP_CH;                    // [17][16]
  float* redq = red + 17 * SGP_CH;                    // [16][16] (a region of its own: the tile is only T + 2 halo rows)
  float* mu = redq + 16 * SGP_CH;                     // [T]
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
  SF_STAMP(1);
  tile_commit<T>(tv, T_len, c0, C, tile, halo, res);          // res starts as the raw rows: y = x + (branches) grows in place
  dw_commit(wv, wlen, c0, C, wl);
  __syncthreads();
  SF_STAMP(2);
  red[tl * SGP_CH + c] = ln_apply_tile(tile, T_len, halo, mu, rs, lw, lb, cok);      // + this thread's share of mean_T
  __syncthreads();
  SF_STAMP(3);
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
  SF_STAMP(4);
  store_tile<T>(res, y + base, C, 0, T_len, c0, C, (const T*)nullptr, 0);
  // fp32 residual stream over a WIDE feature dimension: a bf16 copy of y as the fc1 contraction's operand (its fp32 rows would
  // double the bytes every column tile of that launch pulls through its CU: 34.9 vs 28.1 us at C = 768, 1600 rows)
  if (y16) store_tile<bf16_t>(res, y16 + base, C, 0, T_len, c0, C, (const bf16_t*)nullptr, 0);
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
  SF_STAMP(5);
}

static size_t front_smem(int T_len, int ks, int up, int ntiles, int nres, int nstat) {
  const int halo = up / 2, wlen = 2 * ks + up + 2;
  return (size_t)(ntiles * (T_len + 2 * halo) * SGP_CH + nres * T_len * SGP_CH + ntiles * wlen * SGP_CH + 33 * SGP_CH +
                  nstat * 2 * T_len) * sizeof(float);
}

extern "C" int tdeed_sgp_front_fwd(const void* x, int B, int T, int C, int ks, int up, const float* ln_w,
                                   const float* ln_b, float eps, const float* dw, const float* db, void* y,
                                   float* chsum, const float* rowstat, int rowstat_parts, void* y16, int dtype,
                                   void* stream) {
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
                       dw, db, (float*)y, chsum, rowstat, rowstat_parts, (bf16_t*)y16, g_sgp_front_dbg);
  else if (dtype == TDEED_BF16)
    hipLaunchKernelGGL(sgp_front_kernel<bf16_t>, grid, dim3(256), smem, st, (const bf16_t*)x, T, C, ks, up, ln_w, ln_b,
                       eps, dw, db, (bf16_t*)y, chsum, rowstat, rowstat_parts, (bf16_t*)nullptr, g_sgp_front_dbg);
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
  float* red = wl + wlen * SGP_CH;             // [17][16] (+ [16][16] the block kernel uses: one layout, front_smem)
  float* mu = red + 33 * SGP_CH;               // [T_hi]
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
