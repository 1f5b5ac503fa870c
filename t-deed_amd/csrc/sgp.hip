// SGP encoder-decoder pieces (reference: /root/reference/model/modules.py:58-363), NTC layout.
// Everything here is bandwidth / latency bound: channel LayerNorm (wave64 reductions over the
// contiguous C of one row), the depthwise temporal convs with their +-up/2 window staged in LDS,
// GroupNorm(16) over a (C/16 x T) slab, adaptive max-pool, linear up-sampling.  The dense
// C->4C->C / 6C->C contractions run on the MFMA kernel in gemm.hip.
#include "common.h"

// =========================================================================== channel LayerNorm
// one wave per row; lane l owns 16-B chunks l, l+64, ... of the row (C <= 4*64*EPC).
template <typename T>
__global__ __launch_bounds__(256) void layernorm_kernel(const T* __restrict__ x, long ldx, int rows, int C,
                                                        const float* __restrict__ w,
                                                        const float* __restrict__ b, float eps,
                                                        T* __restrict__ y, long ldy) {
  constexpr int EPC = Chunk<T>::N;
  constexpr int MAXCH = 4;
  const int lane = threadIdx.x & 63;
  const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const int nch = C / EPC;
  float v[MAXCH][EPC];
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < MAXCH; ++i) {
    const int ck = lane + 64 * i;
    if (ck < nch) {
      Chunk<T>::load(x + row * ldx + (long)ck * EPC, v[i]);
#pragma unroll
      for (int e = 0; e < EPC; ++e) s += v[i][e];
    }
  }
  const float mu = wave_sum(s) / (float)C;
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < MAXCH; ++i) {
    const int ck = lane + 64 * i;
    if (ck < nch) {
#pragma unroll
      for (int e = 0; e < EPC; ++e) {
        v[i][e] -= mu;
        q += v[i][e] * v[i][e];
      }
    }
  }
  const float den = sqrtf(wave_sum(q) / (float)C + eps);
#pragma unroll
  for (int i = 0; i < MAXCH; ++i) {
    const int ck = lane + 64 * i;
    if (ck < nch) {
      float o[EPC];
#pragma unroll
      for (int e = 0; e < EPC; ++e) {
        const int c = ck * EPC + e;
        o[e] = v[i][e] / den * w[c] + b[c];
      }
      Chunk<T>::store(y + row * ldy + (long)ck * EPC, o);
    }
  }
}

extern "C" int tdeed_layernorm_fwd(const void* x, long ldx, int rows, int C, const float* w, const float* b,
                                   float eps, void* y, long ldy, int dtype, void* stream) {
  TD_CHECK(x && w && b && y, "layernorm: null pointer");
  TD_CHECK(rows > 0 && C > 0 && C % 8 == 0 && ldx % 8 == 0 && ldy % 8 == 0, "layernorm: bad sizes");
  hipStream_t st = (hipStream_t)stream;
  dim3 grid(cdiv(rows, 4));
  if (dtype == TDEED_F32) {
    TD_CHECK(C <= 4 * 64 * 4, "layernorm: C=%d too wide", C);
    hipLaunchKernelGGL(layernorm_kernel<float>, grid, dim3(256), 0, st, (const float*)x, ldx, rows, C, w, b, eps,
                       (float*)y, ldy);
  } else if (dtype == TDEED_BF16) {
    TD_CHECK(C <= 4 * 64 * 8, "layernorm: C=%d too wide", C);
    hipLaunchKernelGGL(layernorm_kernel<bf16_t>, grid, dim3(256), 0, st, (const bf16_t*)x, ldx, rows, C, w, b, eps,
                       (bf16_t*)y, ldy);
  } else { tdeed_set_error("layernorm: bad dtype %d", dtype); return TDEED_ERR_ARG; }
  TD_LAUNCH_CHECK("layernorm");
  return TDEED_OK;
}

#include "sgp_tile.h"

// =========================================================================== SGPBlock front half
template <typename T>
__global__ __launch_bounds__(256) void sgp_branch_kernel(const T* __restrict__ o, const T* __restrict__ x,
                                                         int T_len, int C, int ks, int up,
                                                         const float* __restrict__ dw,
                                                         const float* __restrict__ db, T* __restrict__ y) {
  extern __shared__ float sm[];
  const int halo = up >> 1;
  const int wlen = 2 * ks + up + 2;
  float* tile = sm;                                   // [(T+2h)][16]
  float* res = tile + (T_len + 2 * halo) * SGP_CH;    // [T][16]
  float* wl = res + T_len * SGP_CH;                   // [wlen][16]
  float* red = wl + wlen * SGP_CH;                    // [17][16]
  const int b = blockIdx.x, c0 = blockIdx.y * SGP_CH;
  const long base = (long)b * T_len * C;
  const int c = threadIdx.x & 15, tl = threadIdx.x >> 4;
  const bool cok = c0 + c < C;
  float tv[SGP_TI][Chunk<T>::N], wv[SGP_WI];
  tile_issue<T>(o + base, C, T_len, c0, C, tv);
  dw_issue(dw, wlen, c0, C, wv);
  const Bias5 bb = bias_issue(db, C, c0 + c, C);
  tile_commit<T>(tv, T_len, c0, C, tile, halo);
  dw_commit(wv, wlen, c0, C, wl);
  __syncthreads();
  tile_mean(tile, T_len, halo, red);
  const float mean_c = red[16 * SGP_CH + c];
  // blockIdx.z owns a slice of the time axis (the whole tile is staged by every slice: the mean needs all of T)
  const int tper = (T_len + gridDim.z - 1) / gridDim.z;
  const int t_begin = blockIdx.z * tper, t_end = min(T_len, t_begin + tper);
  for (int t = t_begin + tl; t < t_end; t += 16) {
    BranchOut r = branch_eval(tile, wl, bb, cok, t, c, halo, ks, up, mean_c);
    res[t * SGP_CH + c] = r.inst + r.conv_gate + tile[(halo + t) * SGP_CH + c];
  }
  __syncthreads();
  store_tile<T>(res, y + base, C, t_begin, t_end, c0, C, x + base, C);
}

static size_t sgp_smem(int T_len, int ks, int up, int ntiles, int nres) {
  const int halo = up / 2, wlen = 2 * ks + up + 2;
  return (size_t)(ntiles * (T_len + 2 * halo) * SGP_CH + nres * T_len * SGP_CH + ntiles * wlen * SGP_CH +
                  17 * SGP_CH) * sizeof(float);
}

extern "C" int tdeed_sgp_branch_fwd(const void* o, const void* x, int B, int T, int C, int ks, int up,
                                    const float* dw, const float* db, void* y, int dtype, void* stream) {
  TD_CHECK(o && x && dw && db && y, "sgp_branch: null pointer");
  TD_CHECK(B > 0 && T > 0 && C % 8 == 0 && ks % 2 == 1 && up % 2 == 1 && up >= ks, "sgp_branch: bad sizes");
  TD_CHECK(T <= (dtype == TDEED_BF16 ? 512 : 256) && 2 * ks + up + 2 <= 80, "sgp_branch: T=%d / ks=%d up=%d beyond the staging registers", T, ks, up);
  size_t smem = sgp_smem(T, ks, up, 1, 1);
  TD_CHECK(smem <= 64 * 1024, "sgp_branch: T=%d too long for the LDS window", T);
  dim3 grid(B, cdiv(C, SGP_CH));
  hipStream_t st = (hipStream_t)stream;
  if (dtype == TDEED_F32)
    hipLaunchKernelGGL(sgp_branch_kernel<float>, grid, dim3(256), smem, st, (const float*)o, (const float*)x, T, C,
                       ks, up, dw, db, (float*)y);
  else if (dtype == TDEED_BF16)
    hipLaunchKernelGGL(sgp_branch_kernel<bf16_t>, grid, dim3(256), smem, st, (const bf16_t*)o, (const bf16_t*)x, T,
                       C, ks, up, dw, db, (bf16_t*)y);
  else { tdeed_set_error("sgp_branch: bad dtype %d", dtype); return TDEED_ERR_ARG; }
  TD_LAUNCH_CHECK("sgp_branch");
  return TDEED_OK;
}

// =========================================================================== SGPMixer front half
// cat row = [out1 | out2 | out3 | out4 | zn | xu], each C wide.  zn is already in slab 4.
template <typename T>
__global__ __launch_bounds__(256) void mixer_branch_kernel(const T* __restrict__ xn, int T_hi, int T_lo, int C,
                                                           int ks, int up, const float* __restrict__ dw1,
                                                           const float* __restrict__ db1,
                                                           const float* __restrict__ dw2,
                                                           const float* __restrict__ db2, T* __restrict__ cat) {
  extern __shared__ float sm[];
  const int halo = up >> 1;
  const int wlen = 2 * ks + up + 2;
  const int trows = T_hi + 2 * halo;
  float* zt = sm;                              // zn tile
  float* xt = zt + trows * SGP_CH;             // xu tile
  float* res = xt + trows * SGP_CH;            // [T_hi][16] (also holds xn [T_lo][16] during upsampling)
  float* wl1 = res + T_hi * SGP_CH;
  float* wl2 = wl1 + wlen * SGP_CH;
  float* red = wl2 + wlen * SGP_CH;
  const int b = blockIdx.x, c0 = blockIdx.y * SGP_CH;
  const long ldc = 6L * C;
  T* crow = cat + (long)b * T_hi * ldc;
  const int c = threadIdx.x & 15, tl = threadIdx.x >> 4;
  const bool cok = c0 + c < C;
  {
    float zv[SGP_TI][Chunk<T>::N], xv[SGP_TI][Chunk<T>::N], w1v[SGP_WI], w2v[SGP_WI];
    tile_issue<T>(crow + 4L * C, ldc, T_hi, c0, C, zv);
    tile_issue<T>(xn + (long)b * T_lo * C, C, T_lo, c0, C, xv);
    dw_issue(dw1, wlen, c0, C, w1v);
    dw_issue(dw2, wlen, c0, C, w2v);
    tile_commit<T>(zv, T_hi, c0, C, zt, halo);
    // xn at T_lo -> res (no halo), then linear up-sampling (align_corners=True) into xt
    tile_commit<T>(xv, T_lo, c0, C, res, 0);
    dw_commit(w1v, wlen, c0, C, wl1);
    dw_commit(w2v, wlen, c0, C, wl2);
  }
  const Bias5 bb1 = bias_issue(db1, C, c0 + c, C), bb2 = bias_issue(db2, C, c0 + c, C);
  for (int i = threadIdx.x; i < 2 * halo * SGP_CH; i += 256) {
    int r = i / SGP_CH, c = i - r * SGP_CH;
    int row = r < halo ? r : (T_hi + r);
    xt[row * SGP_CH + c] = 0.f;
  }
  __syncthreads();
  {
    const float scale = (T_hi > 1) ? (float)(T_lo - 1) / (float)(T_hi - 1) : 0.f;
    for (int t = tl; t < T_hi; t += 16) {
      float v;
      if (T_hi == T_lo) {
        v = res[t * SGP_CH + c];
      } else {
        const float src = scale * (float)t;
        const int i0 = (int)src;
        const int i1 = i0 + (i0 < T_lo - 1 ? 1 : 0);
        const float l1 = fminf(fmaxf(src - (float)i0, 0.f), 1.f);
        const float l0 = 1.f - l1;
        v = l0 * res[i0 * SGP_CH + c] + l1 * res[i1 * SGP_CH + c];
      }
      // the up-sampled sequence is a tensor in the reference: round it like a stored activation
      xt[(halo + t) * SGP_CH + c] = round_to<T>(v);
    }
  }
  __syncthreads();
  // blockIdx.z owns a slice of the time axis; the tiles (and the means over all of T) are staged by every slice
  const int tper = (T_hi + gridDim.z - 1) / gridDim.z;
  const int t_begin = blockIdx.z * tper, t_end = min(T_hi, t_begin + tper);
  // slab 5 = xu
  for (int t = t_begin + tl; t < t_end; t += 16) res[t * SGP_CH + c] = xt[(halo + t) * SGP_CH + c];
  __syncthreads();
  store_tile<T>(res, crow + 5L * C, ldc, t_begin, t_end, c0, C, (const T*)nullptr, 0);
  tile_mean(zt, T_hi, halo, red);
  const float mz = red[16 * SGP_CH + c];
  __syncthreads();
  tile_mean(xt, T_hi, halo, red);
  const float mx = red[16 * SGP_CH + c];
  // one sweep per source: out1 (slab 0) and out3 (slab 2) from z, then out2 (slab 1) and out4 (slab 3) from x;
  // each branch_eval yields both products, staged in two [T][16] tiles (res, res2) and stored as 16-byte chunks
  float* res2 = red + 17 * SGP_CH;
  for (int src = 0; src < 2; ++src) {
    const float* tile = src == 0 ? zt : xt;
    const float* wl = src == 0 ? wl1 : wl2;
    const Bias5 bb = src == 0 ? bb1 : bb2;
    const float mean_c = src == 0 ? mz : mx;
    __syncthreads();
    for (int t = t_begin + tl; t < t_end; t += 16) {
      BranchOut r = branch_eval(tile, wl, bb, cok, t, c, halo, ks, up, mean_c);
      res[t * SGP_CH + c] = r.conv_gate;
      res2[t * SGP_CH + c] = r.inst;
    }
    __syncthreads();
    store_tile<T>(res, crow + (long)src * C, ldc, t_begin, t_end, c0, C, (const T*)nullptr, 0);
    store_tile<T>(res2, crow + (long)(2 + src) * C, ldc, t_begin, t_end, c0, C, (const T*)nullptr, 0);
  }
}

extern "C" int tdeed_mixer_branch_fwd(const void* xn, int B, int T_hi, int T_lo, int C, int ks, int up,
                                      const float* dw1, const float* db1, const float* dw2, const float* db2,
                                      void* cat, int dtype, void* stream) {
  TD_CHECK(xn && dw1 && db1 && dw2 && db2 && cat, "mixer_branch: null pointer");
  TD_CHECK(B > 0 && T_hi >= T_lo && T_lo > 0 && C % 8 == 0 && ks % 2 == 1 && up % 2 == 1 && up >= ks,
           "mixer_branch: bad sizes");
  TD_CHECK(T_hi <= (dtype == TDEED_BF16 ? 512 : 256) && 2 * ks + up + 2 <= 80, "mixer_branch: T=%d / ks=%d up=%d beyond the staging registers", T_hi, ks, up);
  size_t smem = sgp_smem(T_hi, ks, up, 2, 2);
  TD_CHECK(smem <= 128 * 1024, "mixer_branch: T=%d too long for the LDS window", T_hi);
  static TdDevOnce attr_set;
  if (!attr_set.get()) {        // long clips (T=250) need more than the default 64 KB of dynamic LDS
    hipError_t e = hipFuncSetAttribute((const void*)mixer_branch_kernel<float>, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
    if (e == hipSuccess) e = hipFuncSetAttribute((const void*)mixer_branch_kernel<bf16_t>, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
    if (e != hipSuccess) { tdeed_set_error("mixer_branch: hipFuncSetAttribute: %s", hipGetErrorString(e)); return TDEED_ERR_RUNTIME; }
    attr_set.set();
  }
  dim3 grid(B, cdiv(C, SGP_CH));
  hipStream_t st = (hipStream_t)stream;
  if (dtype == TDEED_F32)
    hipLaunchKernelGGL(mixer_branch_kernel<float>, grid, dim3(256), smem, st, (const float*)xn, T_hi, T_lo, C, ks,
                       up, dw1, db1, dw2, db2, (float*)cat);
  else if (dtype == TDEED_BF16)
    hipLaunchKernelGGL(mixer_branch_kernel<bf16_t>, grid, dim3(256), smem, st, (const bf16_t*)xn, T_hi, T_lo, C, ks,
                       up, dw1, db1, dw2, db2, (bf16_t*)cat);
  else { tdeed_set_error("mixer_branch: bad dtype %d", dtype); return TDEED_ERR_ARG; }
  TD_LAUNCH_CHECK("mixer_branch");
  return TDEED_OK;
}

// =========================================================================== GroupNorm
template <typename T>
__global__ __launch_bounds__(256) void groupnorm_kernel(const T* __restrict__ x, int T_len, int C, int G,
                                                        const float* __restrict__ w,
                                                        const float* __restrict__ bia, float eps,
                                                        T* __restrict__ y) {
  extern __shared__ float sm[];     // [T][cg] cached slab + 4 scratch
  const int b = blockIdx.x, g = blockIdx.y;
  const int cg = C / G;
  const int n = T_len * cg;
  float* slab = sm;
  float* scratch = sm + n;
  const T* xb = x + (long)b * T_len * C + g * cg;
  float s = 0.f;
  for (int i = threadIdx.x; i < n; i += 256) {
    int t = i / cg, cl = i - t * cg;
    float v = (float)xb[(long)t * C + cl];
    slab[i] = v;
    s += v;
  }
  const float mean = block_sum<4>(s, scratch) / (float)n;
  float q = 0.f;
  for (int i = threadIdx.x; i < n; i += 256) {
    float d = slab[i] - mean;
    q += d * d;
  }
  const float var = block_sum<4>(q, scratch) / (float)n;
  const float rstd = 1.0f / sqrtf(var + eps);
  T* yb = y + (long)b * T_len * C + g * cg;
  for (int i = threadIdx.x; i < n; i += 256) {
    int t = i / cg, cl = i - t * cg;
    int c = g * cg + cl;
    yb[(long)t * C + cl] = (T)((slab[i] - mean) * rstd * w[c] + bia[c]);
  }
}

extern "C" int tdeed_groupnorm_fwd(const void* x, int B, int T, int C, int G, const float* w, const float* b,
                                   float eps, void* y, int dtype, void* stream) {
  TD_CHECK(x && w && b && y, "groupnorm: null pointer");
  TD_CHECK(B > 0 && T > 0 && G > 0 && C % G == 0, "groupnorm: bad sizes");
  size_t smem = ((size_t)T * (C / G) + 8) * sizeof(float);
  TD_CHECK(smem <= 64 * 1024, "groupnorm: slab too large");
  dim3 grid(B, G);
  hipStream_t st = (hipStream_t)stream;
  if (dtype == TDEED_F32)
    hipLaunchKernelGGL(groupnorm_kernel<float>, grid, dim3(256), smem, st, (const float*)x, T, C, G, w, b, eps,
                       (float*)y);
  else if (dtype == TDEED_BF16)
    hipLaunchKernelGGL(groupnorm_kernel<bf16_t>, grid, dim3(256), smem, st, (const bf16_t*)x, T, C, G, w, b, eps,
                       (bf16_t*)y);
  else { tdeed_set_error("groupnorm: bad dtype %d", dtype); return TDEED_ERR_ARG; }
  TD_LAUNCH_CHECK("groupnorm");
  return TDEED_OK;
}

// =========================================================================== adaptive max-pool over T
template <typename T>
__global__ void maxpool_kernel(const T* __restrict__ x, int T_in, int T_out, int C, T* __restrict__ y,
                               long total) {
  constexpr int EPC = Chunk<T>::N;
  const int cpr = C / EPC;
  for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
    const int ck = (int)(idx % cpr);
    const long r = idx / cpr;
    const int i = (int)(r % T_out);
    const long b = r / T_out;
    const int lo = (int)(((long)i * T_in) / T_out);
    const int hi = (int)((((long)(i + 1)) * T_in + T_out - 1) / T_out);
    float m[EPC];
    Chunk<T>::load(x + ((long)b * T_in + lo) * C + ck * EPC, m);
    for (int t = lo + 1; t < hi; ++t) {
      float v[EPC];
      Chunk<T>::load(x + ((long)b * T_in + t) * C + ck * EPC, v);
#pragma unroll
      for (int e = 0; e < EPC; ++e) m[e] = fmaxf(m[e], v[e]);
    }
    Chunk<T>::store(y + ((long)b * T_out + i) * C + ck * EPC, m);
  }
}

// The same pooling by one wave per POOLED row, which therefore also holds the whole row: it leaves the LayerNorm statistics
// (mean, rstd over C of the stored values; modules.py:353-357) of every pooled row for the next block's front kernel.  Used
// behind encoder blocks whose length does not halve (25 -> 13, 125 -> 63): those cannot pool inside the fc2 launch, and without
// the statistics the front kernel re-derives them from the clip's slab (16 vs 9 us at C = 768).
template <typename T>
__global__ __launch_bounds__(256) void maxpool_rowstat_kernel(const T* __restrict__ x, int T_in, int T_out, int C,
                                                              T* __restrict__ y, float* __restrict__ rowstat, float eps,
                                                              long rows) {
  constexpr int EPC = Chunk<T>::N;
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const long r = (long)blockIdx.x * 4 + wid;
  if (r >= rows) return;
  const int i = (int)(r % T_out);
  const long b = r / T_out;
  const int lo = (int)(((long)i * T_in) / T_out);
  const int hi = (int)((((long)(i + 1)) * T_in + T_out - 1) / T_out);
  const int cpr = C / EPC;
  float s1 = 0.f, s2 = 0.f;
  for (int ck = lane; ck < cpr; ck += 64) {
    float m[EPC];
    Chunk<T>::load(x + ((long)b * T_in + lo) * C + ck * EPC, m);
    for (int t = lo + 1; t < hi; ++t) {
      float v[EPC];
      Chunk<T>::load(x + ((long)b * T_in + t) * C + ck * EPC, v);
#pragma unroll
      for (int e = 0; e < EPC; ++e) m[e] = fmaxf(m[e], v[e]);
    }
    Chunk<T>::store(y + r * C + ck * EPC, m);
#pragma unroll
    for (int e = 0; e < EPC; ++e) {
      s1 += m[e];                       // (a maximum of stored values is a stored value: no rounding to repeat)
      s2 = fmaf(m[e], m[e], s2);
    }
  }
  s1 = wave_sum(s1);
  s2 = wave_sum(s2);
  if (lane == 0) {
    const float mu = s1 / (float)C;
    rowstat[r * 2] = mu;
    rowstat[r * 2 + 1] = 1.0f / sqrtf(fmaxf(s2 / (float)C - mu * mu, 0.f) + eps);
  }
}

extern "C" int tdeed_maxpool_rowstat_fwd(const void* x, int B, int T_in, int T_out, int C, void* y, float* rowstat, float eps,
                                         int dtype, void* stream) {
  TD_CHECK(x && y && rowstat, "maxpool_rowstat: null pointer");
  TD_CHECK(B > 0 && T_in > 0 && T_out > 0 && T_out <= T_in && C % 8 == 0, "maxpool_rowstat: bad sizes");
  hipStream_t st = (hipStream_t)stream;
  const long rows = (long)B * T_out;
  if (dtype == TDEED_F32)
    hipLaunchKernelGGL(maxpool_rowstat_kernel<float>, dim3(cdiv(rows, 4)), dim3(256), 0, st, (const float*)x, T_in, T_out, C,
                       (float*)y, rowstat, eps, rows);
  else if (dtype == TDEED_BF16)
    hipLaunchKernelGGL(maxpool_rowstat_kernel<bf16_t>, dim3(cdiv(rows, 4)), dim3(256), 0, st, (const bf16_t*)x, T_in, T_out,
                       C, (bf16_t*)y, rowstat, eps, rows);
  else { tdeed_set_error("maxpool_rowstat: bad dtype %d", dtype); return TDEED_ERR_ARG; }
  TD_LAUNCH_CHECK("maxpool_rowstat");
  return TDEED_OK;
}

extern "C" int tdeed_maxpool_fwd(const void* x, int B, int T_in, int T_out, int C, void* y, int dtype,
                                 void* stream) {
  TD_CHECK(x && y, "maxpool: null pointer");
  TD_CHECK(B > 0 && T_in > 0 && T_out > 0 && T_out <= T_in && C % 8 == 0, "maxpool: bad sizes");
  hipStream_t st = (hipStream_t)stream;
  if (dtype == TDEED_F32) {
    long total = (long)B * T_out * (C / 4);
    hipLaunchKernelGGL(maxpool_kernel<float>, dim3(cdiv(total, 256)), dim3(256), 0, st, (const float*)x, T_in,
                       T_out, C, (float*)y, total);
  } else if (dtype == TDEED_BF16) {
    long total = (long)B * T_out * (C / 8);
    hipLaunchKernelGGL(maxpool_kernel<bf16_t>, dim3(cdiv(total, 256)), dim3(256), 0, st, (const bf16_t*)x, T_in,
                       T_out, C, (bf16_t*)y, total);
  } else { tdeed_set_error("maxpool: bad dtype %d", dtype); return TDEED_ERR_ARG; }
  TD_LAUNCH_CHECK("maxpool");
  return TDEED_OK;
}
