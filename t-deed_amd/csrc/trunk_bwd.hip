// Train-mode pieces of the RegNetY trunk (timm Bottleneck / BatchNorm2d / SEModule as called from
// /root/reference/model/model.py:38-45,133-135): batch-statistics BatchNorm forward and backward on channels-last
// [M][C] maps, the SE excitation with its intermediates kept for the backward, and the small row-wise helpers
// around them.  Parameter gradients come out of ordered partial sums (no float atomics).
#include "common.h"
#include <cstdlib>

// =========================================================================== column statistics of an [M][C] map
// part[slab][0][c] = sum_m v(m,c), part[slab][1][c] = sum_m v(m,c) * u(m,c) over the slab's rows, where
//   mode 0 (BN forward):  v = z,             u = z                      -> sum, sum of squares
//   mode 1 (BN backward): v = g,             u = (z - mean) * rstd      -> sum g, sum g * xhat,
//                         g = dy masked by y > 0 when a ReLU follows the BN (relu = 1)
// lanes = (row lane, channel chunk); every lane keeps CS_B row loads (x up to 3 tensors) in flight.
constexpr int CS_B = 4;
template <typename T, int MODE>                                // 0: forward sums; 1: backward, mask from z (or none); 2: backward, mask from y
__global__ __launch_bounds__(256) void colstats_kernel(const T* __restrict__ z, const T* __restrict__ dy,
                                                       const T* __restrict__ y, long M, int C, int relu,
                                                       const float* __restrict__ mean, const float* __restrict__ rstd,
                                                       const float* __restrict__ fa, const float* __restrict__ fb,
                                                       long rows_per_slab, float* __restrict__ part) {
  constexpr int EPC = Chunk<T>::N;
  extern __shared__ float sred[];                              // [RL][2][C]
  const int nch = C / EPC;
  const int RL = 256 / nch > 0 ? 256 / nch : 1;
  const long m0 = (long)blockIdx.x * rows_per_slab, m1 = min(M, m0 + rows_per_slab);
  for (int ch = threadIdx.x % nch, rl = threadIdx.x / nch; rl < RL && ch < nch; ch += 256) {      // one pass if nch <= 256
    const int c0 = ch * EPC;
    // ReLU mask of the backward: from the stored activation y, or (no residual in front of the ReLU: y == nullptr)
    // recomputed from z through the forward affine, y > 0 <=> fa * z + fb > 0 -- one map less to read
    constexpr bool mode = MODE != 0;
    const bool zmask = MODE == 1 && relu;
    float s1[EPC], s2[EPC], mu[EPC], rs[EPC], ma[EPC], mb[EPC];
#pragma unroll
    for (int e = 0; e < EPC; ++e) {
      s1[e] = 0.f; s2[e] = 0.f;
      mu[e] = mode ? mean[c0 + e] : 0.f;
      rs[e] = mode ? rstd[c0 + e] : 0.f;
      ma[e] = zmask ? fa[c0 + e] : 0.f;
      mb[e] = zmask ? fb[c0 + e] : 1.f;                          // no ReLU: act = 1 > 0 keeps every gradient
    }
    for (long r0 = m0 + rl; r0 < m1; r0 += (long)RL * CS_B) {
      float zv[CS_B][EPC], gv[MODE ? CS_B : 1][EPC], yv[MODE == 2 ? CS_B : 1][EPC];
#pragma unroll
      for (int b = 0; b < CS_B; ++b) {
        const long r = min(r0 + (long)b * RL, m1 - 1);
        Chunk<T>::load(z + r * C + c0, zv[b]);
        if constexpr (MODE != 0) Chunk<T>::load(dy + r * C + c0, gv[b]);
        if constexpr (MODE == 2) Chunk<T>::load(y + r * C + c0, yv[b]);
      }
#pragma unroll
      for (int b = 0; b < CS_B; ++b)
        if (r0 + (long)b * RL < m1) {
#pragma unroll
          for (int e = 0; e < EPC; ++e) {
            if constexpr (MODE == 0) {
              s1[e] += zv[b][e];
              s2[e] = fmaf(zv[b][e], zv[b][e], s2[e]);
            } else {
              float act;
              if constexpr (MODE == 2) act = relu ? yv[b][e] : 1.f;
              else act = fmaf(zv[b][e], ma[e], mb[e]);
              const float g = act > 0.f ? gv[b][e] : 0.f;
              s1[e] += g;
              s2[e] = fmaf(g, zv[b][e] - mu[e], s2[e]);         // times rstd once, below
            }
          }
        }
    }
#pragma unroll
    for (int e = 0; e < EPC; ++e) {
      sred[(rl * 2 + 0) * C + c0 + e] = s1[e];
      sred[(rl * 2 + 1) * C + c0 + e] = mode ? s2[e] * rs[e] : s2[e];
    }
  }
  __syncthreads();
  for (int j = threadIdx.x; j < 2 * C; j += 256) {
    float v = 0.f;
    for (int rl = 0; rl < RL; ++rl) v += sred[rl * 2 * C + j];
    part[(long)blockIdx.x * 2 * C + j] = v;
  }
}

// sums [2][C] (folded partials) -> mean, rstd, the affine a = w * rstd, b = bias - mean * a, running statistics
// (torch BatchNorm: biased variance to normalise, unbiased M/(M-1) into running_var, momentum 0.1)
__global__ __launch_bounds__(256) void bn_finalize_kernel(const float* __restrict__ part_s, const float* __restrict__ part_q,
                                                          long pstride, int P, long M, int C,
                                                          const float* __restrict__ w, const float* __restrict__ bias,
                                                          float eps, float momentum, float* __restrict__ mean,
                                                          float* __restrict__ rstd, float* __restrict__ a,
                                                          float* __restrict__ b, float* __restrict__ run_mean,
                                                          float* __restrict__ run_var) {
  // one workgroup per 8 channels: lanes = (32 partial lanes, 8 channels), 8 row loads per lane in flight; the P slab
  // partials are folded in double (lane-strided, then a fixed-order chain)
  __shared__ double r1[32][9], r2[32][9];
  const int cl = threadIdx.x & 7, pl = threadIdx.x >> 3;
  const int c = blockIdx.x * 8 + cl;
  double s1 = 0.0, s2 = 0.0;
  for (int p0 = pl; p0 < P; p0 += 8 * 32) {
    float v1[8], v2[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const long p = min(p0 + u * 32, P - 1);
      v1[u] = part_s[p * pstride + c];
      v2[u] = part_q[p * pstride + c];
    }
#pragma unroll
    for (int u = 0; u < 8; ++u)
      if (p0 + u * 32 < P) {
        s1 += (double)v1[u];
        s2 += (double)v2[u];
      }
  }
  r1[pl][cl] = s1;
  r2[pl][cl] = s2;
  __syncthreads();
  if (pl != 0) return;
  s1 = 0.0;
  s2 = 0.0;
  for (int i = 0; i < 32; ++i) {
    s1 += r1[i][cl];
    s2 += r2[i][cl];
  }
  const double mu = s1 / (double)M;
  double var = s2 / (double)M - mu * mu;
  if (var < 0.0) var = 0.0;
  const float rs = (float)(1.0 / sqrt(var + (double)eps));
  mean[c] = (float)mu;
  rstd[c] = rs;
  a[c] = w[c] * rs;
  b[c] = bias[c] - (float)mu * w[c] * rs;
  if (run_mean) {
    const double unb = M > 1 ? var * (double)M / (double)(M - 1) : var;
    run_mean[c] = (1.f - momentum) * run_mean[c] + momentum * (float)mu;
    run_var[c] = (1.f - momentum) * run_var[c] + momentum * (float)unb;
  }
}

static int colstats_slabs(long M, long* rows_per_slab) {
  long slabs = (M + 63) / 64;                                  // >= 600 workgroups already for the 7x7 maps of a batch
  if (slabs > 2048) slabs = 2048;
  if (slabs < 1) slabs = 1;
  *rows_per_slab = (M + slabs - 1) / slabs;
  return (int)((M + *rows_per_slab - 1) / *rows_per_slab);
}

extern "C" int tdeed_bn_slabs(long M) { long r; return colstats_slabs(M, &r); }

template <typename T>
static int launch_colstats(const void* z, const void* dy, const void* y, long M, int C, int mode, int relu,
                           const float* mean, const float* rstd, const float* fa, const float* fb, float* part,
                           hipStream_t st) {
  long rps;
  const int slabs = colstats_slabs(M, &rps);
  const int nch = C / Chunk<T>::N;
  const int RL = 256 / nch > 0 ? 256 / nch : 1;
  const size_t sm = (size_t)RL * 2 * C * sizeof(float);
  if (mode == 0)
    hipLaunchKernelGGL((colstats_kernel<T, 0>), dim3(slabs), dim3(256), sm, st, (const T*)z, (const T*)dy, (const T*)y, M, C, relu,
                       mean, rstd, fa, fb, rps, part);
  else if (relu && y)
    hipLaunchKernelGGL((colstats_kernel<T, 2>), dim3(slabs), dim3(256), sm, st, (const T*)z, (const T*)dy, (const T*)y, M, C, relu,
                       mean, rstd, fa, fb, rps, part);
  else
    hipLaunchKernelGGL((colstats_kernel<T, 1>), dim3(slabs), dim3(256), sm, st, (const T*)z, (const T*)dy, (const T*)y, M, C, relu,
                       mean, rstd, fa, fb, rps, part);
  return slabs;
}

// BatchNorm (training) statistics of z [M][C]: mean/rstd [C] (kept for the backward), a/b [C] (the affine the apply
// pass uses), running statistics updated in place when given.  part: fp32 [tdeed_bn_slabs(M)][2][C].
extern "C" int tdeed_bn_train_stats(const void* z, long M, int C, const float* w, const float* bias, float eps,
                                    float momentum, float* part, float* mean, float* rstd, float* a, float* b,
                                    float* run_mean, float* run_var, int dtype, void* stream) {
  TD_CHECK(z && w && bias && part && mean && rstd && a && b, "bn_train_stats: null pointer");
  TD_CHECK(M > 0 && C > 0 && C % 8 == 0 && C <= 2048, "bn_train_stats: bad sizes");
  hipStream_t st = (hipStream_t)stream;
  int slabs;
  if (dtype == TDEED_F32) slabs = launch_colstats<float>(z, nullptr, nullptr, M, C, 0, 0, nullptr, nullptr, nullptr, nullptr, part, st);
  else if (dtype == TDEED_BF16) slabs = launch_colstats<bf16_t>(z, nullptr, nullptr, M, C, 0, 0, nullptr, nullptr, nullptr, nullptr, part, st);
  else { tdeed_set_error("bn_train_stats: bad dtype %d", dtype); return TDEED_ERR_ARG; }
  TD_LAUNCH_CHECK("bn colstats");
  hipLaunchKernelGGL(bn_finalize_kernel, dim3(C / 8), dim3(256), 0, st, part, part + C, 2L * C, slabs, M, C, w, bias, eps,
                     momentum, mean, rstd, a, b, run_mean, run_var);
  TD_LAUNCH_CHECK("bn_finalize");
  return TDEED_OK;
}

// The same finalisation from partial sums a producer wrote in its own epilogue: part_s / part_q hold P rows of per-channel
// sums / sums of squares, `pstride` floats apart (tdeed_gemm_fwd's colpart: part_q = part_s + N, pstride = 2N;
// tdeed_gconv3x3_fwd's pooled / pooled_sq: pstride = C).  M = rows the sums cover.
extern "C" int tdeed_bn_finalize(const float* part_s, const float* part_q, long pstride, int P, long M, int C,
                                 const float* w, const float* bias, float eps, float momentum, float* mean, float* rstd,
                                 float* a, float* b, float* run_mean, float* run_var, void* stream) {
  TD_CHECK(part_s && part_q && w && bias && mean && rstd && a && b, "bn_finalize: null pointer");
  TD_CHECK(P > 0 && M > 0 && C > 0 && C % 8 == 0 && pstride >= C, "bn_finalize: bad sizes");
  hipLaunchKernelGGL(bn_finalize_kernel, dim3(C / 8), dim3(256), 0, (hipStream_t)stream, part_s, part_q, pstride, P, M, C, w,
                     bias, eps, momentum, mean, rstd, a, b, run_mean, run_var);
  TD_LAUNCH_CHECK("bn_finalize");
  return TDEED_OK;
}

// out[s][j] = sum over the rows p of slice s of part[p * pstride + j], j < n: a first, wide fold for producers that leave
// one partial row per (frame, band) -- tens of thousands of rows for a few dozen channels -- so that the finalisation
// above (one workgroup per 8 channels) is left with `slices` rows.  Ordered, no atomics.
__global__ __launch_bounds__(256) void fold_rows_kernel(const float* __restrict__ part, long pstride, int P, int n,
                                                        int rows_per_slice, float* __restrict__ out, long ostride) {
  __shared__ float red[32][9];
  const int cl = threadIdx.x & 7, pl = threadIdx.x >> 3;
  const int j = blockIdx.x * 8 + cl, jj = min(j, n - 1);
  const int p_lo = blockIdx.y * rows_per_slice, p_hi = min(P, p_lo + rows_per_slice);
  float s = 0.f;
  for (int p0 = p_lo + pl; p0 < p_hi; p0 += 8 * 32) {
    float v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = part[(long)min(p0 + u * 32, p_hi - 1) * pstride + jj];
#pragma unroll
    for (int u = 0; u < 8; ++u) s += p0 + u * 32 < p_hi ? v[u] : 0.f;
  }
  red[pl][cl] = s;
  __syncthreads();
  if (pl == 0 && j < n) {
    float a = 0.f;
#pragma unroll
    for (int i = 0; i < 32; ++i) a += red[i][cl];
    out[(long)blockIdx.y * ostride + j] = a;
  }
}

extern "C" int tdeed_fold_rows(const float* part, long pstride, int P, int n, int slices, float* out, long ostride,
                               void* stream) {
  TD_CHECK(part && out && P > 0 && n > 0 && slices > 0 && slices <= 65535 && pstride >= n && ostride >= n, "fold_rows: bad arguments");
  const int rps = (P + slices - 1) / slices;
  hipLaunchKernelGGL(fold_rows_kernel, dim3((unsigned)((n + 7) / 8), (unsigned)slices), dim3(256), 0, (hipStream_t)stream, part,
                     pstride, P, n, rps, out, ostride);
  TD_LAUNCH_CHECK("fold_rows");
  return TDEED_OK;
}

// =========================================================================== per-channel affine (+ residual, ReLU)
// y = act(z * a[c] + b[c] + res)
// Thread map of the row-wise kernels below: thread = (row lane rl, 16-byte channel chunk ck), both fixed for the thread's
// life, so the per-channel constants are loaded once and no per-element index division is needed; a workgroup walks
// ROWS_PER_WG rows with RW_U row loads per lane in flight.
constexpr int RW_U = 4;
struct RowMap {
  int ck, rl, RL;
  bool on;
  __device__ __forceinline__ explicit RowMap(int nch) {
    RL = 256 / nch;
    rl = threadIdx.x / nch;
    ck = threadIdx.x - rl * nch;
    on = rl < RL;
  }
};
static inline int rows_per_wg(int nch) { return (256 / nch) * RW_U * 4; }

template <typename T>
__global__ __launch_bounds__(256) void affine_kernel(const T* __restrict__ z, const float* __restrict__ a,
                                                     const float* __restrict__ b, const T* __restrict__ res,
                                                     const float* __restrict__ ra, const float* __restrict__ rb, int relu,
                                                     T* __restrict__ y, long M, int nch, int rpw, T* __restrict__ y2 = nullptr,
                                                     int F2 = 0, int Fp2 = 0) {
  constexpr int EPC = Chunk<T>::N;
  const RowMap mp(nch);
  if (!mp.on) return;
  const int C = nch * EPC, c0 = mp.ck * EPC;
  // ra / rb: the residual is itself a raw conv output (the shortcut conv) whose BatchNorm affine is applied here: the
  // normalised shortcut map is not materialised (rounded to T like the map would be: same values as the two-pass form)
  const bool raff = ra != nullptr;
  const float* pra = raff ? ra : a;
  const float* prb = raff ? rb : a;
  float av[EPC], bv[EPC], rav[EPC], rbv[EPC];
#pragma unroll
  for (int e = 0; e < EPC; ++e) {
    av[e] = a[c0 + e];
    bv[e] = b[c0 + e];
    rav[e] = pra[c0 + e];
    rbv[e] = prb[c0 + e];
  }
  const long m0 = (long)blockIdx.x * rpw, m1 = min(M, m0 + rpw);
  for (long r0 = m0 + mp.rl; r0 < m1; r0 += (long)mp.RL * RW_U) {
    float v[RW_U][EPC], rv[RW_U][EPC];
#pragma unroll
    for (int u = 0; u < RW_U; ++u) {
      const long r = min(r0 + (long)u * mp.RL, m1 - 1);
      Chunk<T>::load(z + r * C + c0, v[u]);
      if (res) Chunk<T>::load(res + r * C + c0, rv[u]);
    }
#pragma unroll
    for (int u = 0; u < RW_U; ++u) {
      const long r = r0 + (long)u * mp.RL;
      if (r < m1) {
#pragma unroll
        for (int e = 0; e < EPC; ++e) {
          float o = fmaf(v[u][e], av[e], bv[e]);
          if (res) o += raff ? round_to<T>(fmaf(rv[u][e], rav[e], rbv[e])) : rv[u][e];
          v[u][e] = relu ? fmaxf(o, 0.f) : o;
        }
        Chunk<T>::store(y + r * C + c0, v[u]);
        if (y2 && c0 < Fp2) {               // compact copy of channels [0, F2) (zeros up to Fp2): the next block's gate-shift slice
#pragma unroll
          for (int e = 0; e < EPC; ++e)
            if (c0 + e >= F2) v[u][e] = 0.f;
          Chunk<T>::store(y2 + r * Fp2 + c0, v[u]);
        }
      }
    }
  }
}

static int bn_apply_launch(const void* z, long M, int C, const float* a, const float* b, const void* res, const float* ra,
                           const float* rb, int relu, void* y, int dtype, void* stream, void* y2 = nullptr, int F2 = 0, int Fp2 = 0);
// the same (ra / rb may be NULL) + y2 [M][Fp2]: a compact copy of output channels [0, F2), zeros in [F2, Fp2): the dense slice
// the NEXT block's gate-shift module normalises and shifts (tdeed_gsf_slice's result, without its pass over the map)
extern "C" int tdeed_bn_apply_slice(const void* z, long M, int C, const float* a, const float* b, const void* res, const float* ra,
                                    const float* rb, int relu, void* y, void* y2, int F2, int Fp2, int dtype, void* stream) {
  TD_CHECK(y2 && F2 > 0 && Fp2 >= F2 && Fp2 % 8 == 0 && Fp2 <= C && !ra == !rb, "bn_apply_slice: bad slice arguments");
  return bn_apply_launch(z, M, C, a, b, res, ra, rb, relu, y, dtype, stream, y2, F2, Fp2);
}
extern "C" int tdeed_bn_apply(const void* z, long M, int C, const float* a, const float* b, const void* res, int relu,
                              void* y, int dtype, void* stream) {
  return bn_apply_launch(z, M, C, a, b, res, nullptr, nullptr, relu, y, dtype, stream);
}
// y = act(z * a + b + (res * ra + rb)): conv3's BatchNorm + the shortcut conv's BatchNorm + ReLU of a downsampling bottleneck in
// one pass (timm Bottleneck.forward: x = act3(bn3(conv3(x)) + downsample(shortcut)))
extern "C" int tdeed_bn_apply2(const void* z, long M, int C, const float* a, const float* b, const void* res, const float* ra,
                               const float* rb, int relu, void* y, int dtype, void* stream) {
  TD_CHECK(res && ra && rb, "bn_apply2: the residual and its affine are required");
  return bn_apply_launch(z, M, C, a, b, res, ra, rb, relu, y, dtype, stream);
}
static int bn_apply_launch(const void* z, long M, int C, const float* a, const float* b, const void* res, const float* ra,
                           const float* rb, int relu, void* y, int dtype, void* stream, void* y2, int F2, int Fp2) {
  TD_CHECK(z && a && b && y && M > 0 && C > 0 && C % 8 == 0, "bn_apply: bad arguments");
  hipStream_t st = (hipStream_t)stream;
  TD_CHECK(dtype == TDEED_F32 || dtype == TDEED_BF16, "bn_apply: bad dtype %d", dtype);
  const int nch = C / (dtype == TDEED_F32 ? 4 : 8);
  TD_CHECK(nch <= 256, "bn_apply: C=%d too wide", C);
  const int rpw = rows_per_wg(nch);
  const long nwg = (M + rpw - 1) / rpw;
  TD_CHECK(nwg < 0x7fffffffL, "bn_apply: too many rows");
  if (dtype == TDEED_F32)
    hipLaunchKernelGGL(affine_kernel<float>, dim3((unsigned)nwg), dim3(256), 0, st, (const float*)z, a, b, (const float*)res,
                       ra, rb, relu, (float*)y, M, nch, rpw, (float*)y2, F2, Fp2);
  else
    hipLaunchKernelGGL(affine_kernel<bf16_t>, dim3((unsigned)nwg), dim3(256), 0, st, (const bf16_t*)z, a, b,
                       (const bf16_t*)res, ra, rb, relu, (bf16_t*)y, M, nch, rpw, (bf16_t*)y2, F2, Fp2);
  TD_LAUNCH_CHECK("bn_apply");
  return TDEED_OK;
}

// =========================================================================== BatchNorm (training) backward
// g = dy (masked by y > 0 if a ReLU followed), xhat = (z - mean) * rstd:
//   dz = w * rstd * (g - sum(g)/M - xhat * sum(g * xhat)/M),   dw = sum(g * xhat),   db = sum(g)
// d_res (optional) = g: the gradient of the residual that was added before the ReLU.
// with k1 = w * rstd, k2 = -k1 * rstd * sum(g xhat) / M, k3 = k1 * (mean * rstd * sum(g xhat) - sum(g)) / M (per channel,
// computed once per thread):  dz = k1 * g + k2 * z + k3
template <typename T>
__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(const T* __restrict__ z, const T* __restrict__ dy,
                                                           const T* __restrict__ y, int relu,
                                                           const float* __restrict__ mean, const float* __restrict__ rstd,
                                                           const float* __restrict__ w, const float* __restrict__ sums,
                                                           const float* __restrict__ fa, const float* __restrict__ fb,
                                                           float inv_M, T* __restrict__ dz, T* __restrict__ d_res,
                                                           long M, int nch, int rpw) {
  constexpr int EPC = Chunk<T>::N;
  const RowMap mp(nch);
  if (!mp.on) return;
  const int C = nch * EPC, c0 = mp.ck * EPC;
  const bool zmask = relu && !y;
  // the forward affine of the z-mask is read unconditionally (valid dummy when unused): no load under a select
  const float* pa = zmask ? fa : mean;
  const float* pb = zmask ? fb : mean;
  float k1[EPC], k2[EPC], k3[EPC], av[EPC], bv[EPC];
#pragma unroll
  for (int e = 0; e < EPC; ++e) {
    const int c = c0 + e;
    const float rs = rstd[c], mu = mean[c], s1 = sums[c] * inv_M, s2 = sums[C + c] * inv_M;
    k1[e] = w[c] * rs;
    k2[e] = -k1[e] * rs * s2;
    k3[e] = k1[e] * (mu * rs * s2 - s1);
    av[e] = pa[c];
    bv[e] = pb[c];
  }
  const long m0 = (long)blockIdx.x * rpw, m1 = min(M, m0 + rpw);
  for (long r0 = m0 + mp.rl; r0 < m1; r0 += (long)mp.RL * RW_U) {
    float zv[RW_U][EPC], gv[RW_U][EPC], yv[RW_U][EPC];
#pragma unroll
    for (int u = 0; u < RW_U; ++u) {
      const long r = min(r0 + (long)u * mp.RL, m1 - 1);
      Chunk<T>::load(z + r * C + c0, zv[u]);
      Chunk<T>::load(dy + r * C + c0, gv[u]);
      if (relu && !zmask) Chunk<T>::load(y + r * C + c0, yv[u]);
    }
#pragma unroll
    for (int u = 0; u < RW_U; ++u) {
      const long r = r0 + (long)u * mp.RL;
      if (r < m1) {
#pragma unroll
        for (int e = 0; e < EPC; ++e) {
          const float act = zmask ? fmaf(zv[u][e], av[e], bv[e]) : yv[u][e];
          const float g = (relu && !(act > 0.f)) ? 0.f : gv[u][e];
          gv[u][e] = g;
          zv[u][e] = fmaf(k1[e], g, fmaf(k2[e], zv[u][e], k3[e]));
        }
        Chunk<T>::store(dz + r * C + c0, zv[u]);
        if (d_res) Chunk<T>::store(d_res + r * C + c0, gv[u]);
      }
    }
  }
}

// part: fp32 [tdeed_bn_slabs(M)][2][C]; sums: fp32 [2][C] scratch; dw, db: fp32 [C]
extern "C" int tdeed_bn_train_bwd(const void* z, const void* dy, const void* y, int relu, long M, int C,
                                  const float* mean, const float* rstd, const float* w, const float* fa, const float* fb,
                                  float* part, float* sums, void* dz, void* d_res, float* dw, float* db, int dtype,
                                  void* stream) {
  TD_CHECK(z && dy && (!relu || y || (fa && fb)) && mean && rstd && w && part && sums && dz, "bn_train_bwd: null pointer");
  TD_CHECK(M > 0 && C > 0 && C % 8 == 0 && C <= 2048, "bn_train_bwd: bad sizes");
  hipStream_t st = (hipStream_t)stream;
  int slabs;
  if (dtype == TDEED_F32) slabs = launch_colstats<float>(z, dy, y, M, C, 1, relu, mean, rstd, fa, fb, part, st);
  else if (dtype == TDEED_BF16) slabs = launch_colstats<bf16_t>(z, dy, y, M, C, 1, relu, mean, rstd, fa, fb, part, st);
  else { tdeed_set_error("bn_train_bwd: bad dtype %d", dtype); return TDEED_ERR_ARG; }
  TD_LAUNCH_CHECK("bn bwd colstats");
  int rc = tdeed_reduce_partials(part, slabs, 2L * C, sums, 0, stream);
  if (rc != TDEED_OK) return rc;
  const float inv_M = 1.0f / (float)M;
  const int nch = C / (dtype == TDEED_F32 ? 4 : 8);
  TD_CHECK(nch <= 256, "bn_train_bwd: C=%d too wide", C);
  const int rpw = rows_per_wg(nch);
  const long nwg = (M + rpw - 1) / rpw;
  TD_CHECK(nwg < 0x7fffffffL, "bn_train_bwd: too many rows");
  if (dtype == TDEED_F32)
    hipLaunchKernelGGL(bn_bwd_apply_kernel<float>, dim3((unsigned)nwg), dim3(256), 0, st, (const float*)z, (const float*)dy,
                       (const float*)y, relu, mean, rstd, w, sums, fa, fb, inv_M, (float*)dz, (float*)d_res, M, nch, rpw);
  else
    hipLaunchKernelGGL(bn_bwd_apply_kernel<bf16_t>, dim3((unsigned)nwg), dim3(256), 0, st, (const bf16_t*)z, (const bf16_t*)dy,
                       (const bf16_t*)y, relu, mean, rstd, w, sums, fa, fb, inv_M, (bf16_t*)dz, (bf16_t*)d_res, M, nch, rpw);
  TD_LAUNCH_CHECK("bn_bwd_apply");
  // db = sums[0:C], dw = sums[C:2C]: callers that pass NULL read them straight out of `sums`
  if (db) {
    hipError_t e1 = hipMemcpyAsync(db, sums, (size_t)C * sizeof(float), hipMemcpyDeviceToDevice, st);
    if (e1 != hipSuccess) { tdeed_set_error("bn_train_bwd: copy failed"); return TDEED_ERR_RUNTIME; }
  }
  if (dw) {
    hipError_t e2 = hipMemcpyAsync(dw, sums + C, (size_t)C * sizeof(float), hipMemcpyDeviceToDevice, st);
    if (e2 != hipSuccess) { tdeed_set_error("bn_train_bwd: copy failed"); return TDEED_ERR_RUNTIME; }
  }
  return TDEED_OK;
}

// The same backward when the PRODUCER of dy has already left the masked column sums (tdeed_gconv3x3_dgrad_stats: conv2's
// input gradient arriving at conv1's BatchNorm + ReLU): part_s / part_q hold P rows (pstride floats apart) of per-channel
// sum g and sum g * (z - mean); they are folded here (x rstd for the second), then only the apply pass runs.
__global__ __launch_bounds__(256) void bn_sums_from_parts_kernel(const float* __restrict__ part_s, const float* __restrict__ part_q,
                                                                long pstride, int P, int C, const float* __restrict__ rstd,
                                                                float* __restrict__ sums) {
  __shared__ double r1[32][9], r2[32][9];
  const int cl = threadIdx.x & 7, pl = threadIdx.x >> 3;
  const int c = blockIdx.x * 8 + cl;
  double s1 = 0.0, s2 = 0.0;
  for (int p0 = pl; p0 < P; p0 += 8 * 32) {
    float v1[8], v2[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const long p = min(p0 + u * 32, P - 1);
      v1[u] = part_s[p * pstride + c];
      v2[u] = part_q[p * pstride + c];
    }
#pragma unroll
    for (int u = 0; u < 8; ++u)
      if (p0 + u * 32 < P) {
        s1 += (double)v1[u];
        s2 += (double)v2[u];
      }
  }
  r1[pl][cl] = s1;
  r2[pl][cl] = s2;
  __syncthreads();
  if (pl != 0) return;
  s1 = 0.0;
  s2 = 0.0;
  for (int i = 0; i < 32; ++i) {
    s1 += r1[i][cl];
    s2 += r2[i][cl];
  }
  sums[c] = (float)s1;
  sums[C + c] = (float)(s2 * (double)rstd[c]);
}

extern "C" int tdeed_bn_sums_from_parts(const float* part_s, const float* part_q, long pstride, int P, int C, const float* rstd,
                                        float* sums, void* stream) {
  TD_CHECK(part_s && part_q && rstd && sums && P > 0 && C > 0 && C % 8 == 0 && pstride >= C, "bn_sums_from_parts: bad arguments");
  hipLaunchKernelGGL(bn_sums_from_parts_kernel, dim3(C / 8), dim3(256), 0, (hipStream_t)stream, part_s, part_q, pstride, P, C,
                     rstd, sums);
  TD_LAUNCH_CHECK("bn_sums_from_parts");
  return TDEED_OK;
}

extern "C" int tdeed_bn_bwd_masked_from_parts(const void* z, const void* dy, long M, int C, const float* mean, const float* rstd,
                                              const float* w, const float* fa, const float* fb, const float* part_s,
                                              const float* part_q, long pstride, int P, float* sums, void* dz, int dtype,
                                              void* stream) {
  TD_CHECK(z && dy && mean && rstd && w && fa && fb && part_s && part_q && sums && dz, "bn_bwd_masked_from_parts: null pointer");
  TD_CHECK(M > 0 && C > 0 && C % 8 == 0 && P > 0 && pstride >= C, "bn_bwd_masked_from_parts: bad sizes");
  TD_CHECK(dtype == TDEED_F32 || dtype == TDEED_BF16, "bn_bwd_masked_from_parts: bad dtype %d", dtype);
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(bn_sums_from_parts_kernel, dim3(C / 8), dim3(256), 0, st, part_s, part_q, pstride, P, C, rstd, sums);
  TD_LAUNCH_CHECK("bn_sums_from_parts");
  const float inv_M = 1.0f / (float)M;
  const int nch = C / (dtype == TDEED_F32 ? 4 : 8);
  TD_CHECK(nch <= 256, "bn_bwd_masked_from_parts: C=%d too wide", C);
  const int rpw = rows_per_wg(nch);
  const long nwg = (M + rpw - 1) / rpw;
  TD_CHECK(nwg < 0x7fffffffL, "bn_bwd_masked_from_parts: too many rows");
  if (dtype == TDEED_F32)
    hipLaunchKernelGGL(bn_bwd_apply_kernel<float>, dim3((unsigned)nwg), dim3(256), 0, st, (const float*)z, (const float*)dy,
                       (const float*)nullptr, 1, mean, rstd, w, sums, fa, fb, inv_M, (float*)dz, (float*)nullptr, M, nch, rpw);
  else
    hipLaunchKernelGGL(bn_bwd_apply_kernel<bf16_t>, dim3((unsigned)nwg), dim3(256), 0, st, (const bf16_t*)z, (const bf16_t*)dy,
                       (const bf16_t*)nullptr, 1, mean, rstd, w, sums, fa, fb, inv_M, (bf16_t*)dz, (bf16_t*)nullptr, M, nch, rpw);
  TD_LAUNCH_CHECK("bn_bwd_apply");
  return TDEED_OK;
}

// =========================================================================== SE (training): squeeze, excitation, scale
// mean over the hw pixels of a frame: x [N][hw][C] -> p [N][C] fp32 (lanes = (pixel slice, channel chunk), batched loads)
// aff_on = 1 / 2: x / x2 is a raw conv output and relu(in_a[c] * . + in_b[c]) is applied on load (the post-BN map is not
// materialised in training)
template <typename T>
__global__ __launch_bounds__(256) void pool_mean_kernel(const T* __restrict__ x, const T* __restrict__ x2, int hw, int C,
                                                        const float* __restrict__ in_a, const float* __restrict__ in_b,
                                                        int aff_on, float* __restrict__ p) {
  constexpr int EPC = Chunk<T>::N;
  extern __shared__ float red[];       // [S][C]
  const long f = blockIdx.x;
  const int nch = C / EPC;
  const int S = 256 / nch > 0 ? 256 / nch : 1;
  for (int ch = threadIdx.x % nch, s = threadIdx.x / nch; s < S && ch < nch; ch += 256) {
    const int c0 = ch * EPC;
    float a[EPC], ia[EPC], ib[EPC];
    const float* pa = aff_on ? in_a : p;                         // unconditional loads (valid dummy when unused)
    const float* pb = aff_on ? in_b : p;
#pragma unroll
    for (int e = 0; e < EPC; ++e) {
      a[e] = 0.f;
      ia[e] = aff_on ? pa[c0 + e] : 1.f;
      ib[e] = aff_on ? pb[c0 + e] : 0.f;
    }
    const long base = f * hw * C + c0;
    for (int p0 = s; p0 < hw; p0 += S * 4) {
      float v[4][EPC], u[4][EPC];
#pragma unroll
      for (int b = 0; b < 4; ++b) {
        const long off = base + (long)min(p0 + b * S, hw - 1) * C;
        Chunk<T>::load(x + off, v[b]);
        if (x2) Chunk<T>::load(x2 + off, u[b]);
      }
#pragma unroll
      for (int b = 0; b < 4; ++b)
        if (p0 + b * S < hw) {
#pragma unroll
          for (int e = 0; e < EPC; ++e) {
            float xv = v[b][e], uv = x2 ? u[b][e] : 1.f;
            // rounded to T like the materialised map would be: bit-identical results with and without the map
            if (aff_on == 1) xv = (float)(T)fmaxf(fmaf(xv, ia[e], ib[e]), 0.f);
            if (aff_on == 2) uv = (float)(T)fmaxf(fmaf(uv, ia[e], ib[e]), 0.f);
            a[e] += x2 ? xv * uv : xv;
          }
        }
    }
#pragma unroll
    for (int e = 0; e < EPC; ++e) red[s * C + c0 + e] = a[e];
  }
  __syncthreads();
  const float inv = x2 ? 1.0f : 1.0f / (float)hw;              // squeeze: mean;  x2 given: plain sum of products
  for (int c = threadIdx.x; c < C; c += 256) {
    float v = 0.f;
    for (int s = 0; s < S; ++s) v += red[s * C + c];
    p[f * C + c] = v * inv;
  }
}

// p [N][C] = mean_px x  (x2 == NULL)   or   sum_px x * x2  (the gradient of the SE gate)
extern "C" int tdeed_pool_rows(const void* x, const void* x2, int N, int hw, int C, const float* in_a, const float* in_b,
                               int aff_on, float* p, int dtype, void* stream) {
  TD_CHECK(x && p && N > 0 && hw > 0 && C > 0 && C % 8 == 0 && C <= 2048, "pool_rows: bad arguments");
  TD_CHECK(aff_on == 0 || ((aff_on == 1 || (aff_on == 2 && x2)) && in_a && in_b), "pool_rows: bad on-load affine arguments");
  hipStream_t st = (hipStream_t)stream;
  if (dtype == TDEED_F32) {
    const int nch = C / 4, S = 256 / nch > 0 ? 256 / nch : 1;
    hipLaunchKernelGGL(pool_mean_kernel<float>, dim3(N), dim3(256), (size_t)S * C * sizeof(float), st, (const float*)x,
                       (const float*)x2, hw, C, in_a, in_b, aff_on, p);
  } else if (dtype == TDEED_BF16) {
    const int nch = C / 8, S = 256 / nch > 0 ? 256 / nch : 1;
    hipLaunchKernelGGL(pool_mean_kernel<bf16_t>, dim3(N), dim3(256), (size_t)S * C * sizeof(float), st, (const bf16_t*)x,
                       (const bf16_t*)x2, hw, C, in_a, in_b, aff_on, p);
  } else { tdeed_set_error("pool_rows: bad dtype %d", dtype); return TDEED_ERR_ARG; }
  TD_LAUNCH_CHECK("pool_rows");
  return TDEED_OK;
}

// hid = relu(W1 p + b1), gate = sigmoid(W2 hid + b2) per frame, hid kept for the backward.
// w1t [C][R], w2t [R][C] (transposed: adjacent lanes read adjacent addresses).  SEF frames share every weight load; a
// contraction whose output is narrower than the workgroup (R <= 128 hidden units) is split over 256 / RP slices of its
// input range (partial sums folded through LDS in a fixed order); loops are unrolled so that 8 loads are in flight.
constexpr int SEF = 4;
__device__ __forceinline__ int se_rp(int R) { return R <= 32 ? 32 : (R <= 64 ? 64 : (R <= 128 ? 128 : 256)); }

// out[f][j] (j < J) = sum_{i < I} in[f][i] * wt[i * J + j], all 256 threads: (j, slice of I); part: [nsl][SEF][J] floats
__device__ __forceinline__ void se_contract(const float* in /*LDS [SEF][I]*/, int I, const float* __restrict__ wt, int J,
                                            float* part, float* out /*LDS [SEF][J]*/) {
  const int JP = se_rp(J);
  const int nsl = 256 / JP;
  for (int j0 = 0; j0 < J; j0 += JP) {                          // one pass when J <= 256
    const int j = j0 + threadIdx.x % JP, sl = threadIdx.x / JP;
    float a[SEF];
#pragma unroll
    for (int f = 0; f < SEF; ++f) a[f] = 0.f;
    if (j < J) {
      const int per = (I + nsl - 1) / nsl;
      const int i_lo = sl * per, i_hi = min(I, i_lo + per);
#pragma unroll 16
      for (int i = i_lo; i < i_hi; ++i) {       // 16 weight loads in flight per trip (a trip is one exposed L2 round trip)
        const float wv = wt[(long)i * J + j];
#pragma unroll
        for (int f = 0; f < SEF; ++f) a[f] = fmaf(in[f * I + i], wv, a[f]);
      }
#pragma unroll
      for (int f = 0; f < SEF; ++f) part[(sl * SEF + f) * J + j] = a[f];
    }
    __syncthreads();
    for (int idx = threadIdx.x; idx < SEF * JP; idx += 256) {
      const int f = idx / JP, jj = j0 + idx % JP;
      if (jj < J) {
        float v = 0.f;
        for (int q = 0; q < nsl; ++q) v += part[(q * SEF + f) * J + jj];
        out[f * J + jj] = v;
      }
    }
    __syncthreads();
  }
}

__global__ __launch_bounds__(256) void se_train_fwd_kernel(const float* __restrict__ p, int N, int C, int R,
                                                           const float* __restrict__ w1t, const float* __restrict__ b1,
                                                           const float* __restrict__ w2t, const float* __restrict__ b2,
                                                           float* __restrict__ hid, float* __restrict__ gate) {
  extern __shared__ float sm[];        // p [SEF][C], hid [SEF][R], out [SEF][C], part [slices][SEF][J]
  float* sp = sm;
  float* sh = sp + SEF * C;
  float* so = sh + SEF * R;
  float* part = so + SEF * C;
  const long f0 = (long)blockIdx.x * SEF;
  for (int i = threadIdx.x; i < SEF * C; i += 256) {
    const int f = i / C;
    sp[i] = f0 + f < N ? p[(f0 + f) * C + (i - f * C)] : 0.f;
  }
  __syncthreads();
  se_contract(sp, C, w1t, R, part, sh);
  for (int i = threadIdx.x; i < SEF * R; i += 256) {
    const int f = i / R, j = i - f * R;
    const float a = fmaxf(sh[i] + b1[j], 0.f);
    sh[i] = a;
    if (f0 + f < N) hid[(f0 + f) * R + j] = a;
  }
  __syncthreads();
  se_contract(sh, R, w2t, C, part, so);
  for (int i = threadIdx.x; i < SEF * C; i += 256) {
    const int f = i / C, c = i - f * C;
    if (f0 + f < N) gate[(f0 + f) * C + c] = sigmoidf_(so[i] + b2[c]);
  }
}

static size_t se_train_smem(int C, int R) {
  const int mx = C > R ? C : R;
  const int part = SEF * mx > 1024 ? SEF * mx : 1024;           // (256 / RP) slices x SEF x J <= max(1024, SEF * J)
  return (size_t)(SEF * (2 * C + R) + part) * sizeof(float);
}

extern "C" int tdeed_se_train_fwd(const float* p, int N, int C, int R, const float* w1t, const float* b1,
                                  const float* w2t, const float* b2, float* hid, float* gate, void* stream) {
  TD_CHECK(p && w1t && b1 && w2t && b2 && hid && gate && N > 0 && C > 0 && R > 0, "se_train_fwd: bad arguments");
  const size_t smem = se_train_smem(C, R);
  TD_CHECK(smem <= 64 * 1024, "se_train_fwd: C=%d R=%d beyond the LDS budget", C, R);
  hipLaunchKernelGGL(se_train_fwd_kernel, dim3(cdiv(N, SEF)), dim3(256), smem, (hipStream_t)stream, p, N, C, R, w1t, b1, w2t,
                     b2, hid, gate);
  TD_LAUNCH_CHECK("se_train_fwd");
  return TDEED_OK;
}

// d_pre2 = d_gate * g * (1 - g);  d_hid = (hid > 0) * W2^T d_pre2;  d_p = W1^T d_hid
// w1 [R][C], w2 [C][R] (the reference layouts: here the contraction runs down the rows, so they are already "transposed")
__global__ __launch_bounds__(256) void se_train_bwd_kernel(const float* __restrict__ d_gate, const float* __restrict__ gate,
                                                           const float* __restrict__ hid, int N, int C, int R,
                                                           const float* __restrict__ w1, const float* __restrict__ w2,
                                                           float* __restrict__ d_pre2, float* __restrict__ d_hid,
                                                           float* __restrict__ d_p) {
  extern __shared__ float sm[];        // d_pre2 [SEF][C], d_hid [SEF][R], out [SEF][C], part
  float* s2 = sm;
  float* sh = s2 + SEF * C;
  float* so = sh + SEF * R;
  float* part = so + SEF * C;
  const long f0 = (long)blockIdx.x * SEF;
  for (int i = threadIdx.x; i < SEF * C; i += 256) {
    const int f = i / C, c = i - f * C;
    float v = 0.f;
    if (f0 + f < N) {
      const float g = gate[(f0 + f) * C + c];
      v = d_gate[(f0 + f) * C + c] * g * (1.f - g);
      d_pre2[(f0 + f) * C + c] = v;
    }
    s2[i] = v;
  }
  __syncthreads();
  se_contract(s2, C, w2, R, part, sh);                         // d_hid[j] = sum_c d_pre2[c] * w2[c][j]
  for (int i = threadIdx.x; i < SEF * R; i += 256) {
    const int f = i / R, j = i - f * R;
    float a = 0.f;
    if (f0 + f < N) {
      a = hid[(f0 + f) * R + j] > 0.f ? sh[i] : 0.f;
      d_hid[(f0 + f) * R + j] = a;
    }
    sh[i] = a;
  }
  __syncthreads();
  se_contract(sh, R, w1, C, part, so);                         // d_p[c] = sum_j d_hid[j] * w1[j][c]
  for (int i = threadIdx.x; i < SEF * C; i += 256) {
    const int f = i / C, c = i - f * C;
    if (f0 + f < N) d_p[(f0 + f) * C + c] = so[i];
  }
}

extern "C" int tdeed_se_train_bwd(const float* d_gate, const float* gate, const float* hid, int N, int C, int R,
                                  const float* w1, const float* w2, float* d_pre2, float* d_hid, float* d_p,
                                  void* stream) {
  TD_CHECK(d_gate && gate && hid && w1 && w2 && d_pre2 && d_hid && d_p && N > 0 && C > 0 && R > 0,
           "se_train_bwd: bad arguments");
  const size_t smem = se_train_smem(C, R);
  TD_CHECK(smem <= 64 * 1024, "se_train_bwd: C=%d R=%d beyond the LDS budget", C, R);
  hipLaunchKernelGGL(se_train_bwd_kernel, dim3(cdiv(N, SEF)), dim3(256), smem, (hipStream_t)stream, d_gate, gate, hid, N, C,
                     R, w1, w2, d_pre2, d_hid, d_p);
  TD_LAUNCH_CHECK("se_train_bwd");
  return TDEED_OK;
}

// y[n][px][c] = x[n][px][c] * s[n][c] + add[n][c] * add_scale   (s, add fp32 [N][C]; add may be NULL)
// forward: the SE scale; backward: d y2 = d(y2*gate) * gate + d_pool / hw
template <typename T>
__global__ __launch_bounds__(256) void scale_rows_kernel(const T* __restrict__ x, const float* __restrict__ s,
                                                         const float* __restrict__ add, float add_scale, int hw,
                                                         const float* __restrict__ in_a, const float* __restrict__ in_b,
                                                         T* __restrict__ y, int nch, int rpw) {
  constexpr int EPC = Chunk<T>::N;
  const RowMap mp(nch);
  if (!mp.on) return;
  const int C = nch * EPC, c0 = mp.ck * EPC;
  const long n = blockIdx.y;
  const float* pad = add ? add : s;                             // read unconditionally (scaled by 0 when there is no add)
  const float asc = add ? add_scale : 0.f;
  // in_a given: x is a raw conv output, relu(in_a[c] * x + in_b[c]) (BatchNorm + ReLU) is applied on load
  const bool aff = in_a != nullptr;
  const float* pia = aff ? in_a : s;
  const float* pib = aff ? in_b : s;
  float sv[EPC], ad[EPC], ia[EPC], ib[EPC];
#pragma unroll
  for (int e = 0; e < EPC; ++e) {
    sv[e] = s[n * C + c0 + e];
    ad[e] = pad[n * C + c0 + e] * asc;
    ia[e] = pia[c0 + e];
    ib[e] = pib[c0 + e];
  }
  const int m0 = blockIdx.x * rpw, m1 = min(hw, m0 + rpw);
  const T* xf = x + n * hw * C + c0;
  T* yf = y + n * hw * C + c0;
  for (int r0 = m0 + mp.rl; r0 < m1; r0 += mp.RL * RW_U) {
    float v[RW_U][EPC];
#pragma unroll
    for (int u = 0; u < RW_U; ++u) Chunk<T>::load(xf + (long)min(r0 + u * mp.RL, m1 - 1) * C, v[u]);
#pragma unroll
    for (int u = 0; u < RW_U; ++u) {
      const int r = r0 + u * mp.RL;
      if (r < m1) {
#pragma unroll
        for (int e = 0; e < EPC; ++e) {
          const float xv = aff ? (float)(T)fmaxf(fmaf(v[u][e], ia[e], ib[e]), 0.f) : v[u][e];     // rounded like the map
          v[u][e] = fmaf(xv, sv[e], ad[e]);
        }
        Chunk<T>::store(yf + (long)r * C, v[u]);
      }
    }
  }
}

extern "C" int tdeed_scale_rows(const void* x, const float* s, const float* add, float add_scale, int N, int hw, int C,
                                const float* in_a, const float* in_b, void* y, int dtype, void* stream) {
  TD_CHECK(x && s && y && N > 0 && hw > 0 && C > 0 && C % 8 == 0, "scale_rows: bad arguments");
  TD_CHECK(!in_a == !in_b, "scale_rows: in_a and in_b come together");
  hipStream_t st = (hipStream_t)stream;
  TD_CHECK(dtype == TDEED_F32 || dtype == TDEED_BF16, "scale_rows: bad dtype %d", dtype);
  const int nch = C / (dtype == TDEED_F32 ? 4 : 8);
  TD_CHECK(nch <= 256 && N <= 65535, "scale_rows: C=%d / N=%d out of range", C, N);
  int rpw = rows_per_wg(nch);
  const dim3 grid((unsigned)cdiv(hw, rpw), (unsigned)N);
  if (dtype == TDEED_F32)
    hipLaunchKernelGGL(scale_rows_kernel<float>, grid, dim3(256), 0, st, (const float*)x, s, add, add_scale, hw, in_a, in_b,
                       (float*)y, nch, rpw);
  else
    hipLaunchKernelGGL(scale_rows_kernel<bf16_t>, grid, dim3(256), 0, st, (const bf16_t*)x, s, add, add_scale, hw, in_a, in_b,
                       (bf16_t*)y, nch, rpw);
  TD_LAUNCH_CHECK("scale_rows");
  return TDEED_OK;
}

// =========================================================================== grouped 3x3 conv backward
// forward: y[n][oy][ox][g*gw+co] = sum_{ky,kx,ci} x[n][oy*s+ky-1][ox*s+kx-1][g*gw+ci] * w[g][ky*3+kx][ci][co]
// (w in the forward's packed fp32 layout [G][9][gw in][gw out]).  First versions on the vector ALU: one lane per
// (input pixel, group) for the input gradient, LDS-staged pixel chunks for the weight gradient.
template <typename T, int GW>
__global__ __launch_bounds__(256) void gconv_dgrad_kernel(const T* __restrict__ dy, int Hi, int Wi, int Ho, int Wo, int C,
                                                          int stride, const float* __restrict__ w, T* __restrict__ dx,
                                                          long npix) {
  __shared__ float sw[9 * GW * GW];
  const int g = blockIdx.y;
  for (int i = threadIdx.x; i < 9 * GW * GW; i += 256) sw[i] = w[(long)g * 9 * GW * GW + i];
  __syncthreads();
  const long pix = (long)blockIdx.x * 256 + threadIdx.x;
  if (pix >= npix) return;
  const int ix = (int)(pix % Wi);
  const int iy = (int)((pix / Wi) % Hi);
  const long n = pix / ((long)Wi * Hi);
  float acc[GW];
#pragma unroll
  for (int c = 0; c < GW; ++c) acc[c] = 0.f;
#pragma unroll
  for (int ky = 0; ky < 3; ++ky) {
    const int ty = iy + 1 - ky;
    if (ty < 0 || ty % stride != 0) continue;
    const int oy = ty / stride;
    if (oy >= Ho) continue;
#pragma unroll
    for (int kx = 0; kx < 3; ++kx) {
      const int tx = ix + 1 - kx;
      if (tx < 0 || tx % stride != 0) continue;
      const int ox = tx / stride;
      if (ox >= Wo) continue;
      float d[GW];
      const T* src = dy + ((n * Ho + oy) * Wo + ox) * C + g * GW;
#pragma unroll
      for (int c = 0; c < GW; c += Chunk<T>::N) Chunk<T>::load(src + c, *reinterpret_cast<float(*)[Chunk<T>::N]>(&d[c]));
      const float* wt = sw + (ky * 3 + kx) * GW * GW;
#pragma unroll
      for (int ci = 0; ci < GW; ++ci) {
        float a = acc[ci];
#pragma unroll
        for (int co = 0; co < GW; ++co) a = fmaf(d[co], wt[ci * GW + co], a);
        acc[ci] = a;
      }
    }
  }
  T* dst = dx + pix * C + g * GW;
#pragma unroll
  for (int c = 0; c < GW; c += Chunk<T>::N) Chunk<T>::store(dst + c, *reinterpret_cast<float(*)[Chunk<T>::N]>(&acc[c]));
}

// Stride-2 input gradient, tiled (bf16 and fp32).  A workgroup owns a 16x16 tile of input pixels of one frame; the
// (8+2) x (8+2) output pixels whose gradient reaches it sit in LDS as whole rows (coalesced 16-byte loads; the row-per-lane
// form above fetches 16 bytes per 100..700-byte row).  A wave serves one group at a time (4 groups in flight); its 64 lanes
// are the 64 pixels of ONE parity class (iy & 1, ix & 1), so all lanes take the same taps -- 1, 2, 2 or 4 of the 9 --
// with no divergence; the group's 9 x gw x gw weights are wave-uniform LDS reads.
template <typename T, int GW>
__global__ __launch_bounds__(256) void gconv_dgrad_s2_kernel(const T* __restrict__ dy, int Hi, int Wi, int Ho, int Wo, int C,
                                                             const float* __restrict__ w, T* __restrict__ dx) {
  constexpr int EPC = Chunk<T>::N;
  constexpr int TO = 10;                                        // output rows / columns staged: 16 / 2 + 2
  extern __shared__ __attribute__((aligned(16))) unsigned char smraw[];
  const int CP = C + EPC;                                       // row pad: rows shift by one 16-byte bank slot
  T* dyt = reinterpret_cast<T*>(smraw);                         // [TO*TO][CP]
  float* sw = reinterpret_cast<float*>(dyt + TO * TO * CP);     // [4][9*GW*GW]
  const int n = blockIdx.z;
  const int iy0 = blockIdx.y * 16, ix0 = blockIdx.x * 16;
  const int oyb = iy0 / 2 - (iy0 > 0 ? 0 : 0), oxb = ix0 / 2;   // output row of tile-local ty = 0 is iy0 / 2 (iy0 even)
  // ty = iy + 1 - ky in [iy0 - 1, iy0 + 16]: outputs oy in [iy0/2 - 0 (ty = iy0: ky = 1) ... ] -> stage oy in [iy0/2 - 1 + 1, ...]:
  // valid ty are even; smallest even ty >= iy0 - 1 is iy0 (oy = iy0/2), largest <= iy0 + 16 is iy0 + 16 (oy = iy0/2 + 8)
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int nck = C / EPC;
  {
    const IDiv dck(nck);
    for (int i0 = tid; i0 < TO * TO * nck; i0 += 256 * 4) {
      u32x4 v[4];
      bool ok[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int i = min(i0 + u * 256, TO * TO * nck - 1);
        int px, ck;
        dck.divmod(i, px, ck);
        const int ry = px / TO, rx = px - ry * TO;
        const int oy = oyb + ry, ox = oxb + rx;
        ok[u] = oy < Ho && ox < Wo;
        v[u] = *reinterpret_cast<const u32x4*>(dy + (((long)n * Ho + (ok[u] ? oy : 0)) * Wo + (ok[u] ? ox : 0)) * C + ck * EPC);
      }
      TD_ISSUE_FENCE();
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int i = i0 + u * 256;
        if (i < TO * TO * nck) {
          int px, ck;
          dck.divmod(i, px, ck);
          *reinterpret_cast<u32x4*>(dyt + px * CP + ck * EPC) = ok[u] ? v[u] : (u32x4){0u, 0u, 0u, 0u};
        }
      }
    }
  }
  const int G = C / GW;
  const int ly = lane >> 3, lx = lane & 7;                      // this lane's position inside its parity class (8 x 8)
  for (int g0 = 0; g0 < G; g0 += 4) {
    __syncthreads();                                            // dy tile staged / previous round's weights consumed
    for (int i = tid; i < 4 * 9 * GW * GW; i += 256) {
      const int gg = i / (9 * GW * GW);
      sw[i] = g0 + gg < G ? w[(long)(g0 + gg) * 9 * GW * GW + (i - gg * 9 * GW * GW)] : 0.f;
    }
    __syncthreads();
    const int g = g0 + wv;
    if (g < G) {
      const float* wg = sw + wv * 9 * GW * GW;
#pragma unroll
      for (int cls = 0; cls < 4; ++cls) {
        const int pyc = cls >> 1, pxc = cls & 1;                // parity of (iy, ix)
        const int iy = iy0 + 2 * ly + pyc, ix = ix0 + 2 * lx + pxc;
        float acc[GW];
#pragma unroll
        for (int c = 0; c < GW; ++c) acc[c] = 0.f;
        // ty = iy + 1 - ky even: iy even -> ky = 1; iy odd -> ky in {0, 2}
#pragma unroll
        for (int a = 0; a < 2; ++a) {
          if (pyc == 0 && a == 1) continue;
          const int ky = pyc == 0 ? 1 : 2 * a;
          const int ry = (2 * ly + pyc + 1 - ky) / 2;           // (iy + 1 - ky) / 2 - iy0 / 2, in [0, 8]
#pragma unroll
          for (int bq = 0; bq < 2; ++bq) {
            if (pxc == 0 && bq == 1) continue;
            const int kx = pxc == 0 ? 1 : 2 * bq;
            const int rx = (2 * lx + pxc + 1 - kx) / 2;
            float d[GW];
            const T* src = dyt + (ry * TO + rx) * CP + g * GW;
#pragma unroll
            for (int c = 0; c < GW; c += EPC) Chunk<T>::load(src + c, *reinterpret_cast<float(*)[EPC]>(&d[c]));
            const float* wt = wg + (ky * 3 + kx) * GW * GW;
#pragma unroll
            for (int ci = 0; ci < GW; ++ci) {
              float s = acc[ci];
#pragma unroll
              for (int co = 0; co < GW; ++co) s = fmaf(d[co], wt[ci * GW + co], s);
              acc[ci] = s;
            }
          }
        }
        if (iy < Hi && ix < Wi) {
          T* dst = dx + (((long)n * Hi + iy) * Wi + ix) * C + g * GW;
#pragma unroll
          for (int c = 0; c < GW; c += EPC) Chunk<T>::store(dst + c, *reinterpret_cast<float(*)[EPC]>(&acc[c]));
        }
      }
    }
  }
}

// Stride-2 input gradient on the MFMA pipe (bf16).  Per parity class of input pixels the gradient is a small GEMM
//   dx[pixel][ci] = sum_{taps of the class} sum_co dy[tap pixel][co] * W[tap][ci][co]
// over a 16-channel unit (one gw = 16 group, or two gw = 8 groups as a block-diagonal weight): K = (tap slot, co) = 32 is one
// v_mfma_f32_16x16x32_bf16 for the classes with 1 or 2 taps and two for the (odd, odd) class with 4.  The weights are the
// MFMA A operand (rows = ci), so a lane ends with 4 consecutive channels of one pixel.
// wfrag: [C/16][5][64] fragments: f = 0 (even, even): tap (1,1); f = 1 (even y, odd x): (1,0) (1,2); f = 2 (odd y, even x):
// (0,1) (2,1); f = 3, 4 (odd, odd): (0,0) (0,2) and (2,0) (2,2).
__global__ __launch_bounds__(256) void gconv_dgrad_s2_pack_kernel(const float* __restrict__ w, int C, int gw,
                                                                  bf16x8* __restrict__ wfrag) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  const int U = (C + 15) / 16;
  if (i >= U * 5 * 64) return;
  const int lane = i & 63, f = (i >> 6) % 5, u = i / 320;
  const int m = lane & 15, kq = lane >> 4, ts = kq >> 1, co0 = (kq & 1) * 8;
  int ky, kx;
  if (f == 0) { ky = 1; kx = 1; }
  else if (f == 1) { ky = 1; kx = ts ? 2 : 0; }
  else if (f == 2) { kx = 1; ky = ts ? 2 : 0; }
  else { ky = f == 3 ? 0 : 2; kx = ts ? 2 : 0; }
  const bool tap_ok = !(f == 0 && ts == 1);
  bf16x8 out;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int co = co0 + j;
    float v = 0.f;
    if (tap_ok) {
      if (gw == 16) v = w[(((long)u * 9 + ky * 3 + kx) * 16 + m) * 16 + co];
      else if ((co >> 3) == (m >> 3) && (2 * u + (m >> 3)) * 8 < C)
        v = w[(((long)(2 * u + (m >> 3)) * 9 + ky * 3 + kx) * 8 + (m & 7)) * 8 + (co & 7)];
    }
    out[j] = (__bf16)v;
  }
  wfrag[i] = out;
}

// workgroup = (band of 16 input rows, frame, chunk of 64 channels), walking the 16-pixel-wide tiles of the band; wave =
// one 16-channel unit of the chunk with its 5 weight fragments in registers; the 10 x 10 output pixels that reach a tile
// are staged in LDS (zero outside the map).
// bz given (training): dx arrives at conv1's BatchNorm + ReLU, whose backward needs sum g and sum g * (z1 - mean), g = dx masked
// by [fa z1 + fb > 0]: every lane sums them over what it stores (z1 read here once, 8 bytes beside each store), one fold over
// the 16 pixel lanes at the end leaves the partial row part_s / part_q [frame * bands + band][C] of this workgroup's 64 channels.
struct S2Stat {
  const bf16_t* z; const float* fa; const float* fb; const float* mean; float* part_s; float* part_q;
};
__global__ __launch_bounds__(256) void gconv_dgrad_s2_mfma_kernel(const bf16_t* __restrict__ dy, int Hi, int Wi, int Ho,
                                                                  int Wo, int C, const bf16x8* __restrict__ wfrag,
                                                                  bf16_t* __restrict__ dx, const S2Stat bst) {
  constexpr int TO = 10, CP = 64 + 8;                           // pixel stride 144 B: odd number of 16-byte slots
  __shared__ __attribute__((aligned(16))) bf16_t dyt[TO * TO * CP];
  const int n = blockIdx.y, iy0 = blockIdx.x * 16, c0 = blockIdx.z * 64;
  const int CH = min(64, C - c0), nck = ((CH + 15) >> 4) * 2;   // 16-byte chunks staged (an odd 8-channel tail is zero-padded)
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int kq = lane >> 4, ts = kq >> 1, pl = lane & 15;
  // a wave serves a PAIR of adjacent 16-channel units (pr) on two of the four 16-pixel tiles (mh, mh + 2): after a row swap
  // between the two accumulators (v_permlane16_swap) a lane holds 8 consecutive channels -- 16-byte stores and statistics
  // loads, 64 bytes per pixel and wave instruction instead of 32 (tools/ubench/access_shape.hip: 32-byte segments move the
  // same bytes ~1.5x slower)
  const int pr = wv & 1, mh = wv >> 1;
  const int uA = (c0 >> 4) + 2 * pr, uB = uA + 1;
  const int U = (C + 15) >> 4;
  const bool active = uA < U;
  bf16x8 afA[5], afB[5];
#pragma unroll
  for (int f = 0; f < 5; ++f) {
    afA[f] = wfrag[((long)min(uA, U - 1) * 5 + f) * 64 + lane];
    afB[f] = wfrag[((long)min(uB, U - 1) * 5 + f) * 64 + lane];
  }
  const int chA = uA * 16 + kq * 4, chB = chA + 16;              // accumulator rows
  const int chS = uA * 16 + (kq & 1) * 16 + (kq >> 1) * 8;       // after the swap: this lane's 8 consecutive channels
  const bool sok = chS < C;
  const int oyb = iy0 >> 1;
  const IDiv dck(nck);
  const int ntx = (Wi + 15) >> 4;
  float faA[4], fbA[4], muA[4], faB[4], fbB[4], muB[4];
  float ps1A[4] = {0.f, 0.f, 0.f, 0.f}, ps2A[4] = {0.f, 0.f, 0.f, 0.f}, ps1B[4] = {0.f, 0.f, 0.f, 0.f}, ps2B[4] = {0.f, 0.f, 0.f, 0.f};
  if (bst.z) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int ca = min(chA + r, C - 1), cb = min(chB + r, C - 1);
      faA[r] = bst.fa[ca]; fbA[r] = bst.fb[ca]; muA[r] = bst.mean[ca];
      faB[r] = bst.fa[cb]; fbB[r] = bst.fb[cb]; muB[r] = bst.mean[cb];
    }
  }
  for (int tx = 0; tx < ntx; ++tx) {
    const int ix0 = tx * 16, oxb = ix0 >> 1;
    __syncthreads();
    for (int i0 = tid; i0 < TO * TO * nck; i0 += 256 * 4) {
      u32x4 v[4];
      bool ok[4];
#pragma unroll
      for (int b = 0; b < 4; ++b) {
        const int i = min(i0 + b * 256, TO * TO * nck - 1);
        int px, ck;
        dck.divmod(i, px, ck);
        const int ry = px / TO, rx = px - ry * TO;
        const int oy = oyb + ry, ox = oxb + rx;
        ok[b] = oy < Ho && ox < Wo && ck * 8 < CH;
        v[b] = *reinterpret_cast<const u32x4*>(dy + (((long)n * Ho + (ok[b] ? oy : 0)) * Wo + (ok[b] ? ox : 0)) * C + c0 + (ok[b] ? ck * 8 : 0));
      }
      TD_ISSUE_FENCE();
#pragma unroll
      for (int b = 0; b < 4; ++b) {
        const int i = i0 + b * 256;
        if (i < TO * TO * nck) {
          int px, ck;
          dck.divmod(i, px, ck);
          *reinterpret_cast<u32x4*>(dyt + px * CP + ck * 8) = ok[b] ? v[b] : (u32x4){0u, 0u, 0u, 0u};
        }
      }
    }
    __syncthreads();
    if (!active) continue;
    const bf16_t* colA = dyt + 2 * pr * 16 + (kq & 1) * 8;
    const bf16_t* colB = colA + 16;
#pragma unroll
    for (int cls = 0; cls < 4; ++cls) {
      const int pyc = cls >> 1, pxc = cls & 1;
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int mt = mh + 2 * j;
        const int p = mt * 16 + pl, ly = p >> 3, lx = p & 7;
        f32x4 accA = {0.f, 0.f, 0.f, 0.f}, accB = {0.f, 0.f, 0.f, 0.f};
        int o0, o1 = 0;                                          // pixel offsets of the one or two K slabs of the class
        if (cls == 0) o0 = ly * TO + lx;
        else if (cls == 1) o0 = ly * TO + lx + 1 - ts;           // ky = 1 (ry = ly); kx = 0 -> rx = lx + 1, kx = 2 -> rx = lx
        else if (cls == 2) o0 = (ly + 1 - ts) * TO + lx;         // kx = 1 (rx = lx); ky = 0 -> ry = ly + 1, ky = 2 -> ry = ly
        else { o0 = (ly + 1) * TO + lx + 1 - ts; o1 = ly * TO + lx + 1 - ts; }
        const int f0 = cls;                                      // fragments 0, 1, 2, 3 (+ 4 for the (odd, odd) class)
        accA = __builtin_amdgcn_mfma_f32_16x16x32_bf16(afA[f0], *reinterpret_cast<const bf16x8*>(colA + o0 * CP), accA, 0, 0, 0);
        accB = __builtin_amdgcn_mfma_f32_16x16x32_bf16(afB[f0], *reinterpret_cast<const bf16x8*>(colB + o0 * CP), accB, 0, 0, 0);
        if (cls == 3) {
          accA = __builtin_amdgcn_mfma_f32_16x16x32_bf16(afA[4], *reinterpret_cast<const bf16x8*>(colA + o1 * CP), accA, 0, 0, 0);
          accB = __builtin_amdgcn_mfma_f32_16x16x32_bf16(afB[4], *reinterpret_cast<const bf16x8*>(colB + o1 * CP), accB, 0, 0, 0);
        }
        const int iy = iy0 + 2 * ly + pyc, ix = ix0 + 2 * lx + pxc;
        const bool pok = iy < Hi && ix < Wi;
        const long off = (((long)n * Hi + (pok ? iy : 0)) * Wi + (pok ? ix : 0)) * C + (sok ? chS : 0);
        typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
        const bf16x4 oA = {(__bf16)accA[0], (__bf16)accA[1], (__bf16)accA[2], (__bf16)accA[3]};
        const bf16x4 oB = {(__bf16)accB[0], (__bf16)accB[1], (__bf16)accB[2], (__bf16)accB[3]};
        if (bst.z) {
          // the statistics map arrives in the stored (swapped) layout: the same swap takes it back to accumulator rows
          const u32x4 zs = *reinterpret_cast<const u32x4*>(bst.z + off);
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
              ps1A[r] += ga;
              ps2A[r] = fmaf(ga, za - muA[r], ps2A[r]);
              ps1B[r] += gb;
              ps2B[r] = fmaf(gb, zb - muB[r], ps2B[r]);
            }
          }
        }
        const u32x2 a2 = *reinterpret_cast<const u32x2*>(&oA), b2 = *reinterpret_cast<const u32x2*>(&oB);
        const auto s0 = __builtin_amdgcn_permlane16_swap(a2[0], b2[0], false, false);
        const auto s1 = __builtin_amdgcn_permlane16_swap(a2[1], b2[1], false, false);
        if (pok && sok) *reinterpret_cast<u32x4*>(dx + off) = (u32x4){s0[0], s1[0], s0[1], s1[1]};
      }
    }
  }
  if (bst.z) {
    // lanes sharing kq (the 16 pixels of a tile), then the two waves sharing the pair, in a fixed order
    __shared__ float red[4][2][32];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
#pragma unroll
      for (int o = 1; o < 16; o <<= 1) {
        ps1A[r] += __shfl_xor(ps1A[r], o, 64);
        ps2A[r] += __shfl_xor(ps2A[r], o, 64);
        ps1B[r] += __shfl_xor(ps1B[r], o, 64);
        ps2B[r] += __shfl_xor(ps2B[r], o, 64);
      }
    }
    if (pl == 0) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        red[wv][0][kq * 4 + r] = ps1A[r];
        red[wv][0][16 + kq * 4 + r] = ps1B[r];
        red[wv][1][kq * 4 + r] = ps2A[r];
        red[wv][1][16 + kq * 4 + r] = ps2B[r];
      }
    }
    __syncthreads();
    if (tid < 128) {
      const int which = tid >> 6, c = tid & 63, pp = c >> 5, cl = c & 31;
      if (c0 + c < C) {
        const float a = red[pp][which][cl] + red[pp + 2][which][cl];
        const long row = ((long)n * gridDim.x + blockIdx.x) * C + c0 + c;
        (which ? bst.part_q : bst.part_s)[row] = a;
      }
    }
  }
}

// weight gradient: workgroup = (slab of output pixels, group); chunks of 32 output pixels staged in LDS as
// d[32][GW] and x9[32][9][GW]; lane (tap, ci, co) triples accumulate over the chunk.
template <typename T, int GW>
__global__ __launch_bounds__(256) void gconv_wgrad_kernel(const T* __restrict__ x, const T* __restrict__ dy, int Hi, int Wi,
                                                          int Ho, int Wo, int C, int stride, long npix_out,
                                                          long pix_per_slab, float* __restrict__ part) {
  constexpr int NW = 9 * GW * GW;                               // outputs per group
  constexpr int PER = (NW + 255) / 256;
  __shared__ float sd[32][GW];
  __shared__ float sx[32][9][GW];
  const int g = blockIdx.y;
  const long p_begin = (long)blockIdx.x * pix_per_slab, p_end = min(npix_out, p_begin + pix_per_slab);
  float acc[PER];
#pragma unroll
  for (int i = 0; i < PER; ++i) acc[i] = 0.f;
  for (long p0 = p_begin; p0 < p_end; p0 += 32) {
    __syncthreads();
    for (int i = threadIdx.x; i < 32 * GW; i += 256) {
      const int pp = i / GW, c = i % GW;
      const long p = p0 + pp;
      sd[pp][c] = p < p_end ? (float)dy[p * C + g * GW + c] : 0.f;
    }
    for (int i = threadIdx.x; i < 32 * 9 * GW; i += 256) {
      const int c = i % GW, tap = (i / GW) % 9, pp = i / (9 * GW);
      const long p = p0 + pp;
      float v = 0.f;
      if (p < p_end) {
        const int ox = (int)(p % Wo), oy = (int)((p / Wo) % Ho);
        const long n = p / ((long)Wo * Ho);
        const int iy = oy * stride + tap / 3 - 1, ix = ox * stride + tap % 3 - 1;
        if (iy >= 0 && iy < Hi && ix >= 0 && ix < Wi) v = (float)x[((n * Hi + iy) * Wi + ix) * C + g * GW + c];
      }
      sx[pp][tap][c] = v;
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < PER; ++i) {
      const int o = threadIdx.x + i * 256;
      if (o < NW) {
        const int co = o % GW, ci = (o / GW) % GW, tap = o / (GW * GW);
        float a = acc[i];
#pragma unroll 8
        for (int pp = 0; pp < 32; ++pp) a = fmaf(sx[pp][tap][ci], sd[pp][co], a);
        acc[i] = a;
      }
    }
  }
#pragma unroll
  for (int i = 0; i < PER; ++i) {
    const int o = threadIdx.x + i * 256;
    if (o < NW) part[((long)blockIdx.x * gridDim.y + g) * NW + o] = acc[i];
  }
}

// bf16 weight gradient on the MFMA pipe.  Unit = 16 channels (two gw=8 groups, or one gw=16 group); per chunk of 32
// output pixels the gradient rows d[px][16 co] and the nine shifted input rows x[px+tap][16 ci] go to LDS transposed
// ([channel][pixel]) so that the pixel contraction is the MFMA k index: D[ci][co] += x_tap^T . d, one MFMA per tap.
// For gw=8 only the two diagonal 8x8 blocks of D are weight gradients (the off-diagonal cross terms are dropped).
constexpr int GW_LD = 36;
template <int GW>
__global__ __launch_bounds__(256) void gconv_wgrad_mfma_kernel(const bf16_t* __restrict__ x, const bf16_t* __restrict__ dy,
                                                               int Hi, int Wi, int Ho, int Wo, int C, int stride,
                                                               long npix_out, long pix_per_slab, float* __restrict__ part) {
  __shared__ __attribute__((aligned(16))) bf16_t sd[16 * GW_LD];
  __shared__ __attribute__((aligned(16))) bf16_t sx[9 * 16 * GW_LD];
  const int unit = blockIdx.y, c0 = unit * 16;
  const long p_begin = (long)blockIdx.x * pix_per_slab, p_end = min(npix_out, p_begin + pix_per_slab);
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, pl = lane & 15, q = lane >> 4;
  const bool cok8[2] = {c0 < C, c0 + 8 < C};
  f32x4 acc[3];
#pragma unroll
  for (int i = 0; i < 3; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
  const IDiv dwo(Wo), dho(Ho);
  for (long p0 = p_begin; p0 < p_end; p0 += 32) {
    // ---- issue: this lane's pieces (piece = (tap | 9 for dy, pixel, 8-channel half)): 10 * 32 * 2 = 640 pieces
    u32x4 v[3];
    int meta[3];                                                // LDS row base (channel*LD + px) or -1
    bool ok[3];
#pragma unroll
    for (int u = 0; u < 3; ++u) {
      const int i = tid + u * 256;
      meta[u] = -1;
      ok[u] = false;
      v[u] = (u32x4){0u, 0u, 0u, 0u};
      const int ic = min(i, 639);
      const int half = ic & 1, pp = (ic >> 1) & 31, tap = ic >> 6;
      const long p = min(p0 + pp, p_end - 1);
      int oy, ox, tmp;
      dwo.divmod((int)(p % ((long)Ho * Wo)), oy, ox);
      (void)tmp; (void)dho;
      const long n = p / ((long)Ho * Wo);
      const bf16_t* src;
      bool valid = (p0 + pp < p_end) && cok8[half] && i < 640;
      if (tap == 9) {
        src = dy + p * C + c0 + half * 8;
      } else {
        const int iy = oy * stride + tap / 3 - 1, ix = ox * stride + tap % 3 - 1;
        const bool in = iy >= 0 && iy < Hi && ix >= 0 && ix < Wi;
        valid = valid && in;
        src = x + ((n * Hi + (in ? iy : 0)) * Wi + (in ? ix : 0)) * C + c0 + half * 8;
      }
      if (!cok8[half]) src = x;                                 // keep the address valid; the value is dropped
      v[u] = *reinterpret_cast<const u32x4*>(src);
      ok[u] = valid;
      if (i < 640) meta[u] = (tap * 16 + half * 8) * GW_LD + pp;
    }
    TD_ISSUE_FENCE();
    __syncthreads();
#pragma unroll
    for (int u = 0; u < 3; ++u) {
      if (meta[u] < 0) continue;
      const bf16x8 t8 = *reinterpret_cast<const bf16x8*>(&v[u]);
      const bool isd = meta[u] >= 9 * 16 * GW_LD;
      bf16_t* dst = isd ? sd + (meta[u] - 9 * 16 * GW_LD) : sx + meta[u];
#pragma unroll
      for (int e = 0; e < 8; ++e) dst[e * GW_LD] = ok[u] ? t8[e] : (bf16_t)0.f;
    }
    __syncthreads();
    bf16x8 bfr;
    {
      const bf16x4 lo = *reinterpret_cast<const bf16x4*>(sd + pl * GW_LD + q * 8);
      const bf16x4 hi = *reinterpret_cast<const bf16x4*>(sd + pl * GW_LD + q * 8 + 4);
      bfr = (bf16x8){lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    }
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      const int tap = wv + 4 * i;
      if (tap < 9) {
        const bf16x4 lo = *reinterpret_cast<const bf16x4*>(sx + (tap * 16 + pl) * GW_LD + q * 8);
        const bf16x4 hi = *reinterpret_cast<const bf16x4*>(sx + (tap * 16 + pl) * GW_LD + q * 8 + 4);
        const bf16x8 afr = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(afr, bfr, acc[i], 0, 0, 0);
      }
    }
  }
  // D[ci = 4q+e][co = pl] of tap wv + 4i  ->  part[slab][g][tap][ci_local][co_local]
  const int G = C / GW;
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    const int tap = wv + 4 * i;
    if (tap >= 9) continue;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int ci = 4 * q + e, co = pl;
      if (GW == 8 && (ci >> 3) != (co >> 3)) continue;          // cross-group term
      const int g = (c0 + ci) / GW;
      if (c0 + ci >= C || c0 + co >= C) continue;
      part[(((long)blockIdx.x * G + g) * 9 + tap) * GW * GW + (ci % GW) * GW + (co % GW)] = acc[i][e];
    }
  }
}

// bf16 weight gradient, transposing LDS reads (gfx950 ds_read_b64_tr_b16).  The contraction index of
//   dW[tap][ci][co] = sum_pixels x[pixel + tap][ci] * dy[pixel][co]
// is the pixel, which is the STRIDED direction of both channels-last maps, so an MFMA operand lane (one channel, 8
// consecutive pixels) is a column of a [pixel][channel] image.  The kernel above transposes with 2-byte LDS scatter
// writes (8 per 16-byte piece, and every input pixel is fetched once per tap); here the input patch of an 8 x 8 output
// tile and the tile's dy rows go to LDS as they are (16-byte vector writes, each input pixel fetched ONCE), and every
// operand is two transposing reads: per 16-lane group the hardware hands lane i column i of 4 rows whose addresses the
// lanes supply -- 4 consecutive output pixels, wherever the tap and the stride put their input pixels in the patch.
// workgroup = (run of tiles, chunk of 64 channels); wave = one 16-channel unit with all 9 tap accumulators.
constexpr int GWT_RS = 80;                                      // LDS row stride (elements): 64 channels + 32 bytes
// AFF: x is the RAW output z of the conv in front and relu(in_a[c] * z + in_b[c]) -- the BatchNorm + ReLU between the two
// convs -- is applied while the patch is staged (zero padding stays zero): the post-BN map is never materialised.
template <int S, int GW, bool AFF>
__global__ __launch_bounds__(256, S == 1 ? 4 : 3) void gconv_wgrad_tr_kernel(const bf16_t* __restrict__ x, const bf16_t* __restrict__ dy,
                                                             const float* __restrict__ in_a, const float* __restrict__ in_b,
                                                             int N, int Hi, int Wi, int Ho, int Wo, int C, int tiles_per_wg,
                                                             float* __restrict__ part) {
  constexpr int PW = 8 * S + (S == 1 ? 2 : 1);                  // patch width / height in input pixels: 10 or 17
  constexpr int NPIX = PW * PW;
  // LDS row stride (elements).  Stride 2: 64 channels + 16 bytes, 51 KB per workgroup = THREE per CU (with + 32 bytes: 57 KB, two);
  // these kernels are chains of dependent stage -> barrier -> MFMA rounds and live on resident workgroups (a form that put a
  // whole tile's loads in flight at the cost of one workgroup per CU was 30 % slower)
  constexpr int RSW = S == 2 ? 72 : GWT_RS;
  __shared__ __attribute__((aligned(16))) bf16_t patch[NPIX * RSW];
  __shared__ __attribute__((aligned(16))) bf16_t dyt[64 * RSW];
  __shared__ __attribute__((aligned(16))) float saff[2][64];
  const int c0 = blockIdx.y * 64, CH = min(64, C - c0), nck = CH >> 3;
  if constexpr (AFF) {
    if (threadIdx.x < 128) {
      const int c = c0 + (threadIdx.x & 63);
      saff[threadIdx.x >> 6][threadIdx.x & 63] = c < C ? (threadIdx.x < 64 ? in_a : in_b)[c] : 0.f;
    }
  }
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const bool active = c0 + wv * 16 < C;
  const int tyN = (Ho + 7) >> 3, txN = (Wo + 7) >> 3;
  const long total = (long)N * tyN * txN;
  const long t_lo = (long)blockIdx.x * tiles_per_wg, t_hi = min(total, t_lo + tiles_per_wg);
  f32x4 acc[9];
#pragma unroll
  for (int i = 0; i < 9; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
  // the pad columns of both images are read by the unit that holds an 8-channel tail: keep them finite
  for (int i = tid; i < (NPIX + 64) * (RSW / 8); i += 256) {
    bf16_t* base = i < NPIX * (RSW / 8) ? patch + (long)i * 8 : dyt + (long)(i - NPIX * (RSW / 8)) * 8;
    *reinterpret_cast<u32x4*>(base) = (u32x4){0u, 0u, 0u, 0u};
  }
  const IDiv dck(nck), dpw(PW), dtx(txN), dty(tyN);
  const int g4 = lane >> 4, q4 = (lane >> 2) & 3, p4 = lane & 3;
  for (long t = t_lo; t < t_hi; ++t) {
    int rest, txi, tyi, n;
    dtx.divmod((int)t, rest, txi);
    dty.divmod(rest, n, tyi);
    const int oy0 = tyi * 8, ox0 = txi * 8;
    __syncthreads();                                            // previous tile consumed (and the zero fill done)
    const int n_patch = NPIX * nck, n_all = n_patch + 64 * nck;
    for (int i0 = tid; i0 < n_all; i0 += 256 * 4) {
      u32x4 v[4];
      bool ok[4];
      int dst[4], cko[4];
#pragma unroll
      for (int b = 0; b < 4; ++b) {
        const int i = min(i0 + b * 256, n_all - 1);
        int px, ck;
        const bf16_t* src;
        cko[b] = -1;
        if (i < n_patch) {
          dck.divmod(i, px, ck);
          cko[b] = ck * 8;
          int py, pxx;
          dpw.divmod(px, py, pxx);
          const int iy = oy0 * S - 1 + py, ix = ox0 * S - 1 + pxx;
          ok[b] = iy >= 0 && iy < Hi && ix >= 0 && ix < Wi;
          src = x + (((long)n * Hi + (ok[b] ? iy : 0)) * Wi + (ok[b] ? ix : 0)) * C + c0 + ck * 8;
          dst[b] = px * RSW + ck * 8;
        } else {
          dck.divmod(i - n_patch, px, ck);
          const int oy = oy0 + (px >> 3), ox = ox0 + (px & 7);
          ok[b] = oy < Ho && ox < Wo;
          src = dy + (((long)n * Ho + (ok[b] ? oy : 0)) * Wo + (ok[b] ? ox : 0)) * C + c0 + ck * 8;
          dst[b] = NPIX * RSW + px * RSW + ck * 8;        // dyt follows patch in the shared segment? no: flagged below
        }
        v[b] = *reinterpret_cast<const u32x4*>(src);
      }
      TD_ISSUE_FENCE();
#pragma unroll
      for (int b = 0; b < 4; ++b) {
        if (i0 + b * 256 < n_all) {
          bf16_t* d = dst[b] < NPIX * RSW ? patch + dst[b] : dyt + (dst[b] - NPIX * RSW);
          if constexpr (AFF) {
            if (cko[b] >= 0) {                                  // an input piece: BatchNorm affine + ReLU of the layer in front
              const bf16x8 t8 = *reinterpret_cast<const bf16x8*>(&v[b]);
              bf16x8 o8;
#pragma unroll
              for (int e = 0; e < 8; ++e)
                o8[e] = (bf16_t)fmaxf(fmaf((float)t8[e], saff[0][cko[b] + e], saff[1][cko[b] + e]), 0.f);
              v[b] = *reinterpret_cast<const u32x4*>(&o8);
            }
          }
          *reinterpret_cast<u32x4*>(d) = ok[b] ? v[b] : (u32x4){0u, 0u, 0u, 0u};
        }
      }
    }
    __syncthreads();
    if (!active) continue;
    const int colo = wv * 16 + p4 * 4;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const int ty = ks * 4 + g4;
      const bf16x8 bfr = td_tr_read8(dyt + (ty * 8 + q4) * RSW + colo, dyt + (ty * 8 + q4 + 4) * RSW + colo);
#pragma unroll
      for (int tap = 0; tap < 9; ++tap) {
        const int ky = tap / 3, kx = tap - ky * 3;
        const int r0 = (ty * S + ky) * PW + q4 * S + kx;
        const bf16x8 afr = td_tr_read8(patch + r0 * RSW + colo, patch + (r0 + 4 * S) * RSW + colo);
        acc[tap] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(afr, bfr, acc[tap], 0, 0, 0);
      }
    }
  }
  if (!active) return;
  // D[ci = 4 (lane >> 4) + e][co = lane & 15] of every tap  ->  part[slab][g][tap][ci_local][co_local]
  const int G = C / GW, u0 = c0 + wv * 16;
#pragma unroll
  for (int tap = 0; tap < 9; ++tap) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int ci = 4 * g4 + e, co = lane & 15;
      if (GW == 8 && (ci >> 3) != (co >> 3)) continue;          // cross-group term
      if (u0 + ci >= C || u0 + co >= C) continue;
      const int g = (u0 + ci) / GW;
      part[(((long)blockIdx.x * G + g) * 9 + tap) * GW * GW + (ci % GW) * GW + (co % GW)] = acc[tap][e];
    }
  }
}

// pixel slabs of the grouped-conv weight gradient: a workgroup is a chain of dependent 32-pixel steps (stage, barrier,
// MFMA), so slabs are short (>= 1024 pixels = 32 steps) and many -- the chip hides one chain's latency behind the others
extern "C" int tdeed_gconv_wgrad_slabs(long npix_out) {
  long s = (npix_out + 1023) / 1024;                           // >= 2: the head of `part` also carries the stride-2 input
  return (int)(s < 2 ? 2 : (s > 2048 ? 2048 : s));             // gradient's weight fragments (80 C floats)
}

static thread_local S2Stat g_s2_stat = S2Stat{};      // set by tdeed_gconv3x3_bwd_stats for the call it makes, cleared by that call

// dx [N][Hi][Wi][C] (activation dtype), dw fp32 [G][9][gw][gw] (the forward's packed layout);
// part: fp32 [tdeed_gconv_wgrad_slabs(N*Ho*Wo)][G*9*gw*gw]
extern "C" int tdeed_gconv3x3_bwd(const void* x, const void* dy, int N, int Hi, int Wi, int C, int gw, int stride,
                                  const float* w, const float* in_a, const float* in_b, void* dx, float* part, float* dw,
                                  int dtype, void* stream) {
  TD_CHECK(x && dy && w && part && dw, "gconv3x3_bwd: null pointer");      // dx may be NULL: weight gradient only
  TD_CHECK(!in_a == !in_b, "gconv3x3_bwd: in_a and in_b come together");
  TD_CHECK((gw == 8 || gw == 16) && C % gw == 0 && (stride == 1 || stride == 2) && N > 0 && Hi > 0 && Wi > 0,
           "gconv3x3_bwd: bad geometry");
  TD_CHECK(dtype == TDEED_F32 || dtype == TDEED_BF16, "gconv3x3_bwd: bad dtype %d", dtype);
  const int Ho = (Hi - 1) / stride + 1, Wo = (Wi - 1) / stride + 1, G = C / gw;
  const long npix_in = (long)N * Hi * Wi, npix_out = (long)N * Ho * Wo;
  const int slabs = tdeed_gconv_wgrad_slabs(npix_out);
  const long pps = ((npix_out + slabs - 1) / slabs + 31) / 32 * 32;
  const int nsl = (int)((npix_out + pps - 1) / pps);
  hipStream_t st = (hipStream_t)stream;
  dim3 gd((unsigned)((npix_in + 255) / 256), G), gwg(nsl, G);
  constexpr bool wg_valu = false;      // (A/B switches of rounds 2 - 4 retired in round 6; the forms they forced stay as the fp32 / fallback paths)
  const bool mfma_w = dtype == TDEED_BF16 && !wg_valu && C % 8 == 0;
  const dim3 gwm(nsl, (C + 15) / 16);
  constexpr bool dg_old = false;
  const size_t sm_s2 = (size_t)100 * (C + (dtype == TDEED_F32 ? 4 : 8)) * (dtype == TDEED_F32 ? 4 : 2) + (size_t)4 * 9 * gw * gw * 4;
  const bool dg_tiled = dx && stride == 2 && !dg_old && sm_s2 <= 120 * 1024 && Hi % 2 == 0 && Wi % 2 == 0 && N <= 65535;
  if (dg_tiled && sm_s2 > 64 * 1024) {
    static TdDevOnce attr_set;
    if (!attr_set.get()) {
      hipError_t e = hipFuncSetAttribute((const void*)gconv_dgrad_s2_kernel<bf16_t, 8>, hipFuncAttributeMaxDynamicSharedMemorySize, 120 * 1024);
      if (e == hipSuccess) e = hipFuncSetAttribute((const void*)gconv_dgrad_s2_kernel<bf16_t, 16>, hipFuncAttributeMaxDynamicSharedMemorySize, 120 * 1024);
      if (e == hipSuccess) e = hipFuncSetAttribute((const void*)gconv_dgrad_s2_kernel<float, 8>, hipFuncAttributeMaxDynamicSharedMemorySize, 120 * 1024);
      if (e == hipSuccess) e = hipFuncSetAttribute((const void*)gconv_dgrad_s2_kernel<float, 16>, hipFuncAttributeMaxDynamicSharedMemorySize, 120 * 1024);
      if (e != hipSuccess) { tdeed_set_error("gconv3x3_bwd: hipFuncSetAttribute: %s", hipGetErrorString(e)); return TDEED_ERR_RUNTIME; }
      attr_set.set();
    }
  }
  const dim3 gts2(cdiv(Wi, 16), cdiv(Hi, 16), N);
  // bf16: the MFMA form; its weight fragments are packed into the head of `part` (the weight-gradient launch that follows
  // on the same stream overwrites them only after this kernel has finished)
  constexpr bool dg_valu = false;
  const long frag_bytes = (long)((C + 15) / 16) * 5 * 64 * 16;
  const bool dg_mfma = dx && stride == 2 && dtype == TDEED_BF16 && !dg_old && !dg_valu && C % 8 == 0 && Hi % 2 == 0 && Wi % 2 == 0 &&
                       N <= 65535 && frag_bytes <= (long)slabs * G * 9 * gw * gw * 4;
  if (dg_mfma) {
    hipLaunchKernelGGL(gconv_dgrad_s2_pack_kernel, dim3((unsigned)cdiv((C + 15) / 16 * 320, 256)), dim3(256), 0, st, w, C, gw,
                       (bf16x8*)part);
    hipLaunchKernelGGL(gconv_dgrad_s2_mfma_kernel, dim3(cdiv(Hi, 16), N, cdiv(C, 64)), dim3(256), 0, st, (const bf16_t*)dy, Hi,
                       Wi, Ho, Wo, C, (const bf16x8*)part, (bf16_t*)dx, g_s2_stat);
  } else if (g_s2_stat.z) {
    g_s2_stat = S2Stat{};
    tdeed_set_error("gconv3x3_bwd_stats: the geometry is not served by the stride-2 MFMA input-gradient kernel");
    return TDEED_ERR_ARG;
  }
  g_s2_stat = S2Stat{};
#define TD_GC_LAUNCH(TT, GWv)                                                                                           \
  do {                                                                                                                  \
    if (dg_mfma) {                                                                                                      \
    } else if (dg_tiled)                                                                                                \
      hipLaunchKernelGGL((gconv_dgrad_s2_kernel<TT, GWv>), gts2, dim3(256), sm_s2, st, (const TT*)dy, Hi, Wi, Ho, Wo, C,  \
                         w, (TT*)dx);                                                                                   \
    else if (dx)                                                                                                        \
      hipLaunchKernelGGL((gconv_dgrad_kernel<TT, GWv>), gd, dim3(256), 0, st, (const TT*)dy, Hi, Wi, Ho, Wo, C, stride,  \
                         w, (TT*)dx, npix_in);                                                                          \
    if (!mfma_w)                                                                                                        \
      hipLaunchKernelGGL((gconv_wgrad_kernel<TT, GWv>), gwg, dim3(256), 0, st, (const TT*)x, (const TT*)dy, Hi, Wi, Ho,  \
                         Wo, C, stride, npix_out, pps, part);                                                           \
  } while (0)
  if (dtype == TDEED_F32) { if (gw == 8) TD_GC_LAUNCH(float, 8); else TD_GC_LAUNCH(float, 16); }
  else { if (gw == 8) TD_GC_LAUNCH(bf16_t, 8); else TD_GC_LAUNCH(bf16_t, 16); }
#undef TD_GC_LAUNCH
  int nrows = nsl;                                              // partial rows the launches below leave in `part`
  if (mfma_w) {
    constexpr bool wg_scatter = false;
    const long tiles = (long)N * cdiv(Ho, 8) * cdiv(Wo, 8);
    if (!wg_scatter && tiles < (1L << 22)) {                    // transposing LDS reads, one fetch per input pixel
      const int tpw = (int)((tiles + slabs - 1) / slabs);
      nrows = (int)((tiles + tpw - 1) / tpw);
      const dim3 gt((unsigned)nrows, (unsigned)cdiv(C, 64));
#define TD_GWT(Sv, GWv)                                                                                                 \
  do {                                                                                                                  \
    if (in_a)                                                                                                           \
      hipLaunchKernelGGL((gconv_wgrad_tr_kernel<Sv, GWv, true>), gt, dim3(256), 0, st, (const bf16_t*)x, (const bf16_t*)dy, \
                         in_a, in_b, N, Hi, Wi, Ho, Wo, C, tpw, part);                                                  \
    else                                                                                                                \
      hipLaunchKernelGGL((gconv_wgrad_tr_kernel<Sv, GWv, false>), gt, dim3(256), 0, st, (const bf16_t*)x, (const bf16_t*)dy, \
                         in_a, in_b, N, Hi, Wi, Ho, Wo, C, tpw, part);                                                  \
  } while (0)
      if (stride == 1) { if (gw == 8) TD_GWT(1, 8); else TD_GWT(1, 16); }
      else { if (gw == 8) TD_GWT(2, 8); else TD_GWT(2, 16); }
#undef TD_GWT
      in_a = nullptr;                                           // consumed
    } else if (gw == 8) {
      hipLaunchKernelGGL(gconv_wgrad_mfma_kernel<8>, gwm, dim3(256), 0, st, (const bf16_t*)x, (const bf16_t*)dy, Hi, Wi, Ho, Wo,
                         C, stride, npix_out, pps, part);
    } else {
      hipLaunchKernelGGL(gconv_wgrad_mfma_kernel<16>, gwm, dim3(256), 0, st, (const bf16_t*)x, (const bf16_t*)dy, Hi, Wi, Ho,
                         Wo, C, stride, npix_out, pps, part);
    }
  }
  TD_LAUNCH_CHECK("gconv3x3_bwd");
  TD_CHECK(!in_a, "gconv3x3_bwd: the on-load input affine needs the bf16 transposing-read weight-gradient kernel");
  return tdeed_reduce_partials(part, nrows, (long)G * 9 * gw * gw, dw, 0, stream);
}

// The same (stride 2, bf16) with the statistics of conv1's BatchNorm backward out of the input-gradient launch: part_s / part_q
// fp32 [N * tdeed_gconv3x3_bwd_stats_bands(Hi)][C] = sums of g and g * (bz - bmean), g = dx masked by [bfa bz + bfb > 0].
extern "C" int tdeed_gconv3x3_bwd_stats_bands(int Hi) { return cdiv(Hi, 16); }
// 1 when tdeed_gconv3x3_bwd would run the stride-2 MFMA input-gradient kernel for this geometry (the only one with the epilogue)
extern "C" int tdeed_gconv3x3_bwd_stats_fits(int N, int Hi, int Wi, int C, int gw) {
  if (!(gw == 8 || gw == 16) || C % gw != 0 || C % 8 != 0 || Hi % 2 != 0 || Wi % 2 != 0 || N <= 0 || N > 65535)
    return 0;
  const int Ho = (Hi - 1) / 2 + 1, Wo = (Wi - 1) / 2 + 1, G = C / gw;
  const int slabs = tdeed_gconv_wgrad_slabs((long)N * Ho * Wo);
  const long frag_bytes = (long)((C + 15) / 16) * 5 * 64 * 16;
  return frag_bytes <= (long)slabs * G * 9 * gw * gw * 4 ? 1 : 0;
}
extern "C" int tdeed_gconv3x3_bwd_stats(const void* x, const void* dy, int N, int Hi, int Wi, int C, int gw, const float* w,
                                        const float* in_a, const float* in_b, void* dx, float* part, float* dw, const void* bz,
                                        const float* bfa, const float* bfb, const float* bmean, float* part_s, float* part_q,
                                        void* stream) {
  TD_CHECK(dx && bz && bfa && bfb && bmean && part_s && part_q, "gconv3x3_bwd_stats: null pointer");
  g_s2_stat = S2Stat{(const bf16_t*)bz, bfa, bfb, bmean, part_s, part_q};
  const int rc = tdeed_gconv3x3_bwd(x, dy, N, Hi, Wi, C, gw, 2, w, in_a, in_b, dx, part, dw, TDEED_BF16, stream);
  g_s2_stat = S2Stat{};
  return rc;
}

// =========================================================================== row gather / scatter for stride-2 1x1 convs
// mode 0 (gather):  out[(f,yo,xo)][c]  = in[(f, yo*2, xo*2)][c]            (the shortcut conv's operand)
// mode 1 (scatter): out[(f, yo*2, xo*2)][c] += in[(f,yo,xo)][c]            (its input gradient, added to the main path's)
template <typename T>
__global__ __launch_bounds__(256) void stride2_rows_kernel(const T* __restrict__ in, T* __restrict__ out, int hi, int wi,
                                                           int ho, int wo, int C, int mode, long nchunks) {
  constexpr int EPC = Chunk<T>::N;
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= nchunks) return;
  const int nch = C / EPC;
  const int ck = (int)(i % nch);
  const long r = i / nch;
  const int xo = (int)(r % wo), yo = (int)((r / wo) % ho);
  const long f = r / ((long)wo * ho);
  const long big = ((f * hi + yo * 2) * wi + xo * 2) * C + ck * EPC, small_ = r * C + ck * EPC;
  float v[EPC];
  if (mode == 0) {
    Chunk<T>::load(in + big, v);
    Chunk<T>::store(out + small_, v);
  } else {
    float o[EPC];
    Chunk<T>::load(in + small_, v);
    Chunk<T>::load(out + big, o);
#pragma unroll
    for (int e = 0; e < EPC; ++e) o[e] += v[e];
    Chunk<T>::store(out + big, o);
  }
}

extern "C" int tdeed_stride2_rows(const void* in, void* out, int F, int hi, int wi, int C, int mode, int dtype,
                                  void* stream) {
  TD_CHECK(in && out && F > 0 && hi > 0 && wi > 0 && C % 8 == 0 && (mode == 0 || mode == 1), "stride2_rows: bad arguments");
  const int ho = (hi - 1) / 2 + 1, wo = (wi - 1) / 2 + 1;
  hipStream_t st = (hipStream_t)stream;
  if (dtype == TDEED_F32) {
    const long n = (long)F * ho * wo * (C / 4);
    hipLaunchKernelGGL(stride2_rows_kernel<float>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, (const float*)in,
                       (float*)out, hi, wi, ho, wo, C, mode, n);
  } else if (dtype == TDEED_BF16) {
    const long n = (long)F * ho * wo * (C / 8);
    hipLaunchKernelGGL(stride2_rows_kernel<bf16_t>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, (const bf16_t*)in,
                       (bf16_t*)out, hi, wi, ho, wo, C, mode, n);
  } else { tdeed_set_error("stride2_rows: bad dtype %d", dtype); return TDEED_ERR_ARG; }
  TD_LAUNCH_CHECK("stride2_rows");
  return TDEED_OK;
}

// =========================================================================== global average pool + positional encoding
// forward (avgpool_posenc): feat[b][t][c] = mean_p x[f][p][c] + temp_enc[t][c].  Backward:
//   d x[f][p][c] = d feat[f][c] / hw  (broadcast),   d temp_enc[t][c] = sum_b d feat[b][t][c]
template <typename T>
__global__ __launch_bounds__(256) void avgpool_bwd_kernel(const T* __restrict__ d_feat, int hw, int C, T* __restrict__ dx,
                                                          long nchunks, int nch) {
  constexpr int EPC = Chunk<T>::N;
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= nchunks) return;
  const int ck = (int)(i % nch);
  const long f = (i / nch) / hw;
  float v[EPC];
  Chunk<T>::load(d_feat + f * C + ck * EPC, v);
  const float inv = 1.0f / (float)hw;
#pragma unroll
  for (int e = 0; e < EPC; ++e) v[e] *= inv;
  Chunk<T>::store(dx + i * EPC, v);
}

template <typename T>
__global__ __launch_bounds__(256) void posenc_bwd_kernel(const T* __restrict__ d_feat, int B, long TC,
                                                         float* __restrict__ d_enc) {
  const long j = (long)blockIdx.x * 256 + threadIdx.x;
  if (j >= TC) return;
  float a = 0.f;
  for (int b = 0; b < B; ++b) a += (float)d_feat[(long)b * TC + j];
  d_enc[j] = a;
}

extern "C" int tdeed_avgpool_posenc_bwd(const void* d_feat, int B, int T, int hw, int C, void* dx, float* d_temp_enc,
                                        int dtype, void* stream) {
  TD_CHECK(d_feat && dx && d_temp_enc && B > 0 && T > 0 && hw > 0 && C % 8 == 0, "avgpool_posenc_bwd: bad arguments");
  hipStream_t st = (hipStream_t)stream;
  const long TC = (long)T * C;
  if (dtype == TDEED_F32) {
    const long n = (long)B * T * hw * (C / 4);
    hipLaunchKernelGGL(avgpool_bwd_kernel<float>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, (const float*)d_feat,
                       hw, C, (float*)dx, n, C / 4);
    hipLaunchKernelGGL(posenc_bwd_kernel<float>, dim3((unsigned)((TC + 255) / 256)), dim3(256), 0, st, (const float*)d_feat, B,
                       TC, d_temp_enc);
  } else if (dtype == TDEED_BF16) {
    const long n = (long)B * T * hw * (C / 8);
    hipLaunchKernelGGL(avgpool_bwd_kernel<bf16_t>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st,
                       (const bf16_t*)d_feat, hw, C, (bf16_t*)dx, n, C / 8);
    hipLaunchKernelGGL(posenc_bwd_kernel<bf16_t>, dim3((unsigned)((TC + 255) / 256)), dim3(256), 0, st,
                       (const bf16_t*)d_feat, B, TC, d_temp_enc);
  } else { tdeed_set_error("avgpool_posenc_bwd: bad dtype %d", dtype); return TDEED_ERR_ARG; }
  TD_LAUNCH_CHECK("avgpool_posenc_bwd");
  return TDEED_OK;
}

// =========================================================================== stem weight gradient
// forward (stem_kernel): z[n][oy][ox][co] = sum_{c,ky,kx} w[co][c][ky][kx] * in[c][2oy+ky-1][2ox+kx-1], in = the cropped,
// optionally flipped, /255 and ImageNet-standardised uint8 frame.  The input needs no gradient; the weight gradient is
// dw[co][27] = sum_{n,oy,ox} dz * in.  One workgroup per frame walks its 16x16 output tiles (patch and dz tile in LDS,
// lanes own (co, tap) pairs): part[n][32*27], folded by reduce_partials.
template <typename T, typename IN>
__global__ __launch_bounds__(256) void stem_wgrad_kernel(const IN* __restrict__ frames, int H, int W, int top, int left,
                                                         int ch, int cw, int flip_all,
                                                         const unsigned char* __restrict__ flip_mask,
                                                         const T* __restrict__ dz, int Ho, int Wo,
                                                         float* __restrict__ part) {
  __shared__ float tile[3][33][34];
  __shared__ float dzt[256][33];
  const int n = blockIdx.x;
  const int flip = flip_mask ? (int)flip_mask[n] : flip_all;
  const float mean[3] = {0.485f, 0.456f, 0.406f};
  const float stdv[3] = {0.229f, 0.224f, 0.225f};
  const IN* src = frames + (long)n * 3 * H * W;
  float acc[4] = {0.f, 0.f, 0.f, 0.f};                           // outputs o = tid + 256*i < 864
  {
    const int oy0 = blockIdx.y * 16;                            // one row of 16x16 tiles per workgroup
    for (int ox0 = 0; ox0 < Wo; ox0 += 16) {
      const int iy0 = oy0 * 2 - 1, ix0 = ox0 * 2 - 1;
      __syncthreads();
      for (int i = threadIdx.x; i < 3 * 33 * 33; i += 256) {
        const int c = i / (33 * 33);
        const int r = i - c * 33 * 33;
        const int y = r / 33, x = r - y * 33;
        const int iy = iy0 + y, ix = ix0 + x;
        float v = 0.f;
        if (iy >= 0 && iy < ch && ix >= 0 && ix < cw) {
          const int sx = flip ? (cw - 1 - ix) : ix;
          const float u = (float)src[((long)c * H + (top + iy)) * W + (left + sx)];
          v = (u / 255.0f - mean[c]) / stdv[c];
        }
        tile[c][y][x] = v;
      }
      for (int i = threadIdx.x; i < 256 * 32; i += 256) {
        const int p = i >> 5, co = i & 31;
        const int oy = oy0 + (p >> 4), ox = ox0 + (p & 15);
        dzt[p][co] = (oy < Ho && ox < Wo) ? (float)dz[(((long)n * Ho + oy) * Wo + ox) * 32 + co] : 0.f;
      }
      __syncthreads();
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int o = threadIdx.x + 256 * i;
        if (o < 864) {
          const int co = o / 27, k = o - co * 27;
          const int c = k / 9, ky = (k / 3) % 3, kx = k % 3;
          float a = acc[i];
          for (int p = 0; p < 256; ++p) a = fmaf(dzt[p][co], tile[c][2 * (p >> 4) + ky][2 * (p & 15) + kx], a);
          acc[i] = a;
        }
      }
    }
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int o = threadIdx.x + 256 * i;
    if (o < 864) part[((long)n * gridDim.y + blockIdx.y) * 864 + o] = acc[i];
  }
}

// bf16 form on the MFMA pipe: per 16x16 output tile the 256 pixels are the contraction index; dz^T [32 co][256 px] and
// the im2col patch^T [27 -> 32 taps][256 px] go to LDS as bf16 (pixel-contiguous rows), wave (co-tile, tap-tile) runs 8
// MFMAs per tile and keeps its 16x16 block of dW in registers across the tiles of the row.
constexpr int SW_LD = 256 + 8;
template <typename IN>
__global__ __launch_bounds__(256) void stem_wgrad_mfma_kernel(const IN* __restrict__ frames, int H, int W, int top, int left,
                                                              int ch, int cw, int flip_all,
                                                              const unsigned char* __restrict__ flip_mask,
                                                              const bf16_t* __restrict__ dz,
                                                              int Ho, int Wo, float* __restrict__ part) {
  __shared__ float tile[3][33][34];
  __shared__ __attribute__((aligned(16))) bf16_t dzT[32 * SW_LD];
  __shared__ __attribute__((aligned(16))) bf16_t inT[32 * SW_LD];
  const int n = blockIdx.x, oy0 = blockIdx.y * 16;
  const int flip = flip_mask ? (int)flip_mask[n] : flip_all;
  const float mean[3] = {0.485f, 0.456f, 0.406f};
  const float stdv[3] = {0.229f, 0.224f, 0.225f};
  const IN* src = frames + (long)n * 3 * H * W;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, pl = lane & 15, q = lane >> 4;
  const int ty = tid >> 4, tx = tid & 15;
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  for (int i = tid; i < 5 * SW_LD; i += 256) inT[27 * SW_LD + i] = (bf16_t)0.f;      // taps 27..31 stay zero
  for (int ox0 = 0; ox0 < Wo; ox0 += 16) {
    const int iy0 = oy0 * 2 - 1, ix0 = ox0 * 2 - 1;
    __syncthreads();
    for (int i = tid; i < 3 * 33 * 33; i += 256) {
      const int c = i / (33 * 33);
      const int r = i - c * 33 * 33;
      const int y = r / 33, x = r - y * 33;
      const int iy = iy0 + y, ix = ix0 + x;
      float v = 0.f;
      if (iy >= 0 && iy < ch && ix >= 0 && ix < cw) {
        const int sx = flip ? (cw - 1 - ix) : ix;
        v = ((float)src[((long)c * H + (top + iy)) * W + (left + sx)] / 255.0f - mean[c]) / stdv[c];
      }
      tile[c][y][x] = v;
    }
    {     // this lane's pixel: its 32 gradient channels, transposed into dzT
      const int oy = oy0 + ty, ox = ox0 + tx;
      const bool ok = oy < Ho && ox < Wo;
      const bf16_t* g = dz + (((long)n * Ho + (ok ? oy : 0)) * Wo + (ok ? ox : 0)) * 32;
      u32x4 v[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) v[j] = *reinterpret_cast<const u32x4*>(g + 8 * j);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const bf16x8 t8 = *reinterpret_cast<const bf16x8*>(&v[j]);
#pragma unroll
        for (int e = 0; e < 8; ++e) dzT[(8 * j + e) * SW_LD + tid] = ok ? t8[e] : (bf16_t)0.f;
      }
    }
    __syncthreads();
#pragma unroll
    for (int c = 0; c < 3; ++c)
#pragma unroll
      for (int ky = 0; ky < 3; ++ky)
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) inT[(c * 9 + ky * 3 + kx) * SW_LD + tid] = (bf16_t)tile[c][2 * ty + ky][2 * tx + kx];
    __syncthreads();
    const bf16_t* ar = dzT + ((wv & 1) * 16 + pl) * SW_LD + q * 8;
    const bf16_t* br = inT + ((wv >> 1) * 16 + pl) * SW_LD + q * 8;
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) {
      const bf16x8 af = *reinterpret_cast<const bf16x8*>(ar + ks * 32);
      const bf16x8 bf_ = *reinterpret_cast<const bf16x8*>(br + ks * 32);
      acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af, bf_, acc, 0, 0, 0);
    }
  }
  const int k = (wv >> 1) * 16 + pl;
  if (k < 27) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int co = (wv & 1) * 16 + 4 * q + e;
      part[((long)n * gridDim.y + blockIdx.y) * 864 + co * 27 + k] = acc[e];
    }
  }
}

// dz [N][Ho][Wo][32] (gradient of the raw stem conv output) -> dw [32][3][3][3] fp32; part fp32 [N*ceil(Ho/16)][864]
int td_stem_wgrad_tr_launch(const void* frames, int frames_f32, int N, int H, int W, int crop_top, int crop_left, int crop_h,
                            int crop_w, int flip, const unsigned char* flip_mask, const void* dz, float* part,
                            hipStream_t st, const void* bz, const float* bsums, const float* bmean, const float* brstd,
                            const float* bw);

extern "C" int tdeed_stem_wgrad(const void* frames, int frames_f32, int N, int H, int W, int crop_top, int crop_left,
                                int crop_h, int crop_w, int flip, const unsigned char* flip_mask, const void* dz,
                                float* part, float* dw, int dtype, void* stream) {
  TD_CHECK(frames && dz && part && dw, "stem_wgrad: null pointer");
  TD_CHECK(N > 0 && crop_h > 0 && crop_w > 0 && crop_top >= 0 && crop_left >= 0 && crop_top + crop_h <= H &&
               crop_left + crop_w <= W, "stem_wgrad: bad geometry");
  const int Ho = (crop_h + 1) / 2, Wo = (crop_w + 1) / 2;
  hipStream_t st = (hipStream_t)stream;
#define TD_SWG(TT, IN)                                                                                                 \
  hipLaunchKernelGGL((stem_wgrad_kernel<TT, IN>), dim3(N, cdiv(Ho, 16)), dim3(256), 0, st, (const IN*)frames, H, W, crop_top, \
                     crop_left, crop_h, crop_w, flip, flip_mask, (const TT*)dz, Ho, Wo, part)
  if (dtype == TDEED_F32) { if (frames_f32) TD_SWG(float, float); else TD_SWG(float, uint8_t); }
  else if (dtype == TDEED_BF16) {
    if (td_stem_wgrad_tr_launch(frames, frames_f32, N, H, W, crop_top, crop_left, crop_h, crop_w, flip, flip_mask,
                                                  dz, part, st, nullptr, nullptr, nullptr, nullptr, nullptr)) {
      // transposing-read form (front.hip): same partial layout
    }
    else if (frames_f32)
      hipLaunchKernelGGL(stem_wgrad_mfma_kernel<float>, dim3(N, cdiv(Ho, 16)), dim3(256), 0, st, (const float*)frames, H, W,
                         crop_top, crop_left, crop_h, crop_w, flip, flip_mask, (const bf16_t*)dz, Ho, Wo, part);
    else
      hipLaunchKernelGGL(stem_wgrad_mfma_kernel<uint8_t>, dim3(N, cdiv(Ho, 16)), dim3(256), 0, st, (const uint8_t*)frames, H,
                         W, crop_top, crop_left, crop_h, crop_w, flip, flip_mask, (const bf16_t*)dz, Ho, Wo, part);
  }
#undef TD_SWG
  else { tdeed_set_error("stem_wgrad: bad dtype %d", dtype); return TDEED_ERR_ARG; }
  TD_LAUNCH_CHECK("stem_wgrad");
  return tdeed_reduce_partials(part, N * cdiv(Ho, 16), 864, dw, 0, stream);
}

// tdeed_stem_wgrad with the stem BatchNorm's backward applied while the gradient rows are staged (bf16, the transposing-read
// kernel): g = the masked gradient at the BatchNorm's OUTPUT, z = the raw stem output, sums fp32 [2][32] = (sum g, sum g xhat)
// as tdeed_bn_bwd_from_parts leaves them, mean / rstd / w [32].  Returns TDEED_ERR_ARG when the geometry is not served
// (tdeed_stem_wgrad_bn_fits): the caller then applies the BatchNorm backward itself and calls tdeed_stem_wgrad.
extern "C" int tdeed_stem_wgrad_bn_fits(int H, int W, int crop_h, int crop_w) {
  const int WoP = ((crop_w + 1) / 2 + 31) / 32 * 32;
  const size_t patch_b = (size_t)((((9 * (crop_w + 2) + 2) * 4 + 7) & ~7)) * 2;
  const size_t smem = patch_b + (size_t)4 * WoP * 64;
  return (smem <= 80 * 1024 && ((size_t)8 * (crop_w + 2) + 2 * (size_t)WoP + 4) * 8 <= smem) ? 1 : 0;      // (front.hip: FRONT_LDS_CAP)
}
extern "C" int tdeed_stem_wgrad_bn(const void* frames, int frames_f32, int N, int H, int W, int crop_top, int crop_left,
                                   int crop_h, int crop_w, int flip, const unsigned char* flip_mask, const void* g,
                                   const void* z, const float* sums, const float* mean, const float* rstd, const float* w,
                                   float* part, float* dw, void* stream) {
  TD_CHECK(frames && g && z && sums && mean && rstd && w && part && dw, "stem_wgrad_bn: null pointer");
  TD_CHECK(N > 0 && crop_h > 0 && crop_w > 0 && crop_top >= 0 && crop_left >= 0 && crop_top + crop_h <= H &&
               crop_left + crop_w <= W, "stem_wgrad_bn: bad geometry");
  const int Ho = (crop_h + 1) / 2;
  TD_CHECK(td_stem_wgrad_tr_launch(frames, frames_f32, N, H, W, crop_top, crop_left, crop_h, crop_w, flip, flip_mask, g, part,
                                   (hipStream_t)stream, z, sums, mean, rstd, w),
           "stem_wgrad_bn: geometry not served (tdeed_stem_wgrad_bn_fits)");
  TD_LAUNCH_CHECK("stem_wgrad_bn");
  return tdeed_reduce_partials(part, N * cdiv(Ho, 16), 864, dw, 0, stream);
}

// =========================================================================== mixup of two uint8 clips
// out[b][i] = lam[b] * a[b][i] + (1 - lam[b]) * b[b][i]   (model.py:246: frame[i] = l*frame[i] + (1-l)*frame2[i]), fp32 out
__global__ __launch_bounds__(256) void mix_frames_kernel(const uint8_t* __restrict__ a, const uint8_t* __restrict__ b,
                                                         const float* __restrict__ lam, long per_clip, long total,
                                                         float* __restrict__ out) {
  const long i = ((long)blockIdx.x * 256 + threadIdx.x) * 4;
  if (i >= total) return;
  const float l = lam[i / per_clip];
  const uchar4 va = *reinterpret_cast<const uchar4*>(a + i), vb = *reinterpret_cast<const uchar4*>(b + i);
  f32x4 o = {l * (float)va.x + (1.f - l) * (float)vb.x, l * (float)va.y + (1.f - l) * (float)vb.y,
             l * (float)va.z + (1.f - l) * (float)vb.z, l * (float)va.w + (1.f - l) * (float)vb.w};
  *reinterpret_cast<f32x4*>(out + i) = o;
}

extern "C" int tdeed_mix_frames(const uint8_t* a, const uint8_t* b, const float* lam, int B, long per_clip, float* out,
                                void* stream) {
  TD_CHECK(a && b && lam && out && B > 0 && per_clip > 0 && per_clip % 4 == 0, "mix_frames: bad arguments");
  const long total = (long)B * per_clip;
  hipLaunchKernelGGL(mix_frames_kernel, dim3((unsigned)((total / 4 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, a, b,
                     lam, per_clip, total, out);
  TD_LAUNCH_CHECK("mix_frames");
  return TDEED_OK;
}
