// Backward of the SGP encoder-decoder pieces (reference forward: /root/reference/model/modules.py:58-363), NTC
// layout, activations/gradients in T (fp32 or bf16), parameter gradients always fp32.
// Parameter gradients are produced as per-workgroup partials and folded by reduce_partials in a fixed order, so a
// training step is bit-reproducible (no float atomics anywhere).
#include "common.h"
#include <cstdlib>
#include "sgp_tile.h"

// =========================================================================== small generic pieces
// out[j] = sum_p part[p][j]  (ordered), optionally out[j] += ...
// A workgroup owns CW consecutive columns; its 256 / CW row lanes each sum every (256/CW)-th partial (8 loads in flight
// per lane), then the lanes' sums are folded in a fixed order through LDS: same result on every run, and a fold over
// thousands of partials is a few microseconds instead of one serial chain per output.
template <int CW>
__global__ __launch_bounds__(256) void reduce_partials_kernel(const float* __restrict__ part, int P, long stride, long n,
                                                              float* __restrict__ out, int accumulate) {
  constexpr int PL = 256 / CW;
  __shared__ float red[PL][CW + 1];
  const int c = threadIdx.x % CW, pl = threadIdx.x / CW;
  const long j = (long)blockIdx.x * CW + c;
  const long jj = min(j, n - 1);
  float s = 0.f;
  for (int p0 = pl; p0 < P; p0 += 8 * PL) {
    float v[8];
#pragma unroll
    for (int b = 0; b < 8; ++b) v[b] = part[(long)min(p0 + b * PL, P - 1) * stride + jj];
#pragma unroll
    for (int b = 0; b < 8; ++b) s += p0 + b * PL < P ? v[b] : 0.f;
  }
  red[pl][c] = s;
  __syncthreads();
  if (pl == 0 && j < n) {
    float a = 0.f;
#pragma unroll
    for (int i = 0; i < PL; ++i) a += red[i][c];
    out[j] = accumulate ? out[j] + a : a;
  }
}

// few partials (P < 64): one thread per output (or per 4 outputs when rows are 16-byte aligned), the P rows summed in
// order with 8 loads in flight -- a 64-column x 4-lane workgroup per 64 outputs is pure dispatch overhead when a step folds
// tens of millions of weight-gradient elements from 3-4 partial slices each.  tdeed_multi_fold sums in the same order.
__device__ __forceinline__ float fold_seq1(const float* __restrict__ src, int P, long pstride, long j) {
  float s = 0.f;
  for (int p0 = 0; p0 < P; p0 += 8) {
    float v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = src[(long)min(p0 + u, P - 1) * pstride + j];
#pragma unroll
    for (int u = 0; u < 8; ++u) s += p0 + u < P ? v[u] : 0.f;
  }
  return s;
}
__device__ __forceinline__ f32x4 fold_seq4(const float* __restrict__ src, int P, long pstride, long j) {
  f32x4 s = {0.f, 0.f, 0.f, 0.f};
  for (int p0 = 0; p0 < P; p0 += 4) {
    f32x4 v[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) v[u] = *reinterpret_cast<const f32x4*>(src + (long)min(p0 + u, P - 1) * pstride + j);
#pragma unroll
    for (int u = 0; u < 4; ++u)
      if (p0 + u < P) s += v[u];
  }
  return s;
}
__global__ __launch_bounds__(256) void reduce_seq_kernel(const float* __restrict__ part, int P, long stride, long n,
                                                         float* __restrict__ out, int accumulate, int vec) {
  if (vec) {
    const long j = ((long)blockIdx.x * 256 + threadIdx.x) * 4;
    if (j >= n) return;
    f32x4 s = fold_seq4(part, P, stride, j);
    if (accumulate) s += *reinterpret_cast<const f32x4*>(out + j);
    *reinterpret_cast<f32x4*>(out + j) = s;
  } else {
    const long j = (long)blockIdx.x * 256 + threadIdx.x;
    if (j >= n) return;
    const float s = fold_seq1(part, P, stride, j);
    out[j] = accumulate ? out[j] + s : s;
  }
}

static void launch_reduce(const float* part, int P, long stride, long n, float* out, int accumulate, hipStream_t st) {
  if (P < 64) {
    const int vec = (n % 4 == 0 && stride % 4 == 0 && ((uintptr_t)part % 16) == 0 && ((uintptr_t)out % 16) == 0) ? 1 : 0;
    const long per = vec ? 1024 : 256;
    hipLaunchKernelGGL(reduce_seq_kernel, dim3((unsigned)((n + per - 1) / per)), dim3(256), 0, st, part, P, stride, n, out,
                       accumulate, vec);
    return;
  }
  // many partials of a narrow output: 8 columns x 32 lanes (more workgroups, shorter chains), else 32 columns x 8 lanes
  if (P >= 512 && n <= 4096)
    hipLaunchKernelGGL(reduce_partials_kernel<8>, dim3((unsigned)((n + 7) / 8)), dim3(256), 0, st, part, P, stride, n, out, accumulate);
  else
    hipLaunchKernelGGL(reduce_partials_kernel<32>, dim3((unsigned)((n + 31) / 32)), dim3(256), 0, st, part, P, stride, n, out, accumulate);
}

extern "C" int tdeed_reduce_partials(const float* part, int P, long n, float* out, int accumulate, void* stream) {
  TD_CHECK(part && out && P > 0 && n > 0, "reduce_partials: bad arguments");
  launch_reduce(part, P, n, n, out, accumulate, (hipStream_t)stream);
  TD_LAUNCH_CHECK("reduce_partials");
  return TDEED_OK;
}

// rows of `part` hold several parameter groups side by side: fold n columns of rows that are `stride` floats apart
extern "C" int tdeed_reduce_strided(const float* part, int P, long stride, long n, float* out, void* stream) {
  TD_CHECK(part && out && P > 0 && n > 0 && stride >= n, "reduce_strided: bad arguments");
  launch_reduce(part, P, stride, n, out, 0, (hipStream_t)stream);
  TD_LAUNCH_CHECK("reduce_strided");
  return TDEED_OK;
}

// gradient write-out: a table of (source pointer, destination offset in `dst`, element count) for every parameter
// tensor of a step; dst[off + i] = (accumulate ? dst[off + i] : 0) + scale * src[i].  One launch for all tensors (a
// per-tensor device copy is ~3 us; a 200MF step has 431 of them).  Workgroup b serves chunk b of the concatenated ranges
// (chunk = 4096 elements; `first_chunk[t]` = first chunk of tensor t, binary-searched).
struct CopyEnt { const float* src; long off; long n; long first_chunk; };
__global__ __launch_bounds__(256) void multi_copy_kernel(const CopyEnt* __restrict__ tab, int nt, float* __restrict__ dst,
                                                         float scale, int accumulate) {
  const long b = blockIdx.x;
  int lo = 0, hi = nt - 1;
  while (lo < hi) {                                              // last entry with first_chunk <= b
    const int mid = (lo + hi + 1) >> 1;
    if (tab[mid].first_chunk <= b) lo = mid; else hi = mid - 1;
  }
  const CopyEnt e = tab[lo];
  const long i0 = (b - e.first_chunk) * 4096;
  for (long i = i0 + threadIdx.x; i < min(e.n, i0 + 4096); i += 256) {
    const float v = scale * e.src[i];
    dst[e.off + i] = accumulate ? dst[e.off + i] + v : v;
  }
}

// The same write-out with the fold of per-workgroup partials inside: entry = {src, dst offset, n, first workgroup,
// P | cw << 32, pstride, cols, ld}.  P == 1: a copy of n elements whose source is `cols` contiguous elements per row, rows
// `ld` apart (cols == n: contiguous), 4096 elements per workgroup.  P > 1: dst[off + j] = scale * sum_p src[p * pstride + j]
// (the weight-gradient kernels' partials), cw columns per workgroup, 256 / cw lanes walking the P rows (ordered fold through
// LDS: the same sum as tdeed_reduce_partials).  One launch per gradient bucket instead of one fold per parameter tensor.
struct FoldEnt { const float* src; long off; long n; long first_wg; long pcw; long pstride; long cols; long ld; };
__global__ __launch_bounds__(256) void multi_fold_kernel(const FoldEnt* __restrict__ tab, int nt, float* __restrict__ dst,
                                                         float scale, int accumulate) {
  __shared__ float red[32 * 65];
  const long b = blockIdx.x;
  int lo = 0, hi = nt - 1;
  while (lo < hi) {                                              // last entry with first_wg <= b
    const int mid = (lo + hi + 1) >> 1;
    if (tab[mid].first_wg <= b) lo = mid; else hi = mid - 1;
  }
  const FoldEnt e = tab[lo];
  const long w = b - e.first_wg;
  const int P = (int)(e.pcw & 0xffffffffL), cw = (int)(e.pcw >> 32);
  if (P <= 1) {
    const long i0 = w * 4096, i1 = min(e.n, i0 + 4096);
    for (long i = i0 + threadIdx.x; i < i1; i += 256) {
      long si = i;
      if (e.cols != e.n) {
        const long r = i / e.cols;
        si = r * e.ld + (i - r * e.cols);
      }
      const float v = scale * e.src[si];
      dst[e.off + i] = accumulate ? dst[e.off + i] + v : v;
    }
    return;
  }
  if (cw == 1024) {                                              // few partials, 16-byte aligned rows: 4 outputs per thread
    const long j = w * 1024 + threadIdx.x * 4;
    if (j >= e.n) return;
    f32x4 sv = fold_seq4(e.src, P, e.pstride, j);
    sv *= scale;
    if (accumulate) sv += *reinterpret_cast<const f32x4*>(dst + e.off + j);
    *reinterpret_cast<f32x4*>(dst + e.off + j) = sv;
    return;
  }
  if (cw == 256) {                                               // few partials: one output per thread
    const long j = w * 256 + threadIdx.x;
    if (j >= e.n) return;
    const float a = scale * fold_seq1(e.src, P, e.pstride, j);
    dst[e.off + j] = accumulate ? dst[e.off + j] + a : a;
    return;
  }
  const int PL = 256 / cw;
  const int c = threadIdx.x % cw, pl = threadIdx.x / cw;
  const long j = w * cw + c, jj = min(j, e.n - 1);
  float s = 0.f;
  for (int p0 = pl; p0 < P; p0 += 8 * PL) {
    float v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = e.src[(long)min(p0 + u * PL, P - 1) * e.pstride + jj];
#pragma unroll
    for (int u = 0; u < 8; ++u) s += p0 + u * PL < P ? v[u] : 0.f;
  }
  red[pl * (cw + 1) + c] = s;
  __syncthreads();
  if (pl == 0 && j < e.n) {
    float a = 0.f;
    for (int i = 0; i < PL; ++i) a += red[i * (cw + 1) + c];
    a *= scale;
    dst[e.off + j] = accumulate ? dst[e.off + j] + a : a;
  }
}

// columns per workgroup of a fold entry (the host sizes the table with it): many partials of a narrow output -> 8
// columns x 32 lanes, else 32 x 8 or 64 x 4 (the same rule as tdeed_reduce_partials)
extern "C" int tdeed_multi_fold_cw(int P, long n) {
  if (P <= 1) return 4096;
  if (P < 64) return n % 4 == 0 ? 1024 : 256;                    // sequential folds (the caller's rows are 16-byte aligned)
  if (P >= 512 && n <= 4096) return 8;
  return 32;
}

extern "C" int tdeed_multi_fold(const void* tab, int nt, long n_wgs, float* dst, float scale, int accumulate, void* stream) {
  TD_CHECK(tab && dst && nt > 0 && n_wgs > 0 && n_wgs < 0x7fffffffL, "multi_fold: bad arguments");
  hipLaunchKernelGGL(multi_fold_kernel, dim3((unsigned)n_wgs), dim3(256), 0, (hipStream_t)stream, (const FoldEnt*)tab, nt,
                     dst, scale, accumulate);
  TD_LAUNCH_CHECK("multi_fold");
  return TDEED_OK;
}

// tab: device array of nt entries {src pointer, dst offset, n, first chunk}; n_chunks = total chunks (sum of ceil(n / 4096))
extern "C" int tdeed_multi_copy(const void* tab, int nt, long n_chunks, float* dst, float scale, int accumulate,
                                void* stream) {
  TD_CHECK(tab && dst && nt > 0 && n_chunks > 0 && n_chunks < 0x7fffffffL, "multi_copy: bad arguments");
  hipLaunchKernelGGL(multi_copy_kernel, dim3((unsigned)n_chunks), dim3(256), 0, (hipStream_t)stream, (const CopyEnt*)tab, nt,
                     dst, scale, accumulate);
  TD_LAUNCH_CHECK("multi_copy");
  return TDEED_OK;
}

// elementwise: mode 0: y = gelu(x);  1: y = dy * gelu'(x);  2: y = x + dy (gradient accumulation);  3: y = x * dy (dropout mask)
template <typename T>
__global__ __launch_bounds__(256) void eltwise_kernel(const T* __restrict__ x, const T* __restrict__ dy, T* __restrict__ y,
                                                      long nchunks, int mode) {
  constexpr int EPC = Chunk<T>::N;
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= nchunks) return;
  float a[EPC], b[EPC], o[EPC];
  Chunk<T>::load(x + i * EPC, a);
  if (mode != 0) Chunk<T>::load(dy + i * EPC, b);
#pragma unroll
  for (int e = 0; e < EPC; ++e) {
    if (mode == 0) {
      o[e] = gelu_erf(a[e]);
    } else if (mode == 1) {
      const float cdf = 0.5f * (1.0f + erff(a[e] * 0.70710678118654752440f));
      const float pdf = 0.39894228040143267794f * expf(-0.5f * a[e] * a[e]);
      o[e] = b[e] * (cdf + a[e] * pdf);
    } else if (mode == 2) {
      o[e] = a[e] + b[e];
    } else {
      o[e] = a[e] * b[e];
    }
  }
  Chunk<T>::store(y + i * EPC, o);
}

extern "C" int tdeed_eltwise(const void* x, const void* dy, void* y, long n, int mode, int dtype, void* stream) {
  TD_CHECK(x && y && (mode == 0 || dy) && n > 0 && n % 8 == 0 && mode >= 0 && mode <= 3, "eltwise: bad arguments");
  hipStream_t st = (hipStream_t)stream;
  if (dtype == TDEED_F32) {
    const long nc = n / 4;
    hipLaunchKernelGGL(eltwise_kernel<float>, dim3((unsigned)((nc + 255) / 256)), dim3(256), 0, st, (const float*)x,
                       (const float*)dy, (float*)y, nc, mode);
  } else if (dtype == TDEED_BF16) {
    const long nc = n / 8;
    hipLaunchKernelGGL(eltwise_kernel<bf16_t>, dim3((unsigned)((nc + 255) / 256)), dim3(256), 0, st, (const bf16_t*)x,
                       (const bf16_t*)dy, (bf16_t*)y, nc, mode);
  } else { tdeed_set_error("eltwise: bad dtype %d", dtype); return TDEED_ERR_ARG; }
  TD_LAUNCH_CHECK("eltwise");
  return TDEED_OK;
}

// [R][Cc] -> [Cc][R] through a padded 32x32 LDS tile
template <typename T>
__global__ __launch_bounds__(256) void transpose_kernel(const T* __restrict__ x, int R, int Cc, T* __restrict__ y) {
  __shared__ float tile[32][33];
  const int c0 = blockIdx.x * 32, r0 = blockIdx.y * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  for (int i = ty; i < 32; i += 8)
    tile[i][tx] = (r0 + i < R && c0 + tx < Cc) ? (float)x[(long)(r0 + i) * Cc + c0 + tx] : 0.f;
  __syncthreads();
  for (int i = ty; i < 32; i += 8)
    if (c0 + i < Cc && r0 + tx < R) y[(long)(c0 + i) * R + r0 + tx] = (T)tile[tx][i];
}

extern "C" int tdeed_transpose(const void* x, int R, int Cc, void* y, int dtype, void* stream) {
  TD_CHECK(x && y && R > 0 && Cc > 0, "transpose: bad arguments");
  dim3 grid(cdiv(Cc, 32), cdiv(R, 32));
  hipStream_t st = (hipStream_t)stream;
  if (dtype == TDEED_F32)
    hipLaunchKernelGGL(transpose_kernel<float>, grid, dim3(256), 0, st, (const float*)x, R, Cc, (float*)y);
  else if (dtype == TDEED_BF16)
    hipLaunchKernelGGL(transpose_kernel<bf16_t>, grid, dim3(256), 0, st, (const bf16_t*)x, R, Cc, (bf16_t*)y);
  else { tdeed_set_error("transpose: bad dtype %d", dtype); return TDEED_ERR_ARG; }
  TD_LAUNCH_CHECK("transpose");
  return TDEED_OK;
}

// =========================================================================== weight / bias gradients of a 1x1
// dW[n][k] = sum_m dY[m][n] * X[m][k]   (+ db[n] = sum_m dY[m][n] from the k-tile-0 workgroups)
// 64 x 64 output tile per workgroup, 4 x 4 outputs per lane, rows m staged 32 at a time as fp32 in LDS; blockIdx.z
// slices M, slices land in `part` ([Z][N][K] then [Z][N] for the bias) and are folded by reduce_partials.
template <typename T>
__global__ __launch_bounds__(256) void wgrad_kernel(const T* __restrict__ dY, long ldy, const T* __restrict__ X, long ldx,
                                                    int M, int N, int K, float* __restrict__ part_w,
                                                    float* __restrict__ part_b) {
  __shared__ __attribute__((aligned(16))) float sy[32][64];
  __shared__ __attribute__((aligned(16))) float sx[32][64];
  const int n0 = blockIdx.x * 64, k0 = blockIdx.y * 64, z = blockIdx.z, Z = gridDim.z;
  const int mper = ((M + Z - 1) / Z + 31) / 32 * 32;
  const int m_begin = z * mper, m_end = min(M, m_begin + mper);
  const int tn = threadIdx.x >> 4, tk = threadIdx.x & 15;
  float acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = 0.f;
  float bsum = 0.f;                                             // lanes 0..63 of k-tile 0: column sums of dY
  for (int m0 = m_begin; m0 < m_end; m0 += 32) {
    __syncthreads();
    for (int i = threadIdx.x; i < 32 * 64; i += 256) {
      const int r = i >> 6, c = i & 63;
      const bool rok = m0 + r < m_end;
      sy[r][c] = (rok && n0 + c < N) ? (float)dY[(long)(m0 + r) * ldy + n0 + c] : 0.f;
      sx[r][c] = (rok && k0 + c < K) ? (float)X[(long)(m0 + r) * ldx + k0 + c] : 0.f;
    }
    __syncthreads();
#pragma unroll 8
    for (int r = 0; r < 32; ++r) {
      const f32x4 a = *reinterpret_cast<const f32x4*>(&sy[r][tn * 4]);
      const f32x4 b = *reinterpret_cast<const f32x4*>(&sx[r][tk * 4]);
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = fmaf(a[i], b[j], acc[i][j]);
    }
    if (part_b && blockIdx.y == 0 && threadIdx.x < 64)
      for (int r = 0; r < 32; ++r) bsum += sy[r][threadIdx.x];
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int n = n0 + tn * 4 + i;
    if (n >= N) continue;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int k = k0 + tk * 4 + j;
      if (k < K) part_w[((long)z * N + n) * K + k] = acc[i][j];
    }
  }
  if (part_b && blockIdx.y == 0 && threadIdx.x < 64 && n0 + threadIdx.x < N)
    part_b[(long)z * N + n0 + threadIdx.x] = bsum;
}

// bf16 form on the MFMA pipe.  The contraction runs over the ROWS of both operands, so a 32-row chunk is transposed on
// its way into LDS (sT[col][row], row stride 36 elements: 8-byte aligned fragment reads, column groups on different
// banks); a fragment is then 8 consecutive rows of one column.  dY^T is the MFMA A operand, X^T the B operand:
// D[n][k], lanes of a wave hold consecutive k => coalesced partial stores.  The next chunk's global loads are issued
// before the MFMAs of the current one.
constexpr int WG_LD = 36;
__global__ __launch_bounds__(256) void wgrad_mfma_kernel(const bf16_t* __restrict__ dY, long ldy,
                                                         const bf16_t* __restrict__ X, long ldx, int M, int N, int K,
                                                         float* __restrict__ part_w, float* __restrict__ part_b) {
  __shared__ __attribute__((aligned(16))) bf16_t sTy[64 * WG_LD];
  __shared__ __attribute__((aligned(16))) bf16_t sTx[64 * WG_LD];
  __shared__ float sb[32][65];
  const int n0 = blockIdx.x * 64, k0 = blockIdx.y * 64, z = blockIdx.z, Z = gridDim.z;
  const int mper = ((M + Z - 1) / Z + 31) / 32 * 32;
  const int m_begin = z * mper, m_end = min(M, m_begin + mper);
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, pl = lane & 15, q = lane >> 4;
  const int r = tid >> 3, c8 = (tid & 7) * 8;
  const bool nok = n0 + c8 < N, kok = k0 + c8 < K;
  const bool want_b = part_b && blockIdx.y == 0;
  f32x4 acc[4];
#pragma unroll
  for (int kt = 0; kt < 4; ++kt) acc[kt] = (f32x4){0.f, 0.f, 0.f, 0.f};
  float bsum[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  const long ycol = min(n0 + c8, N - 8), xcol = min(k0 + c8, K - 8);
  u32x4 vy, vx;
  auto issue = [&](int m0) {
    const long row = min(m0 + r, M - 1);
    vy = *reinterpret_cast<const u32x4*>(dY + row * ldy + ycol);
    vx = *reinterpret_cast<const u32x4*>(X + row * ldx + xcol);
  };
  if (m_begin < m_end) issue(m_begin);
  for (int m0 = m_begin; m0 < m_end; m0 += 32) {
    const bool rok = m0 + r < m_end;
    const bf16x8 ty = *reinterpret_cast<const bf16x8*>(&vy);
    const bf16x8 tx = *reinterpret_cast<const bf16x8*>(&vx);
    __syncthreads();
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const bf16_t a = (rok && nok) ? ty[e] : (bf16_t)0.f;
      sTy[(c8 + e) * WG_LD + r] = a;
      sTx[(c8 + e) * WG_LD + r] = (rok && kok) ? tx[e] : (bf16_t)0.f;
      if (want_b) bsum[e] += (float)a;
    }
    __syncthreads();
    if (m0 + 32 < m_end) issue(m0 + 32);
    bf16x8 af;
    {
      const bf16x4 lo = *reinterpret_cast<const bf16x4*>(sTy + (wv * 16 + pl) * WG_LD + q * 8);
      const bf16x4 hi = *reinterpret_cast<const bf16x4*>(sTy + (wv * 16 + pl) * WG_LD + q * 8 + 4);
      af = (bf16x8){lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    }
#pragma unroll
    for (int kt = 0; kt < 4; ++kt) {
      const bf16x4 lo = *reinterpret_cast<const bf16x4*>(sTx + (kt * 16 + pl) * WG_LD + q * 8);
      const bf16x4 hi = *reinterpret_cast<const bf16x4*>(sTx + (kt * 16 + pl) * WG_LD + q * 8 + 4);
      const bf16x8 bfr = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
      acc[kt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af, bfr, acc[kt], 0, 0, 0);
    }
  }
#pragma unroll
  for (int kt = 0; kt < 4; ++kt) {
    const int k = k0 + kt * 16 + pl;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int n = n0 + wv * 16 + 4 * q + e;
      if (n < N && k < K) part_w[((long)z * N + n) * K + k] = acc[kt][e];
    }
  }
  if (want_b) {
    __syncthreads();
#pragma unroll
    for (int e = 0; e < 8; ++e) sb[r][c8 + e] = bsum[e];
    __syncthreads();
    if (tid < 64 && n0 + tid < N) {
      float a = 0.f;
      for (int rr = 0; rr < 32; ++rr) a += sb[rr][tid];
      part_b[(long)z * N + n0 + tid] = a;
    }
  }
}

// The same contraction with transposing LDS reads (gfx950 ds_read_b64_tr_b16): 64-row chunks of dY and X go to LDS as
// they are (16-byte vector writes instead of eight 2-byte scatter writes per piece); an MFMA operand -- one column, 8
// consecutive rows -- is two transposing reads (per 16-lane group the hardware returns lane i column i of 4 rows).
// Half the barriers and a quarter of the LDS instructions per row of the kernel above.
constexpr int WT_RS = 80;                                       // LDS row stride in elements: 64 columns + 32 bytes
// X0 (optional): the columns k < k0 of the X operand come from X0 (row stride ldx0) instead of X -- the gate-shift splice of
// a s3 / s4 conv1 (shift.py:89-93): [G | x[:, Fp:]] is never materialised.
__global__ __launch_bounds__(256) void wgrad_tr_kernel(const bf16_t* __restrict__ dY, long ldy, const bf16_t* __restrict__ X,
                                                       long ldx, const bf16_t* __restrict__ X0, long ldx0, int k0s, int M, int N,
                                                       int K, float* __restrict__ part_w, float* __restrict__ part_b) {
  __shared__ __attribute__((aligned(16))) bf16_t sY[64 * WT_RS];
  __shared__ __attribute__((aligned(16))) bf16_t sX[64 * WT_RS];
  __shared__ float sb[32][65];
  const int n0 = blockIdx.x * 64, k0 = blockIdx.y * 64, z = blockIdx.z, Z = gridDim.z;
  const int mper = ((M + Z - 1) / Z + 63) / 64 * 64;
  const int m_begin = z * mper, m_end = min(M, m_begin + mper);
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, pl = lane & 15;
  const int g4 = lane >> 4, q4 = (lane >> 2) & 3, p4 = lane & 3;
  const int r = tid >> 3, c8 = (tid & 7) * 8;
  const bool nok = n0 + c8 < N, kok = k0 + c8 < K;
  const bool want_b = part_b && blockIdx.y == 0;
  f32x4 acc[4];
#pragma unroll
  for (int kt = 0; kt < 4; ++kt) acc[kt] = (f32x4){0.f, 0.f, 0.f, 0.f};
  float bsum[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  const long ycol = min(n0 + c8, N - 8), xcol = min(k0 + c8, K - 8);
  // the pad columns are never read (a unit reads columns < 64), rows are rewritten every chunk: no initial fill needed
  u32x4 vy[2], vx[2];
  auto issue = [&](int m0) {
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const long row = min(m0 + r + 32 * h, M - 1);
      vy[h] = *reinterpret_cast<const u32x4*>(dY + row * ldy + ycol);
      vx[h] = *reinterpret_cast<const u32x4*>(xcol < k0s ? X0 + row * ldx0 + xcol : X + row * ldx + xcol);
    }
  };
  if (m_begin < m_end) issue(m_begin);
  for (int m0 = m_begin; m0 < m_end; m0 += 64) {
    __syncthreads();
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const bool rok = m0 + r + 32 * h < m_end;
      const u32x4 zy = (rok && nok) ? vy[h] : (u32x4){0u, 0u, 0u, 0u};
      *reinterpret_cast<u32x4*>(sY + (r + 32 * h) * WT_RS + c8) = zy;
      *reinterpret_cast<u32x4*>(sX + (r + 32 * h) * WT_RS + c8) = (rok && kok) ? vx[h] : (u32x4){0u, 0u, 0u, 0u};
      if (want_b) {
        const bf16x8 t8 = *reinterpret_cast<const bf16x8*>(&zy);
#pragma unroll
        for (int e = 0; e < 8; ++e) bsum[e] += (float)t8[e];
      }
    }
    __syncthreads();
    if (m0 + 64 < m_end) issue(m0 + 64);
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const int row = ks * 32 + g4 * 8 + q4;
      const bf16x8 af = td_tr_read8(sY + row * WT_RS + wv * 16 + p4 * 4, sY + (row + 4) * WT_RS + wv * 16 + p4 * 4);
#pragma unroll
      for (int kt = 0; kt < 4; ++kt) {
        const bf16x8 bfr = td_tr_read8(sX + row * WT_RS + kt * 16 + p4 * 4, sX + (row + 4) * WT_RS + kt * 16 + p4 * 4);
        acc[kt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af, bfr, acc[kt], 0, 0, 0);
      }
    }
  }
#pragma unroll
  for (int kt = 0; kt < 4; ++kt) {
    const int k = k0 + kt * 16 + pl;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int n = n0 + wv * 16 + 4 * g4 + e;
      if (n < N && k < K) part_w[((long)z * N + n) * K + k] = acc[kt][e];
    }
  }
  if (want_b) {
    __syncthreads();
#pragma unroll
    for (int e = 0; e < 8; ++e) sb[r][c8 + e] = bsum[e];
    __syncthreads();
    if (tid < 64 && n0 + tid < N) {
      float a = 0.f;
      for (int rr = 0; rr < 32; ++rr) a += sb[rr][tid];
      part_b[(long)z * N + n0 + tid] = a;
    }
  }
}

// 128 x 128 output tile of the same contraction for the wide layers (N, K >= 96): waves 2 x 2, each 64 x 64 (16 accumulator
// tiles), so a 64-row chunk feeds 32 MFMAs per wave between two barriers instead of 8, and the operands are re-read by
// 3 x 3 instead of 5 x 5 workgroups at N = K = 320 (the 64 x 64 form ran at 229 TFLOP/s and 1.4 TB/s on M = 313600,
// N = K = 320: bound by neither).
constexpr int WT_RS2 = 144;                                     // LDS row stride in elements: 128 columns + 32 bytes
__global__ __launch_bounds__(256) void wgrad_tr128_kernel(const bf16_t* __restrict__ dY, long ldy, const bf16_t* __restrict__ X,
                                                          long ldx, const bf16_t* __restrict__ X0, long ldx0, int k0s, int M, int N,
                                                          int K, float* __restrict__ part_w, float* __restrict__ part_b) {
  __shared__ __attribute__((aligned(16))) bf16_t sY[64 * WT_RS2];
  __shared__ __attribute__((aligned(16))) bf16_t sX[64 * WT_RS2];
  // 1-D grid, renumbered: the tiles of one row slice sit on one XCD and share its L2 (see wgrad_tr160_kernel)
  const int ntx_ = (N + 127) >> 7, nty_ = (K + 127) >> 7, ntl_ = ntx_ * nty_;
  const long lid_ = xcd_logical_id(blockIdx.x, gridDim.x);
  const int tile_ = (int)(lid_ % ntl_), z = (int)(lid_ / ntl_), Z = (int)(gridDim.x / ntl_);
  const int bx_ = tile_ % ntx_, by_ = tile_ / ntx_;
  const int n0 = bx_ * 128, k0 = by_ * 128;
  const int mper = ((M + Z - 1) / Z + 63) / 64 * 64;
  const int m_begin = z * mper, m_end = min(M, m_begin + mper);
  const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6), pl = lane & 15;
  const int wn = wv >> 1, wk = wv & 1;
  const int g4 = lane >> 4, q4 = (lane >> 2) & 3, p4 = lane & 3;
  const int r = tid >> 4, c8 = (tid & 15) * 8;                  // staging: rows r + 16 h, 16-byte chunk c8
  const bool nok = n0 + c8 < N, kok = k0 + c8 < K;
  const bool want_b = part_b && by_ == 0;
  f32x4 acc[4][4];
#pragma unroll
  for (int nt = 0; nt < 4; ++nt)
#pragma unroll
    for (int kt = 0; kt < 4; ++kt) acc[nt][kt] = (f32x4){0.f, 0.f, 0.f, 0.f};
  float bsum[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  const long ycol = min(n0 + c8, N - 8), xcol = min(k0 + c8, K - 8);
  u32x4 vy[4], vx[4];
  auto issue = [&](int m0) {
#pragma unroll
    for (int h = 0; h < 4; ++h) {
      const long row = min(m0 + r + 16 * h, M - 1);
      vy[h] = *reinterpret_cast<const u32x4*>(dY + row * ldy + ycol);
      vx[h] = *reinterpret_cast<const u32x4*>(xcol < k0s ? X0 + row * ldx0 + xcol : X + row * ldx + xcol);
    }
  };
  if (m_begin < m_end) issue(m_begin);
  for (int m0 = m_begin; m0 < m_end; m0 += 64) {
    __syncthreads();
#pragma unroll
    for (int h = 0; h < 4; ++h) {
      const bool rok = m0 + r + 16 * h < m_end;
      const u32x4 zy = (rok && nok) ? vy[h] : (u32x4){0u, 0u, 0u, 0u};
      *reinterpret_cast<u32x4*>(sY + (r + 16 * h) * WT_RS2 + c8) = zy;
      *reinterpret_cast<u32x4*>(sX + (r + 16 * h) * WT_RS2 + c8) = (rok && kok) ? vx[h] : (u32x4){0u, 0u, 0u, 0u};
      if (want_b) {
        const bf16x8 t8 = *reinterpret_cast<const bf16x8*>(&zy);
#pragma unroll
        for (int e = 0; e < 8; ++e) bsum[e] += (float)t8[e];
      }
    }
    __syncthreads();
    if (m0 + 64 < m_end) issue(m0 + 64);
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const int row = ks * 32 + g4 * 8 + q4;
      bf16x8 af[4], bfr[4];
#pragma unroll
      for (int nt = 0; nt < 4; ++nt) {
        const bf16_t* a = sY + row * WT_RS2 + wn * 64 + nt * 16 + p4 * 4;
        af[nt] = td_tr_read8(a, a + 4 * WT_RS2);
      }
#pragma unroll
      for (int kt = 0; kt < 4; ++kt) {
        const bf16_t* b = sX + row * WT_RS2 + wk * 64 + kt * 16 + p4 * 4;
        bfr[kt] = td_tr_read8(b, b + 4 * WT_RS2);
      }
#pragma unroll
      for (int nt = 0; nt < 4; ++nt)
#pragma unroll
        for (int kt = 0; kt < 4; ++kt) acc[nt][kt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[nt], bfr[kt], acc[nt][kt], 0, 0, 0);
    }
  }
#pragma unroll
  for (int nt = 0; nt < 4; ++nt)
#pragma unroll
    for (int kt = 0; kt < 4; ++kt) {
      const int k = k0 + wk * 64 + kt * 16 + pl;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int n = n0 + wn * 64 + nt * 16 + 4 * g4 + e;
        if (n < N && k < K) part_w[((long)z * N + n) * K + k] = acc[nt][kt][e];
      }
    }
  if (want_b) {
    __syncthreads();
    float* sb = reinterpret_cast<float*>(sY);                   // [16][129]
#pragma unroll
    for (int e = 0; e < 8; ++e) sb[r * 129 + c8 + e] = bsum[e];
    __syncthreads();
    if (tid < 128 && n0 + tid < N) {
      float a = 0.f;
      for (int rr = 0; rr < 16; ++rr) a += sb[rr * 129 + tid];
      part_b[(long)z * N + n0 + tid] = a;
    }
  }
}

// 160 x 160 output tile for N = K = 320 (the s3 layers of RegNetY-800MF: 16 launches per step over 313 600 rows): two exact
// tiles per dimension instead of three 128-wide ones padded to 384 -- the operands are re-read 2 x instead of 3 x and no MFMA
// runs on padding.  Waves 2 x 2, each 80 x 80 (25 accumulator tiles); otherwise wgrad_tr128_kernel.
constexpr int WT_RS3 = 176;                                     // LDS row stride in elements: 160 columns + 32 bytes
__global__ __launch_bounds__(256) void wgrad_tr160_kernel(const bf16_t* __restrict__ dY, long ldy, const bf16_t* __restrict__ X,
                                                          long ldx, const bf16_t* __restrict__ X0, long ldx0, int k0s, int M, int N,
                                                          int K, float* __restrict__ part_w) {
  __shared__ __attribute__((aligned(16))) bf16_t sY[64 * WT_RS3];
  __shared__ __attribute__((aligned(16))) bf16_t sX[64 * WT_RS3];
  // 1-D grid of 4 Z workgroups, renumbered so that the four output tiles of one row slice run on ONE XCD next to each other:
  // each half of dY / X is read by two of them, the second time from that XCD's L2 (as a (2, 2, Z) grid the hardware dealt the
  // four over four XCDs and every operand byte crossed from HBM twice: 800 MB per call instead of 400)
  const long lid_ = xcd_logical_id(blockIdx.x, gridDim.x);
  const int tile_ = (int)(lid_ & 3), z = (int)(lid_ >> 2), Z = (int)(gridDim.x >> 2);
  const int n0 = (tile_ & 1) * 160, k0 = (tile_ >> 1) * 160;
  const int mper = ((M + Z - 1) / Z + 63) / 64 * 64;
  const int m_begin = z * mper, m_end = min(M, m_begin + mper);
  const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6), pl = lane & 15;
  const int wn = wv >> 1, wk = wv & 1;
  const int g4 = lane >> 4, q4 = (lane >> 2) & 3, p4 = lane & 3;
  // staging: 64 rows x 20 16-byte chunks per operand = 1280 pieces = 5 per thread: piece i = tid + 256 j -> row i / 20, chunk i % 20
  // (no bias partial here: conv layers have none, and its 40 registers would cost the second wave per SIMD)
  f32x4 acc[5][5];
#pragma unroll
  for (int nt = 0; nt < 5; ++nt)
#pragma unroll
    for (int kt = 0; kt < 5; ++kt) acc[nt][kt] = (f32x4){0.f, 0.f, 0.f, 0.f};
  int prow[5], pcol[5];
#pragma unroll
  for (int j = 0; j < 5; ++j) {
    const int i = tid + 256 * j;
    prow[j] = i / 20;
    pcol[j] = (i - prow[j] * 20) * 8;
  }
  u32x4 vy[5], vx[5];
  auto issue = [&](int m0) {
#pragma unroll
    for (int j = 0; j < 5; ++j) {
      const long row = min(m0 + prow[j], M - 1);
      const long yc = n0 + pcol[j], xc = k0 + pcol[j];
      vy[j] = *reinterpret_cast<const u32x4*>(dY + row * ldy + yc);
      vx[j] = *reinterpret_cast<const u32x4*>(xc < k0s ? X0 + row * ldx0 + xc : X + row * ldx + xc);
    }
  };
  if (m_begin < m_end) issue(m_begin);
  for (int m0 = m_begin; m0 < m_end; m0 += 64) {
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 5; ++j) {
      const bool rok = m0 + prow[j] < m_end;
      *reinterpret_cast<u32x4*>(sY + prow[j] * WT_RS3 + pcol[j]) = rok ? vy[j] : (u32x4){0u, 0u, 0u, 0u};
      *reinterpret_cast<u32x4*>(sX + prow[j] * WT_RS3 + pcol[j]) = rok ? vx[j] : (u32x4){0u, 0u, 0u, 0u};
    }
    __syncthreads();
    if (m0 + 64 < m_end) issue(m0 + 64);
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const int row = ks * 32 + g4 * 8 + q4;
      bf16x8 af[5], bfr[5];
#pragma unroll
      for (int nt = 0; nt < 5; ++nt) {
        const bf16_t* a = sY + row * WT_RS3 + wn * 80 + nt * 16 + p4 * 4;
        af[nt] = td_tr_read8(a, a + 4 * WT_RS3);
      }
#pragma unroll
      for (int kt = 0; kt < 5; ++kt) {
        const bf16_t* b = sX + row * WT_RS3 + wk * 80 + kt * 16 + p4 * 4;
        bfr[kt] = td_tr_read8(b, b + 4 * WT_RS3);
      }
#pragma unroll
      for (int nt = 0; nt < 5; ++nt)
#pragma unroll
        for (int kt = 0; kt < 5; ++kt) acc[nt][kt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[nt], bfr[kt], acc[nt][kt], 0, 0, 0);
    }
  }
#pragma unroll
  for (int nt = 0; nt < 5; ++nt)
#pragma unroll
    for (int kt = 0; kt < 5; ++kt) {
      const int k = k0 + wk * 80 + kt * 16 + pl;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int n = n0 + wn * 80 + nt * 16 + 4 * g4 + e;
        if (n < N && k < K) part_w[((long)z * N + n) * K + k] = acc[nt][kt][e];
      }
    }
}

static bool wgrad_wide(int M, int N, int K) {
  return N >= 96 && K >= 96 && M >= 4096;
}

// number of M slices: enough workgroups that every CU holds several (each one is a chain of dependent 32-row steps:
// latency hidden by its neighbours), without slices shorter than 256 rows or more than 32 MB of partials
static bool wgrad_160(int M, int N, int K) {
  return N == 320 && K == 320 && M >= 4096;
}

extern "C" int tdeed_wgrad_slices(int M, int N, int K) {
  const int tw = wgrad_160(M, N, K) ? 160 : (wgrad_wide(M, N, K) ? 128 : 64);
  const long tiles = (long)((N + tw - 1) / tw) * ((K + tw - 1) / tw);
  long z = ((tw >= 128 ? 1024 : 2048) + tiles - 1) / tiles;
  const long zmax = M >= 8192 ? (M + 255) / 256 : (M + 63) / 64;       // few rows (SE / head layers): 64-row slices
  if (z > zmax) z = zmax;
  const long zbytes = (32L << 20) / ((long)N * K * 4);
  if (z > zbytes) z = zbytes;
  if (z > 2048) z = 2048;
  return (int)(z < 1 ? 1 : z);
}

// part_w: fp32 [Z][N][K], part_b: fp32 [Z][N] or NULL, Z = tdeed_wgrad_slices(M, N, K); dW [N][K], db [N] (fp32)
extern "C" int tdeed_wgrad(const void* dY, long ldy, const void* X, long ldx, const void* X0, long ldx0, int k0, int M,
                           int N, int K, float* part_w, float* part_b, float* dW, float* db, int accumulate, int dtype,
                           void* stream) {
  TD_CHECK(!X0 || (k0 > 0 && k0 % 8 == 0 && k0 <= K && ldx0 % 8 == 0 && dtype == TDEED_BF16 && M >= 4096),
           "wgrad: the spliced X operand needs bf16, M >= 4096 and k0 a multiple of 8");
  TD_CHECK(dY && X && part_w && (dW || accumulate < 0) && (!db || part_b), "wgrad: null pointer");
  TD_CHECK(M > 0 && N > 0 && K > 0, "wgrad: bad sizes");
  const int Z = tdeed_wgrad_slices(M, N, K);
  dim3 grid(cdiv(N, 64), cdiv(K, 64), Z);
  hipStream_t st = (hipStream_t)stream;
  if (dtype == TDEED_F32)
    hipLaunchKernelGGL(wgrad_kernel<float>, grid, dim3(256), 0, st, (const float*)dY, ldy, (const float*)X, ldx, M, N, K,
                       part_w, (db || accumulate < 0) ? part_b : nullptr);
  else if (dtype == TDEED_BF16) {
    constexpr bool valu = false, scatter = false;      // (A/B switches of rounds 2 - 4, retired in round 6: the forms below won)
    const bool vec_ok = N % 8 == 0 && K % 8 == 0 && N >= 8 && K >= 8 && ldy % 8 == 0 && ldx % 8 == 0;
    TD_CHECK(!X0 || (!valu && vec_ok && !scatter), "wgrad: the spliced X operand needs the transposing-read kernel");
    if (!valu && vec_ok && !scatter && wgrad_160(M, N, K) && !part_b)      // N = K = 320 (no bias): two exact 160-wide tiles per dimension
      hipLaunchKernelGGL(wgrad_tr160_kernel, dim3(4 * Z), dim3(256), 0, st, (const bf16_t*)dY, ldy, (const bf16_t*)X, ldx,
                         (const bf16_t*)X0, ldx0, X0 ? k0 : 0, M, N, K, part_w);
    else if (!valu && vec_ok && !scatter && wgrad_wide(M, N, K))     // wide layers: 128 x 128 output tiles
      hipLaunchKernelGGL(wgrad_tr128_kernel, dim3((unsigned)(cdiv(N, 128) * cdiv(K, 128) * Z)), dim3(256), 0, st, (const bf16_t*)dY, ldy,
                         (const bf16_t*)X, ldx, (const bf16_t*)X0, ldx0, X0 ? k0 : 0, M, N, K, part_w,
                         (db || accumulate < 0) ? part_b : nullptr);
    else if (!valu && vec_ok && !scatter && M >= 4096)          // long contractions: transposing LDS reads, 64-row chunks
      hipLaunchKernelGGL(wgrad_tr_kernel, grid, dim3(256), 0, st, (const bf16_t*)dY, ldy, (const bf16_t*)X, ldx,
                         (const bf16_t*)X0, ldx0, X0 ? k0 : 0, M, N, K, part_w, (db || accumulate < 0) ? part_b : nullptr);
    else if (!valu && vec_ok)
      hipLaunchKernelGGL(wgrad_mfma_kernel, grid, dim3(256), 0, st, (const bf16_t*)dY, ldy, (const bf16_t*)X, ldx, M, N, K,
                         part_w, (db || accumulate < 0) ? part_b : nullptr);
    else
      hipLaunchKernelGGL(wgrad_kernel<bf16_t>, grid, dim3(256), 0, st, (const bf16_t*)dY, ldy, (const bf16_t*)X, ldx, M, N,
                         K, part_w, (db || accumulate < 0) ? part_b : nullptr);
  }
  else { tdeed_set_error("wgrad: bad dtype %d", dtype); return TDEED_ERR_ARG; }
  TD_LAUNCH_CHECK("wgrad");
  if (accumulate < 0) return TDEED_OK;           // partials only: folded later (tdeed_multi_fold, with the gradient write-out)
  int rc = tdeed_reduce_partials(part_w, Z, (long)N * K, dW, accumulate, stream);
  if (rc == TDEED_OK && db) rc = tdeed_reduce_partials(part_b, Z, N, db, accumulate, stream);
  return rc;
}

// =========================================================================== channel LayerNorm backward
// one wave per row, LNB_ROWS rows per workgroup; dx = rstd * (g - mean(g) - xhat * mean(g * xhat)), g = dy * w;
// part[blk][0][c] = sum_rows dy * xhat (d weight), part[blk][1][c] = sum_rows dy (d bias)
constexpr int LNB_ROWS = 8;      // (32 until round 4: 1600 rows were 50 workgroups of 8 dependent row rounds per wave)
template <typename T>
__global__ __launch_bounds__(256) void layernorm_bwd_kernel(const T* __restrict__ x, long ldx, const T* __restrict__ dy,
                                                            long ldy, int rows, int C, const float* __restrict__ w,
                                                            float eps, T* __restrict__ dx, int accumulate,
                                                            float* __restrict__ part) {
  constexpr int EPC = Chunk<T>::N;
  constexpr int MAXCH = 4;
  extern __shared__ float sred[];                              // [4 waves][2][C]
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int nch = C / EPC;
  float dg[MAXCH][EPC], db[MAXCH][EPC], wr[MAXCH][EPC];
#pragma unroll
  for (int i = 0; i < MAXCH; ++i) {
    const int ck = min(lane + 64 * i, nch - 1);
#pragma unroll
    for (int e = 0; e < EPC; ++e) { dg[i][e] = 0.f; db[i][e] = 0.f; wr[i][e] = w[ck * EPC + e]; }
  }
  for (int rr = wv; rr < LNB_ROWS; rr += 4) {
    const long row = (long)blockIdx.x * LNB_ROWS + rr;
    if (row >= rows) break;                                     // uniform per wave
    float xv[MAXCH][EPC], gv[MAXCH][EPC];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < MAXCH; ++i) {
      const int ck = min(lane + 64 * i, nch - 1);
      Chunk<T>::load(x + row * ldx + (long)ck * EPC, xv[i]);
      Chunk<T>::load(dy + row * ldy + (long)ck * EPC, gv[i]);
      if (lane + 64 * i < nch) {
#pragma unroll
        for (int e = 0; e < EPC; ++e) s += xv[i][e];
      }
    }
    const float mu = wave_sum(s) / (float)C;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < MAXCH; ++i)
      if (lane + 64 * i < nch) {
#pragma unroll
        for (int e = 0; e < EPC; ++e) { xv[i][e] -= mu; q += xv[i][e] * xv[i][e]; }
      }
    const float rstd = 1.0f / sqrtf(wave_sum(q) / (float)C + eps);
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int i = 0; i < MAXCH; ++i)
      if (lane + 64 * i < nch) {
#pragma unroll
        for (int e = 0; e < EPC; ++e) {
          xv[i][e] *= rstd;                                     // xhat
          dg[i][e] += gv[i][e] * xv[i][e];
          db[i][e] += gv[i][e];
          gv[i][e] *= wr[i][e];                                 // g = dy * w
          s1 += gv[i][e];
          s2 += gv[i][e] * xv[i][e];
        }
      }
    s1 = wave_sum(s1) / (float)C;
    s2 = wave_sum(s2) / (float)C;
#pragma unroll
    for (int i = 0; i < MAXCH; ++i) {
      const int ck = lane + 64 * i;
      if (ck < nch) {
        float o[EPC];
#pragma unroll
        for (int e = 0; e < EPC; ++e) o[e] = rstd * (gv[i][e] - s1 - xv[i][e] * s2);
        if (accumulate) {
          float old[EPC];
          Chunk<T>::load(dx + row * ldx + (long)ck * EPC, old);
#pragma unroll
          for (int e = 0; e < EPC; ++e) o[e] += old[e];
        }
        Chunk<T>::store(dx + row * ldx + (long)ck * EPC, o);
      }
    }
  }
#pragma unroll
  for (int i = 0; i < MAXCH; ++i) {
    const int ck = lane + 64 * i;
    if (ck < nch) {
#pragma unroll
      for (int e = 0; e < EPC; ++e) {
        sred[(wv * 2 + 0) * C + ck * EPC + e] = dg[i][e];
        sred[(wv * 2 + 1) * C + ck * EPC + e] = db[i][e];
      }
    }
  }
  __syncthreads();
  for (int j = threadIdx.x; j < 2 * C; j += 256) {
    float v = 0.f;
#pragma unroll
    for (int w2 = 0; w2 < 4; ++w2) v += sred[w2 * 2 * C + j];
    part[(long)blockIdx.x * 2 * C + j] = v;
  }
}

extern "C" int tdeed_layernorm_bwd_blocks(int rows) { return (rows + LNB_ROWS - 1) / LNB_ROWS; }

// part: fp32 [blocks][2][C]; dw, db: fp32 [C]; dx has x's geometry (row stride ldx); accumulate: dx += instead of =
extern "C" int tdeed_layernorm_bwd(const void* x, long ldx, const void* dy, long ldy, int rows, int C, const float* w,
                                   float eps, void* dx, int accumulate, float* part, float* dw, float* db, int dtype,
                                   void* stream) {
  // dw == db == NULL: partials only (part [blocks][2][C]; the caller folds them, e.g. with the gradient write-out)
  TD_CHECK(x && dy && w && dx && part && (!dw == !db), "layernorm_bwd: null pointer");
  TD_CHECK(rows > 0 && C > 0 && C % 8 == 0 && ldx % 8 == 0 && ldy % 8 == 0, "layernorm_bwd: bad sizes");
  const int nb = tdeed_layernorm_bwd_blocks(rows);
  const size_t smem = (size_t)8 * C * sizeof(float);
  hipStream_t st = (hipStream_t)stream;
  if (dtype == TDEED_F32) {
    TD_CHECK(C <= 4 * 64 * 4, "layernorm_bwd: C=%d too wide", C);
    hipLaunchKernelGGL(layernorm_bwd_kernel<float>, dim3(nb), dim3(256), smem, st, (const float*)x, ldx, (const float*)dy,
                       ldy, rows, C, w, eps, (float*)dx, accumulate, part);
  } else if (dtype == TDEED_BF16) {
    TD_CHECK(C <= 4 * 64 * 8, "layernorm_bwd: C=%d too wide", C);
    hipLaunchKernelGGL(layernorm_bwd_kernel<bf16_t>, dim3(nb), dim3(256), smem, st, (const bf16_t*)x, ldx,
                       (const bf16_t*)dy, ldy, rows, C, w, eps, (bf16_t*)dx, accumulate, part);
  } else { tdeed_set_error("layernorm_bwd: bad dtype %d", dtype); return TDEED_ERR_ARG; }
  TD_LAUNCH_CHECK("layernorm_bwd");
  if (!dw) return TDEED_OK;
  // part viewed as [nb][2C]: the first C columns of a row are d weight, the next C are d bias
  launch_reduce(part, nb, 2L * C, (long)C, dw, 0, st);
  launch_reduce(part + C, nb, 2L * C, (long)C, db, 0, st);
  TD_LAUNCH_CHECK("layernorm_bwd reduce");
  return TDEED_OK;
}

// =========================================================================== GroupNorm backward
// one workgroup per (clip, group): slab [T][cg] of x and dy cached in LDS; statistics recomputed;
// dx = rstd * (g - mean(g) - xhat * mean(g * xhat)), g = dy * w over the slab;
// part[b][0][c] = sum_t dy * xhat, part[b][1][c] = sum_t dy
template <typename T>
__global__ __launch_bounds__(256) void groupnorm_bwd_kernel(const T* __restrict__ x, const T* __restrict__ dy, int T_len,
                                                            int C, int G, const float* __restrict__ w, float eps,
                                                            T* __restrict__ dx, int accumulate,
                                                            float* __restrict__ part) {
  extern __shared__ float sm[];     // xs [n], gs [n], col [2][cg], scratch [8]
  const int b = blockIdx.x, g = blockIdx.y;
  const int cg = C / G;
  const int n = T_len * cg;
  float* xs = sm;
  float* gs = xs + n;
  float* col = gs + n;
  float* scratch = col + 2 * cg;
  const long base = (long)b * T_len * C + g * cg;
  const IDiv dcg(cg);
  float s = 0.f;
  for (int i = threadIdx.x; i < n; i += 256) {
    int t, cl;
    dcg.divmod(i, t, cl);
    const float v = (float)x[base + (long)t * C + cl];
    xs[i] = v;
    gs[i] = (float)dy[base + (long)t * C + cl];
    s += v;
  }
  const float mean = block_sum<4>(s, scratch) / (float)n;
  float q = 0.f;
  for (int i = threadIdx.x; i < n; i += 256) {
    const float d = xs[i] - mean;
    q += d * d;
  }
  const float rstd = 1.0f / sqrtf(block_sum<4>(q, scratch) / (float)n + eps);
  float s1 = 0.f, s2 = 0.f;
  for (int i = threadIdx.x; i < n; i += 256) {
    int t, cl;
    dcg.divmod(i, t, cl);
    const float xh = (xs[i] - mean) * rstd;
    xs[i] = xh;
    const float gg = gs[i] * w[g * cg + cl];
    s1 += gg;
    s2 += gg * xh;
  }
  s1 = block_sum<4>(s1, scratch) / (float)n;
  s2 = block_sum<4>(s2, scratch) / (float)n;
  for (int i = threadIdx.x; i < n; i += 256) {
    int t, cl;
    dcg.divmod(i, t, cl);
    const float gg = gs[i] * w[g * cg + cl];
    float o = rstd * (gg - s1 - xs[i] * s2);
    T* dst = dx + base + (long)t * C + cl;
    if (accumulate) o += (float)*dst;
    *dst = (T)o;
  }
  // per-channel sums over t (ordered): lane cl walks its column
  for (int cl = threadIdx.x; cl < cg; cl += 256) {
    float a = 0.f, c2 = 0.f;
    for (int t = 0; t < T_len; ++t) {
      a += gs[t * cg + cl] * xs[t * cg + cl];
      c2 += gs[t * cg + cl];
    }
    part[((long)b * 2 + 0) * C + g * cg + cl] = a;
    part[((long)b * 2 + 1) * C + g * cg + cl] = c2;
  }
}

// part: fp32 [B][2][C]; dw, db: fp32 [C]
extern "C" int tdeed_groupnorm_bwd(const void* x, const void* dy, int B, int T, int C, int G, const float* w, float eps,
                                   void* dx, int accumulate, float* part, float* dw, float* db, int dtype,
                                   void* stream) {
  TD_CHECK(x && dy && w && dx && part && (!dw == !db), "groupnorm_bwd: null pointer");
  TD_CHECK(B > 0 && T > 0 && G > 0 && C % G == 0, "groupnorm_bwd: bad sizes");
  const size_t smem = ((size_t)2 * T * (C / G) + 2 * (C / G) + 8) * sizeof(float);
  TD_CHECK(smem <= 150 * 1024, "groupnorm_bwd: slab too large");
  static TdDevOnce attr_set;
  if (!attr_set.get()) {      // long clips with wide groups (T=250, 48 channels per group: 96 KB) exceed the default 64 KB
    hipError_t e = hipFuncSetAttribute((const void*)groupnorm_bwd_kernel<float>, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
    if (e == hipSuccess) e = hipFuncSetAttribute((const void*)groupnorm_bwd_kernel<bf16_t>, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
    if (e != hipSuccess) { tdeed_set_error("groupnorm_bwd: hipFuncSetAttribute: %s", hipGetErrorString(e)); return TDEED_ERR_RUNTIME; }
    attr_set.set();
  }
  dim3 grid(B, G);
  hipStream_t st = (hipStream_t)stream;
  if (dtype == TDEED_F32)
    hipLaunchKernelGGL(groupnorm_bwd_kernel<float>, grid, dim3(256), smem, st, (const float*)x, (const float*)dy, T, C, G,
                       w, eps, (float*)dx, accumulate, part);
  else if (dtype == TDEED_BF16)
    hipLaunchKernelGGL(groupnorm_bwd_kernel<bf16_t>, grid, dim3(256), smem, st, (const bf16_t*)x, (const bf16_t*)dy, T, C,
                       G, w, eps, (bf16_t*)dx, accumulate, part);
  else { tdeed_set_error("groupnorm_bwd: bad dtype %d", dtype); return TDEED_ERR_ARG; }
  TD_LAUNCH_CHECK("groupnorm_bwd");
  if (!dw) return TDEED_OK;                                      // partials only (see tdeed_layernorm_bwd)
  // part viewed as [B][2C]: the first C columns of each row are d weight, the next C are d bias
  launch_reduce(part, B, 2L * C, (long)C, dw, 0, st);
  launch_reduce(part + C, B, 2L * C, (long)C, db, 0, st);
  TD_LAUNCH_CHECK("groupnorm_bwd reduce");
  return TDEED_OK;
}

// =========================================================================== depthwise-branch backward
// Forward (per clip b, channel c; o = LayerNorm output):
//   psi = dw_ks(o), cw = dw_ks(o), ckw = dw_up(o), fc = wf*o + bf, phi = relu(wg * mean_t(o) + bg)
//   out = fc * phi + (cw + ckw) * psi + o
// Given g = d out: everything is per channel, so a workgroup owns 16 channels of one clip exactly like the forward
// kernel: o and g tiles in LDS, the forward products recomputed, d psi / d (cw+ckw) staged with a zero halo so the
// input gradient is three more temporal correlations.  Weight gradients: one lane per (tap, channel) walks T.
template <typename T>
__device__ __forceinline__ void branch_bwd_core(const float* ot, const float* gconv, const float* ginst,
                                                const float* gid, float* dpsi, float* ds,
                                                const float* wl, const Bias5& bb, bool cok, int T_len, int halo, int ks,
                                                int up, float* red /*[17][16]*/, float* red5 /*[16][5][16]*/,
                                                float mean_c, T* __restrict__ d_o, long ld_o, int c0, int C,
                                                float* __restrict__ part_w /*[C][wlen] slice of this clip*/,
                                                float* __restrict__ part_b /*[5][C] slice of this clip*/, float* res) {
  const int c = threadIdx.x & 15, tl = threadIdx.x >> 4;
  const int wlen = 2 * ks + up + 2;
  const int hk = ks >> 1, hu = up >> 1;
  const float wf = wl[(2 * ks + up) * SGP_CH + c], wg = wl[(2 * ks + up + 1) * SGP_CH + c];
  const float b_psi = cok ? bb.psi : 0.f, b_cw = cok ? bb.cw : 0.f, b_ckw = cok ? bb.ckw : 0.f,
              b_fc = cok ? bb.fc : 0.f, b_g = cok ? bb.g : 0.f;
  const float pre = fmaf(wg, mean_c, b_g);
  const float phi = fmaxf(pre, 0.f);
  // ---- phase A: forward products again, d psi and d (cw + ckw) into LDS, per-lane partial sums
  float a_dphi = 0.f, a_dpsi = 0.f, a_ds = 0.f, a_g = 0.f, a_go = 0.f;
  for (int t = tl; t < T_len; t += 16) {
    const float* col = ot + (halo + t) * SGP_CH + c;
    float psi = b_psi, cw = b_cw, ckw = b_ckw;
    for (int k = 0; k < ks; ++k) {
      const float v = col[(k - hk) * SGP_CH];
      psi = fmaf(wl[k * SGP_CH + c], v, psi);
      cw = fmaf(wl[(ks + k) * SGP_CH + c], v, cw);
    }
    for (int k = 0; k < up; ++k) ckw = fmaf(wl[(2 * ks + k) * SGP_CH + c], col[(k - hu) * SGP_CH], ckw);
    const float o = col[0];
    const float fc = fmaf(wf, o, b_fc);
    const float gc = gconv[t * SGP_CH + c], gi = ginst[t * SGP_CH + c];
    const float vdpsi = gc * (cw + ckw), vds = gc * psi;
    dpsi[(halo + t) * SGP_CH + c] = vdpsi;
    ds[(halo + t) * SGP_CH + c] = vds;
    a_dphi += gi * fc;
    a_dpsi += vdpsi;
    a_ds += vds;
    a_g += gi;
    a_go += gi * o;
  }
  red5[(tl * 5 + 0) * SGP_CH + c] = a_dphi;
  red5[(tl * 5 + 1) * SGP_CH + c] = a_dpsi;
  red5[(tl * 5 + 2) * SGP_CH + c] = a_ds;
  red5[(tl * 5 + 3) * SGP_CH + c] = a_g;
  red5[(tl * 5 + 4) * SGP_CH + c] = a_go;
  __syncthreads();
  if (threadIdx.x < 5 * SGP_CH) {                              // lane (which, c): ordered sum over the 16 time lanes
    const int which = threadIdx.x >> 4, cc = threadIdx.x & 15;
    float a = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) a += red5[(i * 5 + which) * SGP_CH + cc];
    red[which * SGP_CH + cc] = a;
  }
  __syncthreads();
  const float dphi = red[0 * SGP_CH + c], s_dpsi = red[1 * SGP_CH + c], s_ds = red[2 * SGP_CH + c],
              s_g = red[3 * SGP_CH + c], s_go = red[4 * SGP_CH + c];
  const float dpre = pre > 0.f ? dphi : 0.f;
  const float dmean = dpre * wg / (float)T_len;
  // ---- phase B: input gradient
  for (int t = tl; t < T_len; t += 16) {
    float a = gid[t * SGP_CH + c] + ginst[t * SGP_CH + c] * (wf * phi) + dmean;
    const float* dp = dpsi + (halo + t) * SGP_CH + c;
    const float* dq = ds + (halo + t) * SGP_CH + c;
    for (int k = 0; k < ks; ++k) {
      a = fmaf(wl[k * SGP_CH + c], dp[(hk - k) * SGP_CH], a);
      a = fmaf(wl[(ks + k) * SGP_CH + c], dq[(hk - k) * SGP_CH], a);
    }
    for (int k = 0; k < up; ++k) a = fmaf(wl[(2 * ks + k) * SGP_CH + c], dq[(hu - k) * SGP_CH], a);
    res[t * SGP_CH + c] = a;
  }
  __syncthreads();
  store_tile<T>(res, d_o, ld_o, 0, T_len, c0, C, (const T*)nullptr, 0);
  // ---- phase C: weight gradients, one lane per (tap, channel)
  const int ntap = 2 * ks + up;
  for (int i = threadIdx.x; i < ntap * SGP_CH; i += 256) {
    const int k = i >> 4, cc = i & 15;
    const float* dsrc = k < ks ? dpsi : ds;
    const int off = k < ks ? k - hk : (k < 2 * ks ? k - ks - hk : k - 2 * ks - hu);
    float a = 0.f;
    for (int t = 0; t < T_len; ++t) a = fmaf(dsrc[(halo + t) * SGP_CH + cc], ot[(halo + t + off) * SGP_CH + cc], a);
    if (c0 + cc < C) part_w[(long)(c0 + cc) * wlen + k] = a;
  }
  if (tl == 0 && cok) {
    part_w[(long)(c0 + c) * wlen + ntap] = phi * s_go;                       // d fc.weight
    part_w[(long)(c0 + c) * wlen + ntap + 1] = dpre * mean_c;                 // d global_fc.weight
    part_b[0 * (long)C + c0 + c] = s_dpsi;
    part_b[1 * (long)C + c0 + c] = s_ds;
    part_b[2 * (long)C + c0 + c] = s_ds;
    part_b[3 * (long)C + c0 + c] = phi * s_g;
    part_b[4 * (long)C + c0 + c] = dpre;
  }
}

// o: branch input [B][T][.] (row stride ldo); g_conv / g_inst / g_id: gradients of (convw+convkw)*psi, of fc*phi and of
// the identity term (row stride ldg; SGPBlock passes the same tensor three times, SGPMixer three slabs of d cat);
// d_o out (row stride ld_do).  part_w fp32 [B][C][wlen], part_b fp32 [B][5][C]
template <typename T>
__global__ __launch_bounds__(256) void sgp_branch_bwd_kernel(const T* __restrict__ o, long ldo, const T* __restrict__ g_conv,
                                                             const T* __restrict__ g_inst, const T* __restrict__ g_id,
                                                             long ldg, int T_len, int C, int ks, int up,
                                                             const float* __restrict__ dw, const float* __restrict__ db,
                                                             T* __restrict__ d_o, long ld_do,
                                                             float* __restrict__ part_w, float* __restrict__ part_b) {
  extern __shared__ float sm[];
  const int halo = up >> 1;
  const int wlen = 2 * ks + up + 2;
  const int trows = T_len + 2 * halo;
  float* ot = sm;
  float* dpsi = ot + trows * SGP_CH;
  float* ds = dpsi + trows * SGP_CH;
  float* gc = ds + trows * SGP_CH;
  const bool same = (g_conv == g_inst) && (g_inst == g_id);     // SGPBlock: one gradient tile serves all three roles
  float* gi = same ? gc : gc + T_len * SGP_CH;
  float* gd = same ? gc : gi + T_len * SGP_CH;
  float* res = gc + 3 * T_len * SGP_CH;
  float* wl = res + T_len * SGP_CH;
  float* red = wl + wlen * SGP_CH;                              // [17][16]
  float* red5 = red + 17 * SGP_CH;                              // [16][5][16]
  const int b = blockIdx.x, c0 = blockIdx.y * SGP_CH;
  const int c = threadIdx.x & 15;
  const bool cok = c0 + c < C;
  {
    float ov[SGP_TI][Chunk<T>::N], gv[SGP_TI][Chunk<T>::N], wv[SGP_WI];
    tile_issue<T>(o + (long)b * T_len * ldo, ldo, T_len, c0, C, ov);
    tile_issue<T>(g_conv + (long)b * T_len * ldg, ldg, T_len, c0, C, gv);
    dw_issue(dw, wlen, c0, C, wv);
    tile_commit<T>(ov, T_len, c0, C, ot, halo);
    tile_commit<T>(gv, T_len, c0, C, gc, 0);
    dw_commit(wv, wlen, c0, C, wl);
    if (!same) {
      float g2[SGP_TI][Chunk<T>::N], g3[SGP_TI][Chunk<T>::N];
      tile_issue<T>(g_inst + (long)b * T_len * ldg, ldg, T_len, c0, C, g2);
      tile_issue<T>(g_id + (long)b * T_len * ldg, ldg, T_len, c0, C, g3);
      tile_commit<T>(g2, T_len, c0, C, gi, 0);
      tile_commit<T>(g3, T_len, c0, C, gd, 0);
    }
  }
  const Bias5 bb = bias_issue(db, C, c0 + c, C);
  for (int i = threadIdx.x; i < 2 * halo * SGP_CH; i += 256) {
    const int r = i / SGP_CH, cc = i - r * SGP_CH;
    const int row = r < halo ? r : (T_len + r);
    dpsi[row * SGP_CH + cc] = 0.f;
    ds[row * SGP_CH + cc] = 0.f;
  }
  __syncthreads();
  tile_mean(ot, T_len, halo, red);
  const float mean_c = red[16 * SGP_CH + c];
  __syncthreads();
  branch_bwd_core<T>(ot, gc, gi, gd, dpsi, ds, wl, bb, cok, T_len, halo, ks, up, red, red5, mean_c,
                     d_o + (long)b * T_len * ld_do, ld_do, c0, C, part_w + (long)b * C * wlen, part_b + (long)b * 5 * C,
                     res);
}

// part_w: fp32 [B][C][wlen], part_b: fp32 [B][5][C]; d_dw [C][wlen], d_db [5][C] (the forward's packed layouts)
extern "C" int tdeed_sgp_branch_bwd(const void* o, long ldo, const void* g_conv, const void* g_inst, const void* g_id,
                                    long ldg, int B, int T, int C, int ks, int up, const float* dw, const float* db,
                                    void* d_o, long ld_do, float* part_w, float* part_b, float* d_dw, float* d_db,
                                    int dtype, void* stream) {
  TD_CHECK(o && g_conv && g_inst && g_id && dw && db && d_o && part_w && part_b && d_dw && d_db,
           "sgp_branch_bwd: null pointer");
  TD_CHECK(B > 0 && T > 0 && C % 8 == 0 && ks % 2 == 1 && up % 2 == 1 && up >= ks && ldo % 8 == 0 && ldg % 8 == 0 &&
               ld_do % 8 == 0, "sgp_branch_bwd: bad sizes");
  TD_CHECK(T <= (dtype == TDEED_BF16 ? 512 : 256) && 2 * ks + up + 2 <= 80, "sgp_branch_bwd: T=%d / taps beyond the staging registers", T);
  const int halo = up / 2, wlen = 2 * ks + up + 2;
  const size_t smem = (size_t)(3 * (T + 2 * halo) * SGP_CH + 4 * T * SGP_CH + wlen * SGP_CH + 17 * SGP_CH + 80 * SGP_CH) *
                      sizeof(float);
  TD_CHECK(smem <= 144 * 1024, "sgp_branch_bwd: T=%d too long for the LDS window", T);
  static TdDevOnce attr_set;
  if (!attr_set.get()) {
    hipError_t e = hipFuncSetAttribute((const void*)sgp_branch_bwd_kernel<float>, hipFuncAttributeMaxDynamicSharedMemorySize, 144 * 1024);
    if (e == hipSuccess) e = hipFuncSetAttribute((const void*)sgp_branch_bwd_kernel<bf16_t>, hipFuncAttributeMaxDynamicSharedMemorySize, 144 * 1024);
    if (e != hipSuccess) { tdeed_set_error("sgp_branch_bwd: hipFuncSetAttribute: %s", hipGetErrorString(e)); return TDEED_ERR_RUNTIME; }
    attr_set.set();
  }
  dim3 grid(B, cdiv(C, SGP_CH));
  hipStream_t st = (hipStream_t)stream;
  if (dtype == TDEED_F32)
    hipLaunchKernelGGL(sgp_branch_bwd_kernel<float>, grid, dim3(256), smem, st, (const float*)o, ldo, (const float*)g_conv,
                       (const float*)g_inst, (const float*)g_id, ldg, T, C, ks, up, dw, db, (float*)d_o, ld_do, part_w,
                       part_b);
  else if (dtype == TDEED_BF16)
    hipLaunchKernelGGL(sgp_branch_bwd_kernel<bf16_t>, grid, dim3(256), smem, st, (const bf16_t*)o, ldo,
                       (const bf16_t*)g_conv, (const bf16_t*)g_inst, (const bf16_t*)g_id, ldg, T, C, ks, up, dw, db,
                       (bf16_t*)d_o, ld_do, part_w, part_b);
  else { tdeed_set_error("sgp_branch_bwd: bad dtype %d", dtype); return TDEED_ERR_ARG; }
  TD_LAUNCH_CHECK("sgp_branch_bwd");
  int rc = tdeed_reduce_partials(part_w, B, (long)C * wlen, d_dw, 0, stream);
  if (rc == TDEED_OK) rc = tdeed_reduce_partials(part_b, B, 5L * C, d_db, 0, stream);
  return rc;
}

// =========================================================================== linear up-sampling backward
// forward (mixer_branch): xu[t] = l0 * xn[i0] + l1 * xn[i1], src = t * (T_lo-1)/(T_hi-1), i0 = floor(src),
// i1 = min(i0+1, T_lo-1).  d xn[j] gathers over every t (T_hi is a few hundred): no atomics, fixed order.
template <typename T>
__global__ __launch_bounds__(256) void upsample_bwd_kernel(const T* __restrict__ d_xu, long ld, int T_hi, int T_lo, int C,
                                                           T* __restrict__ d_xn, long total) {
  constexpr int EPC = Chunk<T>::N;
  const long idx = (long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= total) return;
  const int cpr = C / EPC;
  const int ck = (int)(idx % cpr);
  const long r = idx / cpr;
  const int j = (int)(r % T_lo);
  const long b = r / T_lo;
  const float scale = (T_hi > 1) ? (float)(T_lo - 1) / (float)(T_hi - 1) : 0.f;
  float a[EPC];
#pragma unroll
  for (int e = 0; e < EPC; ++e) a[e] = 0.f;
  for (int t = 0; t < T_hi; ++t) {
    float wgt = 0.f;
    if (T_hi == T_lo) {
      wgt = t == j ? 1.f : 0.f;
    } else {
      const float src = scale * (float)t;
      const int i0 = (int)src;
      const int i1 = i0 + (i0 < T_lo - 1 ? 1 : 0);
      const float l1 = fminf(fmaxf(src - (float)i0, 0.f), 1.f);
      if (i0 == j) wgt += 1.f - l1;
      if (i1 == j) wgt += l1;
    }
    if (wgt != 0.f) {
      float v[EPC];
      Chunk<T>::load(d_xu + ((long)b * T_hi + t) * ld + ck * EPC, v);
#pragma unroll
      for (int e = 0; e < EPC; ++e) a[e] = fmaf(wgt, v[e], a[e]);
    }
  }
  Chunk<T>::store(d_xn + ((long)b * T_lo + j) * C + ck * EPC, a);
}

extern "C" int tdeed_upsample_bwd(const void* d_xu, long ld, int B, int T_hi, int T_lo, int C, void* d_xn, int dtype,
                                  void* stream) {
  TD_CHECK(d_xu && d_xn && B > 0 && T_hi >= T_lo && T_lo > 0 && C % 8 == 0 && ld % 8 == 0, "upsample_bwd: bad arguments");
  hipStream_t st = (hipStream_t)stream;
  if (dtype == TDEED_F32) {
    const long total = (long)B * T_lo * (C / 4);
    hipLaunchKernelGGL(upsample_bwd_kernel<float>, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st,
                       (const float*)d_xu, ld, T_hi, T_lo, C, (float*)d_xn, total);
  } else if (dtype == TDEED_BF16) {
    const long total = (long)B * T_lo * (C / 8);
    hipLaunchKernelGGL(upsample_bwd_kernel<bf16_t>, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st,
                       (const bf16_t*)d_xu, ld, T_hi, T_lo, C, (bf16_t*)d_xn, total);
  } else { tdeed_set_error("upsample_bwd: bad dtype %d", dtype); return TDEED_ERR_ARG; }
  TD_LAUNCH_CHECK("upsample_bwd");
  return TDEED_OK;
}

// =========================================================================== adaptive max-pool backward
// d x[t] = sum over the windows i that contain t and whose (first) maximum sits at t of d y[i]; windows as in the
// forward: [floor(i*L/O), ceil((i+1)*L/O)).  Gather form, no atomics.
template <typename T>
__global__ __launch_bounds__(256) void maxpool_bwd_kernel(const T* __restrict__ x, const T* __restrict__ dy, int T_in,
                                                          int T_out, int C, T* __restrict__ dx, long total) {
  constexpr int EPC = Chunk<T>::N;
  const long idx = (long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= total) return;
  const int cpr = C / EPC;
  const int ck = (int)(idx % cpr);
  const long r = idx / cpr;
  const int t = (int)(r % T_in);
  const long b = r / T_in;
  float a[EPC], xt[EPC];
#pragma unroll
  for (int e = 0; e < EPC; ++e) a[e] = 0.f;
  Chunk<T>::load(x + ((long)b * T_in + t) * C + ck * EPC, xt);
  // candidate windows: i with lo_i <= t < hi_i;  i ranges around t*O/L
  const int i_lo = max(0, (int)(((long)t * T_out) / T_in) - 1), i_hi = min(T_out - 1, i_lo + 3);
  for (int i = i_lo; i <= i_hi; ++i) {
    const int lo = (int)(((long)i * T_in) / T_out);
    const int hi = (int)((((long)(i + 1)) * T_in + T_out - 1) / T_out);
    if (t < lo || t >= hi) continue;
    bool first_max[EPC];
#pragma unroll
    for (int e = 0; e < EPC; ++e) first_max[e] = true;
    for (int u = lo; u < hi; ++u) {
      if (u == t) continue;
      float v[EPC];
      Chunk<T>::load(x + ((long)b * T_in + u) * C + ck * EPC, v);
#pragma unroll
      for (int e = 0; e < EPC; ++e)                            // an earlier equal value wins the tie, a later one does not
        if (v[e] > xt[e] || (u < t && v[e] == xt[e])) first_max[e] = false;
    }
    float g[EPC];
    Chunk<T>::load(dy + ((long)b * T_out + i) * C + ck * EPC, g);
#pragma unroll
    for (int e = 0; e < EPC; ++e)
      if (first_max[e]) a[e] += g[e];
  }
  Chunk<T>::store(dx + ((long)b * T_in + t) * C + ck * EPC, a);
}

extern "C" int tdeed_maxpool_bwd(const void* x, const void* dy, int B, int T_in, int T_out, int C, void* dx, int dtype,
                                 void* stream) {
  TD_CHECK(x && dy && dx && B > 0 && T_in >= T_out && T_out > 0 && C % 8 == 0, "maxpool_bwd: bad arguments");
  hipStream_t st = (hipStream_t)stream;
  if (dtype == TDEED_F32) {
    const long total = (long)B * T_in * (C / 4);
    hipLaunchKernelGGL(maxpool_bwd_kernel<float>, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, (const float*)x,
                       (const float*)dy, T_in, T_out, C, (float*)dx, total);
  } else if (dtype == TDEED_BF16) {
    const long total = (long)B * T_in * (C / 8);
    hipLaunchKernelGGL(maxpool_bwd_kernel<bf16_t>, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st,
                       (const bf16_t*)x, (const bf16_t*)dy, T_in, T_out, C, (bf16_t*)dx, total);
  } else { tdeed_set_error("maxpool_bwd: bad dtype %d", dtype); return TDEED_ERR_ARG; }
  TD_LAUNCH_CHECK("maxpool_bwd");
  return TDEED_OK;
}
