// Round-4 training backward of the RegNetY bottleneck (timm Bottleneck + SEModule + BatchNorm2d in batch-statistics mode as
// autograd differentiates it for /root/reference/model/model.py:265-324, modules.py:390-404): the passes between conv3's
// input gradient and conv2's input gradient without the intermediate maps.
//
// Behind conv3's input gradient d = d(y2 * gate) the reference's graph holds, per bottleneck,
//     d_gate[f][c] = sum_px d * y2                       (SE gate gradient)          -- one pass over (d, y2)
//     d_y2         = d * gate[f][c] + d_p[f][c] / hw     (SE scale + squeeze)        -- one pass, writes d_y2
//     g            = d_y2 * (y2 > 0)                     (ReLU)
//     sum g, sum g * xhat                                (BatchNorm statistics)      -- one pass over (d_y2, z2)
//     dz2          = k1 * g + k2 * z2 + k3               (BatchNorm input gradient)  -- one pass, writes dz2
// with y2 = relu(a * z2 + b).  gate and d_p are constant over a frame's pixels, so everything the two reductions need is
// linear in FIVE per-(frame, channel) sums of one pass over (d, z2):
//     S0 = sum d * y2,  S1 = sum d * m,  S2 = sum d * m * (z2 - mean),  S3 = sum m,  S4 = sum m * (z2 - mean),   m = [a z2 + b > 0]
//     d_gate = S0;   sum g = sum_f gate * S1 + (d_p / hw) * S3;   sum g * xhat = rstd * sum_f gate * S2 + (d_p / hw) * S4
// and the apply pass forms g from (d, z2, gate, d_p) on the fly: 5 map passes instead of 9, d_y2 never exists.
#include "common.h"

// --------------------------------------------------------------------------- pass A: per-(frame, channel) sums
// sums [5][N][C] fp32.  One workgroup per frame: lanes = (pixel slice, 16-byte channel chunk), 4 row loads of both maps in
// flight per lane; slices folded through LDS in a fixed order.
template <typename T>
__global__ __launch_bounds__(256) void se_bn_sums_kernel(const T* __restrict__ d, const T* __restrict__ z, int hw, int C,
                                                         const float* __restrict__ fa, const float* __restrict__ fb,
                                                         const float* __restrict__ mean, long NC, float* __restrict__ sums) {
  constexpr int EPC = Chunk<T>::N;
  extern __shared__ float red[];       // [S][5][C]
  const long f = blockIdx.x;
  const int nch = C / EPC;
  const int S = 256 / nch > 0 ? 256 / nch : 1;
  for (int ch = threadIdx.x % nch, s = threadIdx.x / nch; s < S && ch < nch; ch += 256) {
    const int c0 = ch * EPC;
    float a[EPC], b[EPC], mu[EPC], acc[5][EPC];
#pragma unroll
    for (int e = 0; e < EPC; ++e) {
      a[e] = fa[c0 + e];
      b[e] = fb[c0 + e];
      mu[e] = mean[c0 + e];
#pragma unroll
      for (int k = 0; k < 5; ++k) acc[k][e] = 0.f;
    }
    const long base = f * hw * C + c0;
    for (int p0 = s; p0 < hw; p0 += S * 4) {
      float dv[4][EPC], zv[4][EPC];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const long off = base + (long)min(p0 + u * S, hw - 1) * C;
        Chunk<T>::load(d + off, dv[u]);
        Chunk<T>::load(z + off, zv[u]);
      }
#pragma unroll
      for (int u = 0; u < 4; ++u)
        if (p0 + u * S < hw) {
#pragma unroll
          for (int e = 0; e < EPC; ++e) {
            const float act = fmaf(zv[u][e], a[e], b[e]);
            const bool on = act > 0.f;
            const float y = on ? round_to<T>(act) : 0.f;          // y2 as the forward's consumers saw it (rounded to T)
            const float dm = on ? dv[u][e] : 0.f;
            const float zc = on ? zv[u][e] - mu[e] : 0.f;
            acc[0][e] = fmaf(dv[u][e], y, acc[0][e]);
            acc[1][e] += dm;
            acc[2][e] = fmaf(dm, zc, acc[2][e]);
            acc[3][e] += on ? 1.f : 0.f;
            acc[4][e] += zc;
          }
        }
    }
#pragma unroll
    for (int k = 0; k < 5; ++k)
#pragma unroll
      for (int e = 0; e < EPC; ++e) red[(s * 5 + k) * C + c0 + e] = acc[k][e];
  }
  __syncthreads();
  for (int j = threadIdx.x; j < 5 * C; j += 256) {
    const int k = j / C, c = j - k * C;
    float v = 0.f;
    for (int s = 0; s < S; ++s) v += red[(s * 5 + k) * C + c];
    sums[(long)k * NC + f * C + c] = v;
  }
}

extern "C" int tdeed_se_bn_bwd_sums(const void* d, const void* z, int N, int hw, int C, const float* fa, const float* fb,
                                    const float* mean, float* sums, int dtype, void* stream) {
  TD_CHECK(d && z && fa && fb && mean && sums, "se_bn_bwd_sums: null pointer");
  TD_CHECK(N > 0 && hw > 0 && C > 0 && C % 8 == 0 && C <= 2048, "se_bn_bwd_sums: bad sizes N=%d hw=%d C=%d", N, hw, C);
  TD_CHECK(dtype == TDEED_F32 || dtype == TDEED_BF16, "se_bn_bwd_sums: bad dtype %d", dtype);
  hipStream_t st = (hipStream_t)stream;
  const int nch = C / (dtype == TDEED_F32 ? 4 : 8), S = 256 / nch > 0 ? 256 / nch : 1;
  const size_t smem = (size_t)S * 5 * C * sizeof(float);
  TD_CHECK(smem <= 64 * 1024, "se_bn_bwd_sums: C=%d beyond the LDS budget", C);
  if (dtype == TDEED_F32)
    hipLaunchKernelGGL(se_bn_sums_kernel<float>, dim3(N), dim3(256), smem, st, (const float*)d, (const float*)z, hw, C, fa, fb,
                       mean, (long)N * C, sums);
  else
    hipLaunchKernelGGL(se_bn_sums_kernel<bf16_t>, dim3(N), dim3(256), smem, st, (const bf16_t*)d, (const bf16_t*)z, hw, C, fa,
                       fb, mean, (long)N * C, sums);
  TD_LAUNCH_CHECK("se_bn_bwd_sums");
  return TDEED_OK;
}

// --------------------------------------------------------------------------- BatchNorm statistics from the frame sums
// out[0][c] = sum g = db,  out[1][c] = sum g * xhat = dw  (the layout tdeed_bn_train_bwd's `sums` has).
// One workgroup per 8 channels, lanes = (32 frame lanes, 8 channels), folded in double in a fixed order.
__global__ __launch_bounds__(256) void se_bn_finalize_kernel(const float* __restrict__ sums, const float* __restrict__ gate,
                                                             const float* __restrict__ d_p, int N, int C, float inv_hw,
                                                             const float* __restrict__ rstd, float* __restrict__ out) {
  __shared__ double r1[32][9], r2[32][9];
  const int cl = threadIdx.x & 7, fl = threadIdx.x >> 3;
  const int c = blockIdx.x * 8 + cl;
  const long NC = (long)N * C;
  double s1 = 0.0, s2 = 0.0;
  for (int f0 = fl; f0 < N; f0 += 4 * 32) {
    float g[4], q[4], v1[4], v2[4], v3[4], v4[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const long i = (long)min(f0 + u * 32, N - 1) * C + c;
      g[u] = gate[i];
      q[u] = d_p[i] * inv_hw;
      v1[u] = sums[NC + i];
      v2[u] = sums[2 * NC + i];
      v3[u] = sums[3 * NC + i];
      v4[u] = sums[4 * NC + i];
    }
#pragma unroll
    for (int u = 0; u < 4; ++u)
      if (f0 + u * 32 < N) {
        s1 += (double)g[u] * (double)v1[u] + (double)q[u] * (double)v3[u];
        s2 += (double)g[u] * (double)v2[u] + (double)q[u] * (double)v4[u];
      }
  }
  r1[fl][cl] = s1;
  r2[fl][cl] = s2;
  __syncthreads();
  if (fl != 0) return;
  s1 = 0.0;
  s2 = 0.0;
  for (int i = 0; i < 32; ++i) {
    s1 += r1[i][cl];
    s2 += r2[i][cl];
  }
  out[c] = (float)s1;
  out[C + c] = (float)(s2 * (double)rstd[c]);
}

extern "C" int tdeed_se_bn_bwd_finalize(const float* sums, const float* gate, const float* d_p, int N, int hw, int C,
                                        const float* rstd, float* out, void* stream) {
  TD_CHECK(sums && gate && d_p && rstd && out, "se_bn_bwd_finalize: null pointer");
  TD_CHECK(N > 0 && hw > 0 && C > 0 && C % 8 == 0, "se_bn_bwd_finalize: bad sizes");
  hipLaunchKernelGGL(se_bn_finalize_kernel, dim3(C / 8), dim3(256), 0, (hipStream_t)stream, sums, gate, d_p, N, C,
                     1.0f / (float)hw, rstd, out);
  TD_LAUNCH_CHECK("se_bn_bwd_finalize");
  return TDEED_OK;
}

// --------------------------------------------------------------------------- pass B: dz2 from (d, z2, gate, d_p)
// g = m * (d * gate[f][c] + d_p[f][c] / hw),  dz = k1 * g + k2 * z + k3  (k1..k3 as in tdeed_bn_train_bwd).
// thread = (row lane, channel chunk) of one frame's slice of rows: the per-(frame, channel) and per-channel constants are
// loaded once per thread.
constexpr int SB_U = 4;
template <typename T>
__global__ __launch_bounds__(256) void se_bn_apply_kernel(const T* __restrict__ d, const T* __restrict__ z,
                                                          const float* __restrict__ gate, const float* __restrict__ d_p,
                                                          float inv_hw, int hw, const float* __restrict__ fa,
                                                          const float* __restrict__ fb, const float* __restrict__ mean,
                                                          const float* __restrict__ rstd, const float* __restrict__ w,
                                                          const float* __restrict__ sums, float inv_M, T* __restrict__ dz,
                                                          int nch, int rpw) {
  constexpr int EPC = Chunk<T>::N;
  const int RL = 256 / nch, rl = threadIdx.x / nch, ck = threadIdx.x - rl * nch;
  if (rl >= RL) return;
  const int C = nch * EPC, c0 = ck * EPC;
  const long n = blockIdx.y;
  float k1[EPC], k2[EPC], k3[EPC], a[EPC], b[EPC], gv[EPC], qv[EPC];
#pragma unroll
  for (int e = 0; e < EPC; ++e) {
    const int c = c0 + e;
    const float rs = rstd[c], mu = mean[c], s1 = sums[c] * inv_M, s2 = sums[C + c] * inv_M;
    k1[e] = w[c] * rs;
    k2[e] = -k1[e] * rs * s2;
    k3[e] = k1[e] * (mu * rs * s2 - s1);
    a[e] = fa[c];
    b[e] = fb[c];
    gv[e] = gate[n * C + c];
    qv[e] = d_p[n * C + c] * inv_hw;
  }
  const int m0 = blockIdx.x * rpw, m1 = min(hw, m0 + rpw);
  const long fbase = n * hw * C + c0;
  for (int r0 = m0 + rl; r0 < m1; r0 += RL * SB_U) {
    float dv[SB_U][EPC], zv[SB_U][EPC];
#pragma unroll
    for (int u = 0; u < SB_U; ++u) {
      const long off = fbase + (long)min(r0 + u * RL, m1 - 1) * C;
      Chunk<T>::load(d + off, dv[u]);
      Chunk<T>::load(z + off, zv[u]);
    }
#pragma unroll
    for (int u = 0; u < SB_U; ++u) {
      const int r = r0 + u * RL;
      if (r < m1) {
#pragma unroll
        for (int e = 0; e < EPC; ++e) {
          const bool on = fmaf(zv[u][e], a[e], b[e]) > 0.f;
          const float g = on ? fmaf(dv[u][e], gv[e], qv[e]) : 0.f;
          zv[u][e] = fmaf(k1[e], g, fmaf(k2[e], zv[u][e], k3[e]));
        }
        Chunk<T>::store(dz + fbase + (long)r * C, zv[u]);
      }
    }
  }
}

extern "C" int tdeed_se_bn_bwd_apply(const void* d, const void* z, const float* gate, const float* d_p, int N, int hw, int C,
                                     const float* fa, const float* fb, const float* mean, const float* rstd, const float* w,
                                     const float* sums, void* dz, int dtype, void* stream) {
  TD_CHECK(d && z && gate && d_p && fa && fb && mean && rstd && w && sums && dz, "se_bn_bwd_apply: null pointer");
  TD_CHECK(N > 0 && N <= 65535 && hw > 0 && C > 0 && C % 8 == 0, "se_bn_bwd_apply: bad sizes N=%d hw=%d C=%d", N, hw, C);
  TD_CHECK(dtype == TDEED_F32 || dtype == TDEED_BF16, "se_bn_bwd_apply: bad dtype %d", dtype);
  hipStream_t st = (hipStream_t)stream;
  const int nch = C / (dtype == TDEED_F32 ? 4 : 8);
  TD_CHECK(nch <= 256, "se_bn_bwd_apply: C=%d too wide", C);
  const int rpw = (256 / nch) * SB_U * 4;
  const dim3 grid((unsigned)cdiv(hw, rpw), (unsigned)N);
  const float inv_hw = 1.0f / (float)hw, inv_M = 1.0f / ((float)N * (float)hw);
  if (dtype == TDEED_F32)
    hipLaunchKernelGGL(se_bn_apply_kernel<float>, grid, dim3(256), 0, st, (const float*)d, (const float*)z, gate, d_p, inv_hw,
                       hw, fa, fb, mean, rstd, w, sums, inv_M, (float*)dz, nch, rpw);
  else
    hipLaunchKernelGGL(se_bn_apply_kernel<bf16_t>, grid, dim3(256), 0, st, (const bf16_t*)d, (const bf16_t*)z, gate, d_p,
                       inv_hw, hw, fa, fb, mean, rstd, w, sums, inv_M, (bf16_t*)dz, nch, rpw);
  TD_LAUNCH_CHECK("se_bn_bwd_apply");
  return TDEED_OK;
}
