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

// =========================================================================== "gradient sink": ReLU mask + BatchNorm statistics
// at the PRODUCERS of a block-input gradient
// The gradient dx of a bottleneck's input is the gradient `dout` at the previous block's output ReLU.  The previous block's
// backward starts with g = dout * [out > 0] and the column sums (sum g, sum g * xhat) of its conv3 (and shortcut) BatchNorm.
// Both are linear in dout's contributions, so every kernel that writes or adds into dx applies the mask and leaves the sums of
// what it contributed (tdeed_gemm_dgrad's epilogue; here: the gate-shift module's input gradient joining columns [0, Fp)).
// The previous block then runs only the apply pass of its BatchNorm backward (tdeed_bn_bwd_from_parts): the masked-gradient
// map `d_res`, the separate statistics pass and the ReLU pass disappear.

// dx[m][c] (c < Fp, row stride C) = round(dx + (a[m][c] + b[m][c]) * [mask[m][c] > 0]);  part[wg][k][c] (k < 3, c < Fp) =
// sums over the workgroup's rows of delta = new - old, delta * (bz - bmean), delta * (bzd - bmean_d)  (mask / bz / bzd NULL: skipped)
template <typename T>
__global__ __launch_bounds__(256) void gsf_add_cols_sink_kernel(const T* __restrict__ a, const T* __restrict__ b, long M, int C,
                                                                int Fp, T* __restrict__ dx, const T* __restrict__ mask,
                                                                long ldmask, const T* __restrict__ bz, long ldbz,
                                                                const float* __restrict__ bmean, const T* __restrict__ bzd,
                                                                long ldbzd, const float* __restrict__ bmean_d,
                                                                float* __restrict__ part, int nch, long rpw,
                                                                const T* __restrict__ bnx, const float* __restrict__ bn_sums,
                                                                const float* __restrict__ bn_mean,
                                                                const float* __restrict__ bn_rstd, const float* __restrict__ bn_w,
                                                                float inv_M) {
  // bnx: b is the gradient at the OUTPUT of the module's BatchNorm3d and the BatchNorm backward is applied right here,
  // b' = round(k1 b + k2 bnx + k3) from the statistics bn_sums = (sum g, sum g xhat): the dz map is not written and read back
  constexpr int EPC = Chunk<T>::N;
  extern __shared__ float red[];                                 // [RL][3][Fp]
  const int RL = 256 / nch, rl = threadIdx.x / nch, ck = threadIdx.x - rl * nch;
  const int c0 = ck * EPC;
  float s1[EPC], s2[EPC], s3[EPC], bm[EPC], bmd[EPC];
  float k1[EPC], k2[EPC], k3[EPC];
#pragma unroll
  for (int e = 0; e < EPC; ++e) {
    s1[e] = s2[e] = s3[e] = 0.f;
    bm[e] = (part && rl < RL) ? bmean[c0 + e] : 0.f;
    bmd[e] = (part && bzd && rl < RL) ? bmean_d[c0 + e] : 0.f;
    k1[e] = k2[e] = k3[e] = 0.f;
    if (bnx && rl < RL) {
      const int c = c0 + e;
      const float rs = bn_rstd[c], mu = bn_mean[c], m1 = bn_sums[c] * inv_M, m2 = bn_sums[Fp + c] * inv_M;
      k1[e] = bn_w[c] * rs;
      k2[e] = -k1[e] * rs * m2;
      k3[e] = k1[e] * (mu * rs * m2 - m1);
    }
  }
  const long m0 = (long)blockIdx.x * rpw, m1 = min(M, m0 + rpw);
  if (rl < RL) {
    for (long r0 = m0 + rl; r0 < m1; r0 += (long)RL * 2) {
      float av[2][EPC], bv[2][EPC], xv[2][EPC], mv[2][EPC], zv[2][EPC], zd[2][EPC];
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const long r = min(r0 + (long)u * RL, m1 - 1);
        Chunk<T>::load(a + r * Fp + c0, av[u]);
        Chunk<T>::load(b + r * Fp + c0, bv[u]);
        if (bnx) {
          float bx[EPC];
          Chunk<T>::load(bnx + r * Fp + c0, bx);
#pragma unroll
          for (int e = 0; e < EPC; ++e) bv[u][e] = round_to<T>(fmaf(k1[e], bv[u][e], fmaf(k2[e], bx[e], k3[e])));
        }
        Chunk<T>::load(dx + r * C + c0, xv[u]);
        if (mask) Chunk<T>::load(mask + r * ldmask + c0, mv[u]);
        if (part) Chunk<T>::load(bz + r * ldbz + c0, zv[u]);
        if (part && bzd) Chunk<T>::load(bzd + r * ldbzd + c0, zd[u]);
      }
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const long r = r0 + (long)u * RL;
        if (r < m1) {
          float o[EPC];
#pragma unroll
          for (int e = 0; e < EPC; ++e) {
            const float add = (mask && !(mv[u][e] > 0.f)) ? 0.f : av[u][e] + bv[u][e];
            o[e] = round_to<T>(xv[u][e] + add);
            const float dl = o[e] - xv[u][e];
            if (part) {
              s1[e] += dl;
              s2[e] = fmaf(dl, zv[u][e] - bm[e], s2[e]);
              if (bzd) s3[e] = fmaf(dl, zd[u][e] - bmd[e], s3[e]);
            }
          }
          Chunk<T>::store(dx + r * C + c0, o);
        }
      }
    }
  }
  if (!part) return;
  if (rl < RL) {
#pragma unroll
    for (int e = 0; e < EPC; ++e) {
      red[(rl * 3 + 0) * Fp + c0 + e] = s1[e];
      red[(rl * 3 + 1) * Fp + c0 + e] = s2[e];
      red[(rl * 3 + 2) * Fp + c0 + e] = s3[e];
    }
  }
  __syncthreads();
  for (int j = threadIdx.x; j < 3 * Fp; j += 256) {
    float v = 0.f;
    for (int i = 0; i < RL; ++i) v += red[i * 3 * Fp + j];
    part[(long)blockIdx.x * 3 * Fp + j] = v;
  }
}

static long add_cols_rpw(long M, int nch) {
  const long RL = 256 / nch;
  long rpw = RL * 2 * 4;
  if ((M + rpw - 1) / rpw > 2048) rpw = ((M + 2047) / 2048 + RL * 2 - 1) / (RL * 2) * (RL * 2);
  return rpw;
}
extern "C" int tdeed_gsf_add_cols_sink_parts(long M, int Fp, int dtype) {
  const int nch = Fp / (dtype == TDEED_F32 ? 4 : 8);
  if (nch <= 0 || nch > 256) return 0;
  const long rpw = add_cols_rpw(M, nch);
  return (int)((M + rpw - 1) / rpw);
}

static int add_cols_sink_launch(const void* a, const void* b, long M, int C, int Fp, void* dx, const void* mask,
                                long ldmask, const void* bz, long ldbz, const float* bmean, const void* bzd, long ldbzd,
                                const float* bmean_d, float* part, const void* bnx, const float* bn_sums, const float* bn_mean,
                                const float* bn_rstd, const float* bn_w, int dtype, void* stream) {
  TD_CHECK(a && b && dx && M > 0 && Fp > 0 && Fp <= C && Fp % 8 == 0 && C % 8 == 0, "gsf_add_cols_sink: bad arguments");
  TD_CHECK(dtype == TDEED_F32 || dtype == TDEED_BF16, "gsf_add_cols_sink: bad dtype %d", dtype);
  TD_CHECK(!part || (bz && bmean && (!bzd || bmean_d)), "gsf_add_cols_sink: statistics operands missing");
  TD_CHECK((!mask || ldmask % 8 == 0) && (!bz || ldbz % 8 == 0) && (!bzd || ldbzd % 8 == 0), "gsf_add_cols_sink: bad row strides");
  hipStream_t st = (hipStream_t)stream;
  const int nch = Fp / (dtype == TDEED_F32 ? 4 : 8);
  TD_CHECK(nch <= 256, "gsf_add_cols_sink: Fp=%d too wide", Fp);
  const long rpw = add_cols_rpw(M, nch);
  const long nwg = (M + rpw - 1) / rpw;
  const size_t smem = (size_t)(256 / nch) * 3 * Fp * sizeof(float);
  TD_CHECK(smem <= 64 * 1024, "gsf_add_cols_sink: Fp=%d beyond the LDS budget", Fp);
  if (dtype == TDEED_F32)
    hipLaunchKernelGGL(gsf_add_cols_sink_kernel<float>, dim3((unsigned)nwg), dim3(256), smem, st, (const float*)a, (const float*)b,
                       M, C, Fp, (float*)dx, (const float*)mask, ldmask, (const float*)bz, ldbz, bmean, (const float*)bzd, ldbzd,
                       bmean_d, part, nch, rpw, (const float*)bnx, bn_sums, bn_mean, bn_rstd, bn_w, 1.0f / (float)M);
  else
    hipLaunchKernelGGL(gsf_add_cols_sink_kernel<bf16_t>, dim3((unsigned)nwg), dim3(256), smem, st, (const bf16_t*)a,
                       (const bf16_t*)b, M, C, Fp, (bf16_t*)dx, (const bf16_t*)mask, ldmask, (const bf16_t*)bz, ldbz, bmean,
                       (const bf16_t*)bzd, ldbzd, bmean_d, part, nch, rpw, (const bf16_t*)bnx, bn_sums, bn_mean, bn_rstd, bn_w,
                       1.0f / (float)M);
  TD_LAUNCH_CHECK("gsf_add_cols_sink");
  return TDEED_OK;
}
extern "C" int tdeed_gsf_add_cols_sink(const void* a, const void* b, long M, int C, int Fp, void* dx, const void* mask,
                                       long ldmask, const void* bz, long ldbz, const float* bmean, const void* bzd, long ldbzd,
                                       const float* bmean_d, float* part, int dtype, void* stream) {
  return add_cols_sink_launch(a, b, M, C, Fp, dx, mask, ldmask, bz, ldbz, bmean, bzd, ldbzd, bmean_d, part, nullptr, nullptr,
                              nullptr, nullptr, nullptr, dtype, stream);
}
// ... with the backward of the module's BatchNorm3d applied to b on load: b [M][Fp] = the gradient at the BatchNorm's output
// (d_bn of tdeed_gsf_bwd), bnx [M][Fp] its input (the dense slice), bn_sums fp32 [2][Fp] = (sum g, sum g xhat) as
// tdeed_bn_bwd_from_parts leaves them, bn_mean / bn_rstd / bn_w [Fp]: its batch statistics and weight
extern "C" int tdeed_gsf_add_cols_sink_bn(const void* a, const void* b, long M, int C, int Fp, void* dx, const void* mask,
                                          long ldmask, const void* bz, long ldbz, const float* bmean, const void* bzd,
                                          long ldbzd, const float* bmean_d, float* part, const void* bnx, const float* bn_sums,
                                          const float* bn_mean, const float* bn_rstd, const float* bn_w, int dtype,
                                          void* stream) {
  TD_CHECK(bnx && bn_sums && bn_mean && bn_rstd && bn_w, "gsf_add_cols_sink_bn: null pointer");
  return add_cols_sink_launch(a, b, M, C, Fp, dx, mask, ldmask, bz, ldbz, bmean, bzd, ldbzd, bmean_d, part, bnx, bn_sums, bn_mean,
                              bn_rstd, bn_w, dtype, stream);
}

// --------------------------------------------------------------------------- BatchNorm backward from producer partials
// sums[0][c] = sum_p A[p][0][c] + sum_p B[p][0][c] (c < nB),  sums[1][c] = rstd[c] * (the same over row `q` of the partials):
// tmpA [SA][3][C], tmpB [SB][3][nB] are the partial rows already folded to <= 64 slices.  One thread per channel.
__global__ __launch_bounds__(256) void bn_parts_finalize_kernel(const float* __restrict__ tmpA, int SA, const float* __restrict__ tmpB,
                                                                int SB, int nB, int q, int C, const float* __restrict__ rstd,
                                                                float* __restrict__ sums) {
  // one workgroup per 8 channels: lanes = (32 slice lanes, 8 channels), every slice row of both partial sets requested at
  // once, folded in double in a fixed order (one thread per channel walking 64 dependent rows took 29 us per BatchNorm)
  __shared__ double r1[32][9], r2[32][9];
  const int cl = threadIdx.x & 7, sl = threadIdx.x >> 3;
  const int c = blockIdx.x * 8 + cl;
  double s1 = 0.0, s2 = 0.0;
  if (c < C) {
    float a1[2], a2[2], b1[2], b2[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int s = min(sl + 32 * u, SA - 1);
      a1[u] = tmpA[((long)s * 3 + 0) * C + c];
      a2[u] = tmpA[((long)s * 3 + q) * C + c];
      const bool hb = tmpB && c < nB;
      const int sb_ = min(sl + 32 * u, max(SB, 1) - 1);
      b1[u] = hb ? tmpB[((long)sb_ * 3 + 0) * nB + c] : 0.f;
      b2[u] = hb ? tmpB[((long)sb_ * 3 + q) * nB + c] : 0.f;
    }
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      if (sl + 32 * u < SA) { s1 += (double)a1[u]; s2 += (double)a2[u]; }
      if (sl + 32 * u < SB) { s1 += (double)b1[u]; s2 += (double)b2[u]; }
    }
  }
  r1[sl][cl] = s1;
  r2[sl][cl] = s2;
  __syncthreads();
  if (sl != 0 || c >= C) return;
  s1 = 0.0;
  s2 = 0.0;
  for (int i = 0; i < 32; ++i) {
    s1 += r1[i][cl];
    s2 += r2[i][cl];
  }
  sums[c] = (float)s1;
  sums[C + c] = (float)(s2 * (double)rstd[c]);
}

// dz = k1 * g + k2 * z + k3 with g already masked (k1..k3 as in tdeed_bn_train_bwd)
template <typename T>
__global__ __launch_bounds__(256) void bn_bwd_apply_g_kernel(const T* __restrict__ z, const T* __restrict__ g,
                                                             const float* __restrict__ mean, const float* __restrict__ rstd,
                                                             const float* __restrict__ w, const float* __restrict__ sums,
                                                             float inv_M, T* __restrict__ dz, long M, int nch, int rpw) {
  constexpr int EPC = Chunk<T>::N;
  const int RL = 256 / nch, rl = threadIdx.x / nch, ck = threadIdx.x - rl * nch;
  if (rl >= RL) return;
  const int C = nch * EPC, c0 = ck * EPC;
  float k1[EPC], k2[EPC], k3[EPC];
#pragma unroll
  for (int e = 0; e < EPC; ++e) {
    const int c = c0 + e;
    const float rs = rstd[c], mu = mean[c], s1 = sums[c] * inv_M, s2 = sums[C + c] * inv_M;
    k1[e] = w[c] * rs;
    k2[e] = -k1[e] * rs * s2;
    k3[e] = k1[e] * (mu * rs * s2 - s1);
  }
  const long m0 = (long)blockIdx.x * rpw, m1 = min(M, m0 + rpw);
  for (long r0 = m0 + rl; r0 < m1; r0 += (long)RL * SB_U) {
    float zv[SB_U][EPC], gv[SB_U][EPC];
#pragma unroll
    for (int u = 0; u < SB_U; ++u) {
      const long r = min(r0 + (long)u * RL, m1 - 1);
      Chunk<T>::load(z + r * C + c0, zv[u]);
      Chunk<T>::load(g + r * C + c0, gv[u]);
    }
#pragma unroll
    for (int u = 0; u < SB_U; ++u) {
      const long r = r0 + (long)u * RL;
      if (r < m1) {
#pragma unroll
        for (int e = 0; e < EPC; ++e) zv[u][e] = fmaf(k1[e], gv[u][e], fmaf(k2[e], zv[u][e], k3[e]));
        Chunk<T>::store(dz + r * C + c0, zv[u]);
      }
    }
  }
}

// partA: PA rows of [3][C] floats (tdeed_gemm_dgrad's bpart), partB (optional): PB rows of [3][nB] (tdeed_gsf_add_cols_sink's
// part); q = 1 / 2: which product row holds this BatchNorm's sum g * (z - mean).  tmp: fp32 [2 * 64 * 3 * C] scratch;
// sums: fp32 [2][C] (out: d bias, d weight).  dz may be NULL (statistics only).
extern "C" int tdeed_bn_bwd_from_parts(const void* z, const void* g, long M, int C, const float* mean, const float* rstd,
                                       const float* w, const float* partA, int PA, const float* partB, int PB, int nB, int q,
                                       float* tmp, float* sums, void* dz, int dtype, void* stream) {
  TD_CHECK(z && g && mean && rstd && w && partA && tmp && sums, "bn_bwd_from_parts: null pointer");
  TD_CHECK(M > 0 && C > 0 && C % 8 == 0 && PA > 0 && (q == 1 || q == 2), "bn_bwd_from_parts: bad sizes");
  TD_CHECK(!partB || (PB > 0 && nB > 0 && nB <= C), "bn_bwd_from_parts: bad second partial set");
  TD_CHECK(dtype == TDEED_F32 || dtype == TDEED_BF16, "bn_bwd_from_parts: bad dtype %d", dtype);
  hipStream_t st = (hipStream_t)stream;
  const float* tA = partA;
  int SA = PA;
  if (PA > 64) {
    int rc = tdeed_fold_rows(partA, 3L * C, PA, 3 * C, 64, tmp, 3L * C, stream);
    if (rc != TDEED_OK) return rc;
    tA = tmp;
    SA = 64;                                                     // (slices behind the last row hold zeros)
  }
  const float* tB = nullptr;
  int SB = 0;
  if (partB) {
    tB = partB;
    SB = PB;
    if (PB > 64) {
      float* t2 = tmp + 64L * 3 * C;
      int rc = tdeed_fold_rows(partB, 3L * nB, PB, 3 * nB, 64, t2, 3L * nB, stream);
      if (rc != TDEED_OK) return rc;
      tB = t2;
      SB = 64;
    }
  }
  TD_CHECK(SA <= 64 && SB <= 64, "bn_bwd_from_parts: more than 64 folded partial rows");
  hipLaunchKernelGGL(bn_parts_finalize_kernel, dim3((C + 7) / 8), dim3(256), 0, st, tA, SA, tB, SB, nB, q, C, rstd, sums);
  TD_LAUNCH_CHECK("bn_parts_finalize");
  if (!dz) return TDEED_OK;
  const int nch = C / (dtype == TDEED_F32 ? 4 : 8);
  TD_CHECK(nch <= 256, "bn_bwd_from_parts: C=%d too wide", C);
  const int rpw = (256 / nch) * SB_U * 4;
  const long nwg = (M + rpw - 1) / rpw;
  TD_CHECK(nwg < 0x7fffffffL, "bn_bwd_from_parts: too many rows");
  const float inv_M = 1.0f / (float)M;
  if (dtype == TDEED_F32)
    hipLaunchKernelGGL(bn_bwd_apply_g_kernel<float>, dim3((unsigned)nwg), dim3(256), 0, st, (const float*)z, (const float*)g, mean,
                       rstd, w, sums, inv_M, (float*)dz, M, nch, rpw);
  else
    hipLaunchKernelGGL(bn_bwd_apply_g_kernel<bf16_t>, dim3((unsigned)nwg), dim3(256), 0, st, (const bf16_t*)z, (const bf16_t*)g,
                       mean, rstd, w, sums, inv_M, (bf16_t*)dz, M, nch, rpw);
  TD_LAUNCH_CHECK("bn_bwd_apply_g");
  return TDEED_OK;
}
