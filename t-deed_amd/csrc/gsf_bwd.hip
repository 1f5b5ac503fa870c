// Backward of the gate-shift-fuse module (`_GSF.forward`, /root/reference/model/impl/gsf.py:38-93) as wrapped by
// `GatedShift` around conv1 of the s3/s4 bottlenecks (model/shift.py:64-93).  Same tensors as the forward in gsf.hip:
//   x     [N][hw][C]   block input (N = B*T frames), the module acts on channels [0,F)
//   gate  [N][hw][2]   tanh(conv3d(relu(bn(x))))            (saved by the forward)
//   fw    [B][F][T]    fusion weights, indexed by source channel (saved by the forward)
//   ysum, xsum [N][F]  spatial sums of gate*x and x           (saved by the forward)
//   dA    [N*hw][Fp]   gradient w.r.t. the module output in conv1's operand layout: column co = g*Fh + 2j + i holds
//                      source channel ci = g*Fh + i*Fq + j; columns [F,Fp) are pass-through copies of x
// First versions: plain gather kernels (a lane per pixel or per (channel, tap)), ordered partial sums for parameters.
#include "common.h"
#include <stdlib.h>

__device__ __forceinline__ int gsf_out_col(int ci, int Fh, int Fq) {
  const int g = ci >= Fh, cl = ci - g * Fh;
  const int i = cl >= Fq, j = cl - i * Fq;
  return g * Fh + 2 * j + i;
}

// dense copy of the module's channels: xs[m][c] = x[m][c] for c < F, 0 for F <= c < Fp   (operand of the BatchNorm3d)
template <typename T>
__global__ __launch_bounds__(256) void gsf_slice_kernel(const T* __restrict__ x, long M, int C, int F, int Fp,
                                                        T* __restrict__ xs) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= M * Fp) return;
  const long m = i / Fp;
  const int c = (int)(i - m * Fp);
  xs[i] = c < F ? x[m * C + c] : (T)0.f;
}

// ---- d fusion weight: d_wgt[f][ci] = sum_p dA[p][co(ci)] * (ys - r)
template <typename T>
__global__ __launch_bounds__(256) void gsf_bwd_dw_kernel(const T* __restrict__ x, const float* __restrict__ gate,
                                                         const T* __restrict__ dA, int T_len, int hw, int C, int F, int Fp,
                                                         float* __restrict__ d_wgt) {
  extern __shared__ float red[];       // [S][F]
  const long f = blockIdx.x;
  const int t = (int)(f % T_len);
  const int Fh = F >> 1, Fq = F >> 2;
  const int S = 256 / F > 0 ? 256 / F : 1;
  for (int ci = threadIdx.x % F, s = threadIdx.x / F; s < S && ci < F; ci += 256) {
    const int g = ci >= Fh;
    const int co = gsf_out_col(ci, Fh, Fq);
    const int ts = g ? t - 1 : t + 1;
    const bool has = ts >= 0 && ts < T_len;
    const long fs = has ? f + (ts - t) : f;
    float a = 0.f;
    // four pixels per trip with every load issued first (one dependent round trip per pixel before: 72 us per site)
    for (int p0 = s; p0 < hw; p0 += 4 * S) {
      float xv[4], gv[4], xs[4], gs[4], dv[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int p = min(p0 + u * S, hw - 1);
        const long pix = f * hw + p, pixs = fs * hw + p;
        xv[u] = (float)x[pix * C + ci];
        gv[u] = gate[pix * 2 + g];
        xs[u] = (float)x[pixs * C + ci];
        gs[u] = gate[pixs * 2 + g];
        dv[u] = (float)dA[pix * Fp + co];
      }
#pragma unroll
      for (int u = 0; u < 4; ++u)
        if (p0 + u * S < hw) {
          const float r = xv[u] - gv[u] * xv[u];
          const float ys = has ? gs[u] * xs[u] : 0.f;
          a = fmaf(dv[u], ys - r, a);
        }
    }
    red[s * F + ci] = a;
  }
  __syncthreads();
  for (int ci = threadIdx.x; ci < F; ci += 256) {
    float a = 0.f;
    for (int s = 0; s < S; ++s) a += red[s * F + ci];
    d_wgt[f * F + ci] = a;
  }
}

// ---- fusion conv backward, step 1: dpw[b][c][t] = d_wgt * w * (1 - w)
__global__ __launch_bounds__(256) void gsf_bwd_dpw_kernel(const float* __restrict__ d_wgt, const float* __restrict__ fw,
                                                          int T_len, int F, long total, float* __restrict__ dpw) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;          // i = (b*F + c)*T + t
  if (i >= total) return;
  const int t = (int)(i % T_len);
  const long bc = i / T_len;
  const int c = (int)(bc % F);
  const long b = bc / F;
  const float w = fw[i];
  dpw[i] = d_wgt[(b * T_len + t) * F + c] * w * (1.f - w);
}

// ---- step 2: gradients of the two input planes (shifted-y mean, r mean), frame-major [N][F]
__global__ __launch_bounds__(256) void gsf_bwd_planes_kernel(const float* __restrict__ dpw, int T_len, int F,
                                                             const float* __restrict__ cw1, const float* __restrict__ cw2,
                                                             long total, float* __restrict__ d_ym,
                                                             float* __restrict__ d_rm) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;          // i = (b*T + t)*F + c
  if (i >= total) return;
  const int c = (int)(i % F);
  const long ft = i / F;
  const int t = (int)(ft % T_len);
  const long b = ft / T_len;
  const int Fh = F >> 1;
  const int g = c >= Fh, cl = c - g * Fh;
  const float* cw = g ? cw2 : cw1;
  float ay = 0.f, ar = 0.f;
#pragma unroll
  for (int dc = -1; dc <= 1; ++dc) {
    const int c2 = cl - dc;                                     // output position that read this input with tap dc
    if (c2 < 0 || c2 >= Fh) continue;
#pragma unroll
    for (int dt = -1; dt <= 1; ++dt) {
      const int t2 = t - dt;
      if (t2 < 0 || t2 >= T_len) continue;
      const float v = dpw[(b * F + g * Fh + c2) * T_len + t2];
      ay = fmaf(cw[(dc + 1) * 3 + (dt + 1)], v, ay);
      ar = fmaf(cw[9 + (dc + 1) * 3 + (dt + 1)], v, ar);
    }
  }
  d_ym[i] = ay;
  d_rm[i] = ar;
}

// ---- step 3: d channel_conv weights / bias: part[b][g][19] (18 taps: plane 0 = shifted-y mean, plane 1 = r mean; bias)
__global__ __launch_bounds__(256) void gsf_bwd_cw_kernel(const float* __restrict__ dpw, const float* __restrict__ ysum,
                                                         const float* __restrict__ xsum, int T_len, int F, float inv_hw,
                                                         float* __restrict__ part) {
  __shared__ float scratch[8];
  const int b = blockIdx.x, g = blockIdx.y;
  const int Fh = F >> 1;
  float acc[19];
#pragma unroll
  for (int k = 0; k < 19; ++k) acc[k] = 0.f;
  // the (channel, time) plane of one (clip, group) is dealt over gridDim.z workgroups (B x 2 workgroups alone left the chip
  // idle: 61 us for 16 clips); partial rows [b * gridDim.z + z][g][19]
  for (int i = blockIdx.z * 256 + threadIdx.x; i < Fh * T_len; i += 256 * gridDim.z) {
    const int cl = i / T_len, t = i - cl * T_len;
    const float v = dpw[((long)b * F + g * Fh + cl) * T_len + t];
    acc[18] += v;
#pragma unroll
    for (int dc = -1; dc <= 1; ++dc) {
      const int c2 = cl + dc;
      if (c2 < 0 || c2 >= Fh) continue;
      const int cc = g * Fh + c2;
#pragma unroll
      for (int dt = -1; dt <= 1; ++dt) {
        const int t2 = t + dt;
        if (t2 < 0 || t2 >= T_len) continue;
        const long row = ((long)b * T_len + t2) * F + cc;
        const float rm = (xsum[row] - ysum[row]) * inv_hw;
        const int ts = g ? t2 - 1 : t2 + 1;
        const float ysh = (ts >= 0 && ts < T_len) ? ysum[((long)b * T_len + ts) * F + cc] * inv_hw : 0.f;
        acc[(dc + 1) * 3 + (dt + 1)] = fmaf(v, ysh, acc[(dc + 1) * 3 + (dt + 1)]);
        acc[9 + (dc + 1) * 3 + (dt + 1)] = fmaf(v, rm, acc[9 + (dc + 1) * 3 + (dt + 1)]);
      }
    }
  }
#pragma unroll
  for (int k = 0; k < 19; ++k) {
    const float s = block_sum<4>(acc[k], scratch);
    if (threadIdx.x == 0) part[(((long)b * gridDim.z + blockIdx.z) * 2 + g) * 19 + k] = s;
  }
}

// ---- through the blend, the shift and the gate: lane = pixel of frame f.
//   d_ys = dA*w + d_ym/hw,  d_r = dA*(1-w) + d_rm/hw,  d_y[t] = d_ys[t -/+ 1] - d_r[t],
//   d_x(direct) = d_r + d_y*gate,  d_gate = sum_c d_y*x,  d_pre = d_gate * (1 - gate^2)
template <typename T>
__global__ __launch_bounds__(256) void gsf_bwd_gate_kernel(const T* __restrict__ x, const float* __restrict__ gate,
                                                           const float* __restrict__ fw, const T* __restrict__ dA,
                                                           const float* __restrict__ d_ym, const float* __restrict__ d_rm,
                                                           int T_len, int hw, int C, int F, int Fp,
                                                           T* __restrict__ d_xs, float* __restrict__ d_pre) {
  const long f = blockIdx.y;
  const int p = blockIdx.x * 256 + threadIdx.x;
  if (p >= hw) return;
  const int t = (int)(f % T_len);
  const long b = f / T_len;
  const int Fh = F >> 1, Fq = F >> 2;
  const float inv_hw = 1.0f / (float)hw;
  const long pix = f * hw + p;
  float dg[2] = {0.f, 0.f};
  for (int ci = 0; ci < F; ++ci) {
    const int g = ci >= Fh;
    const int co = gsf_out_col(ci, Fh, Fq);
    // gate-shift-FUSE: out = ys * w + r * (1 - w) with the fusion weight w (and its spatial-mean inputs d_ym / d_rm);
    // plain gate-shift (_GSM, fw == nullptr): out = ys + r
    const float w = fw ? fw[(b * F + ci) * T_len + t] : 1.f;
    const float d_o = (float)dA[pix * Fp + co];
    const float d_r = fw ? d_o * (1.f - w) + d_rm[f * F + ci] * inv_hw : d_o;
    // y[t] was read by the output at frame t-1 (g = 0: ys[t-1] = y[t]) or t+1 (g = 1: ys[t+1] = y[t])
    const int tu = g ? t + 1 : t - 1;
    float d_ys = 0.f;
    if (tu >= 0 && tu < T_len) {
      const long fu = f + (tu - t);
      d_ys = fw ? (float)dA[(fu * hw + p) * Fp + co] * fw[(b * F + ci) * T_len + tu] + d_ym[fu * F + ci] * inv_hw
                : (float)dA[(fu * hw + p) * Fp + co];
    }
    const float d_y = d_ys - d_r;
    const float gt = gate[pix * 2 + g];
    const float xv = (float)x[pix * C + ci];
    d_xs[pix * Fp + ci] = (T)(d_r + d_y * gt);
    dg[g] = fmaf(d_y, xv, dg[g]);
  }
  for (int c = F; c < Fp; ++c) d_xs[pix * Fp + c] = dA[pix * Fp + c];        // pass-through pad columns
#pragma unroll
  for (int g = 0; g < 2; ++g) {
    const float gt = gate[pix * 2 + g];
    d_pre[pix * 2 + g] = dg[g] * (1.f - gt * gt);
  }
}

// ---- conv3d (3x3x3, groups = 2) input gradient + ReLU mask: lane = pixel; w3 [F][27] (channel-major), sa/sb = the
// BatchNorm3d affine of this step (a = relu(x*sa + sb) is recomputed for the mask)
template <typename T>
__global__ __launch_bounds__(256) void gsf_bwd_conv3d_dx_kernel(const T* __restrict__ x, const float* __restrict__ d_pre,
                                                                const float* __restrict__ w3,
                                                                const float* __restrict__ sa, const float* __restrict__ sb,
                                                                int T_len, int h, int w, int C, int F, int Fp,
                                                                T* __restrict__ d_bn) {
  extern __shared__ float sw[];        // [F][27]
  for (int i = threadIdx.x; i < F * 27; i += 256) sw[i] = w3[i];
  __syncthreads();
  const long f = blockIdx.y;
  const int hw = h * w;
  const int p = blockIdx.x * 256 + threadIdx.x;
  if (p >= hw) return;
  const int t = (int)(f % T_len);
  const int py = p / w, px = p - py * w;
  const int Fh = F >> 1;
  float dp[2][27];
#pragma unroll
  for (int dt = 0; dt < 3; ++dt) {
    const int t2 = t - dt + 1;                                  // output frame that read this input with tap dt
#pragma unroll
    for (int dy = 0; dy < 3; ++dy) {
      const int y2 = py - dy + 1;
#pragma unroll
      for (int dx = 0; dx < 3; ++dx) {
        const int x2 = px - dx + 1;
        const bool ok = t2 >= 0 && t2 < T_len && y2 >= 0 && y2 < h && x2 >= 0 && x2 < w;
        const long q = ok ? ((f + (t2 - t)) * hw + y2 * w + x2) * 2 : 0;
        dp[0][(dt * 3 + dy) * 3 + dx] = ok ? d_pre[q] : 0.f;
        dp[1][(dt * 3 + dy) * 3 + dx] = ok ? d_pre[q + 1] : 0.f;
      }
    }
  }
  const long pix = f * hw + p;
  for (int c = 0; c < F; ++c) {
    const int g = c >= Fh;
    float a = 0.f;
#pragma unroll
    for (int k = 0; k < 27; ++k) a = fmaf(sw[c * 27 + k], dp[g][k], a);
    const bool on = fmaf((float)x[pix * C + c], sa[c], sb[c]) > 0.f;
    d_bn[pix * Fp + c] = (T)(on ? a : 0.f);
  }
  for (int c = F; c < Fp; ++c) d_bn[pix * Fp + c] = (T)0.f;
}

// 8 consecutive elements <-> fp32 registers (16 bytes of bf16, 2 x 16 bytes of fp32)
template <typename T> __device__ __forceinline__ void ld8(const T* p, float (&v)[8]);
template <> __device__ __forceinline__ void ld8<bf16_t>(const bf16_t* p, float (&v)[8]) { Chunk<bf16_t>::load(p, v); }
template <> __device__ __forceinline__ void ld8<float>(const float* p, float (&v)[8]) {
  const f32x4 a = *reinterpret_cast<const f32x4*>(p), b = *reinterpret_cast<const f32x4*>(p + 4);
#pragma unroll
  for (int e = 0; e < 4; ++e) { v[e] = a[e]; v[4 + e] = b[e]; }
}
template <typename T> __device__ __forceinline__ void st8(T* p, const float (&v)[8]);
template <> __device__ __forceinline__ void st8<bf16_t>(bf16_t* p, const float (&v)[8]) { Chunk<bf16_t>::store(p, v); }
template <> __device__ __forceinline__ void st8<float>(float* p, const float (&v)[8]) {
  *reinterpret_cast<f32x4*>(p) = f32x4{v[0], v[1], v[2], v[3]};
  *reinterpret_cast<f32x4*>(p + 4) = f32x4{v[4], v[5], v[6], v[7]};
}

// ---- coalesced forms of the two pixel-wise kernels above (bf16 / fp32).  A workgroup owns PT consecutive pixels of one
// frame; lane = (pixel, 8-channel chunk in SOURCE-channel order).  The rows of dA (this frame and its two temporal
// neighbours) are contiguous in memory and go to LDS with 16-byte loads; the output-column interleave co(ci) is then an LDS
// index instead of a 2-byte global gather, x is read and d_xs / d_bn are written as 16-byte chunks, the per-pixel sums
// over channels (d gate) are folded through LDS in a fixed order.
template <typename T>
__global__ __launch_bounds__(256) void gsf_bwd_gate2_kernel(const T* __restrict__ x, const float* __restrict__ gate,
                                                            const float* __restrict__ fw, const T* __restrict__ dA,
                                                            const float* __restrict__ d_ym, const float* __restrict__ d_rm,
                                                            int T_len, int hw, int C, int F, int Fp, int PT,
                                                            T* __restrict__ d_xs, float* __restrict__ d_pre) {
  constexpr int EPC = 8;
  extern __shared__ __attribute__((aligned(16))) unsigned char smraw[];
  const int NCH = Fp / EPC;
  T* tA = reinterpret_cast<T*>(smraw);                         // [3][PT][Fp]: frame f, f-1, f+1
  float* par = reinterpret_cast<float*>(tA + 3 * PT * Fp);     // [4][F]: w(t), d_rm/hw, w(tu), d_ym(tu)/hw
  float* dgp = par + 4 * F;                                    // [PT][NCH][2]
  const long f = blockIdx.y;
  const int p0 = blockIdx.x * PT;
  const int np = min(PT, hw - p0);
  const int t = (int)(f % T_len);
  const long b = f / T_len;
  const int Fh = F >> 1, Fq = F >> 2;
  const float inv_hw = 1.0f / (float)hw;
  const int tid = threadIdx.x;
  // ---- stage dA rows (contiguous: np * Fp elements per frame)
  {
    const int n16 = np * Fp / EPC;                             // 16-byte (bf16) / 32-byte (fp32) chunks of 8 elements
    for (int fi = 0; fi < 3; ++fi) {
      const int tt = t + (fi == 0 ? 0 : (fi == 1 ? -1 : 1));
      const bool ok = tt >= 0 && tt < T_len;
      const T* src = dA + ((f + (tt - t)) * hw + p0) * (long)Fp;
      T* dst = tA + fi * PT * Fp;
      for (int i = tid; i < n16; i += 256) {
        T v[EPC];
        if (ok) {
#pragma unroll
          for (int e = 0; e < EPC; ++e) v[e] = src[(long)i * EPC + e];
        } else {
#pragma unroll
          for (int e = 0; e < EPC; ++e) v[e] = (T)0.f;
        }
#pragma unroll
        for (int e = 0; e < EPC; ++e) dst[i * EPC + e] = v[e];
      }
    }
    for (int ci = tid; ci < F; ci += 256) {
      const int g = ci >= Fh;
      const int tu = g ? t + 1 : t - 1;
      const bool ok = tu >= 0 && tu < T_len;
      par[ci] = fw ? fw[(b * F + ci) * T_len + t] : 1.f;
      par[F + ci] = fw ? d_rm[f * F + ci] * inv_hw : 0.f;
      par[2 * F + ci] = ok ? (fw ? fw[(b * F + ci) * T_len + tu] : 1.f) : 0.f;
      par[3 * F + ci] = (ok && fw) ? d_ym[(f + (tu - t)) * F + ci] * inv_hw : 0.f;
    }
  }
  __syncthreads();
  const int pl = tid / NCH, k = tid - pl * NCH;
  float dg0 = 0.f, dg1 = 0.f;
  const bool act = pl < np;
  if (act) {
    const long pix = f * hw + p0 + pl;
    const float g0 = gate[pix * 2], g1 = gate[pix * 2 + 1];
    float xv[EPC], o[EPC];
    ld8<T>(x + pix * C + k * EPC, xv);                         // Fp <= C: the chunk lies inside the row
    const T* rA = tA + pl * Fp;
    const T* rP = tA + (PT + pl) * Fp;
    const T* rN = tA + (2 * PT + pl) * Fp;
#pragma unroll
    for (int e = 0; e < EPC; ++e) {
      const int ci = k * EPC + e;
      if (ci < F) {
        const int g = ci >= Fh, cl = ci - g * Fh;
        const int i2 = cl >= Fq, j = cl - i2 * Fq;
        const int co = g * Fh + 2 * j + i2;
        const float d_o = (float)rA[co];
        const float d_r = fw ? d_o * (1.f - par[ci]) + par[F + ci] : d_o;
        // y[t] was read by the output at frame t-1 (g = 0) or t+1 (g = 1)
        const float dau = (float)(g ? rN[co] : rP[co]);
        const float d_ys = dau * par[2 * F + ci] + par[3 * F + ci];
        const float d_y = d_ys - d_r;
        const float gt = g ? g1 : g0;
        o[e] = d_r + d_y * gt;
        if (g) dg1 = fmaf(d_y, xv[e], dg1); else dg0 = fmaf(d_y, xv[e], dg0);
      } else {
        o[e] = (float)rA[ci];                                  // pass-through pad columns
      }
    }
    st8<T>(d_xs + pix * Fp + k * EPC, o);
    dgp[(pl * NCH + k) * 2] = dg0;
    dgp[(pl * NCH + k) * 2 + 1] = dg1;
  }
  __syncthreads();
  for (int i = tid; i < np * 2; i += 256) {
    const int pp = i >> 1, g = i & 1;
    float a = 0.f;
    for (int kk = 0; kk < NCH; ++kk) a += dgp[(pp * NCH + kk) * 2 + g];
    const long pix = f * hw + p0 + pp;
    const float gt = gate[pix * 2 + g];
    d_pre[pix * 2 + g] = a * (1.f - gt * gt);
  }
}

template <typename T>
__global__ __launch_bounds__(256) void gsf_bwd_conv3d_dx2_kernel(const T* __restrict__ x, const float* __restrict__ d_pre,
                                                                 const float* __restrict__ w3,
                                                                 const float* __restrict__ sa, const float* __restrict__ sb,
                                                                 int T_len, int h, int w, int C, int F, int Fp, int PT,
                                                                 T* __restrict__ d_bn, const float* __restrict__ bn_mean,
                                                                 float* __restrict__ bn_part) {
  // bn_part (training): d_bn enters the BatchNorm3d of the module, whose backward needs sum g and sum g * (x - mean) over all
  // pixels: this workgroup's partial row bn_part[(f * gridDim.x + blockIdx.x)][3][Fp] (rows 0 and 1; of the STORED values)
  // replaces the column-statistics pass over (d_bn, x) -- tdeed_bn_bwd_from_parts folds the rows.
  constexpr int EPC = 8;
  extern __shared__ __attribute__((aligned(16))) unsigned char smraw[];
  float* sw = reinterpret_cast<float*>(smraw);                 // [27][Fp]
  float* dp = sw + Fp * 27;                                    // [PT][2][27] (+1 pad per row)
  float* aff = dp + PT * 55;                                   // [2][F]
  float* bred = aff + 2 * F;                                   // [PT][2][Fp] (bn_part only)
  const int NCH = Fp / EPC;
  const long f = blockIdx.y;
  const int hw = h * w;
  const int p0 = blockIdx.x * PT;
  const int np = min(PT, hw - p0);
  const int t = (int)(f % T_len);
  const int Fh = F >> 1;
  const int tid = threadIdx.x;
  // weights tap-major [27][Fp] (zero in the pad columns): a lane's 8 channels of one tap are two 16-byte reads instead of
  // eight 4-byte ones (the kernel is bound by its LDS reads: 2 per FMA before, 3/8 now)
  // (the global reads walk w3 [F][27] in memory order -- 8.6 KB = 68 lines per workgroup, scattered into the tap-major LDS image;
  // walking the LDS image instead made every one of the 27 F scalar loads a different cache line, 12 800 workgroups over)
  for (int i = tid; i < F * 27; i += 256) {
    const int c = i / 27, kq = i - c * 27;
    sw[kq * Fp + c] = w3[i];
  }
  for (int i = tid; i < (Fp - F) * 27; i += 256) {
    const int kq = i / (Fp - F), c = F + i - kq * (Fp - F);
    sw[kq * Fp + c] = 0.f;
  }
  for (int i = tid; i < F; i += 256) { aff[i] = sa[i]; aff[F + i] = sb[i]; }
  for (int i = tid; i < np * 27; i += 256) {
    const int pp = i / 27, kq = i - pp * 27;
    const int dt = kq / 9, dy = (kq / 3) % 3, dx = kq % 3;
    const int p = p0 + pp;
    const int py = p / w, px = p - py * w;
    const int t2 = t - dt + 1, y2 = py - dy + 1, x2 = px - dx + 1;      // output position that read this input with tap kq
    const bool ok = t2 >= 0 && t2 < T_len && y2 >= 0 && y2 < h && x2 >= 0 && x2 < w;
    const long q = ok ? ((f + (t2 - t)) * hw + y2 * w + x2) * 2 : 0;
    dp[pp * 55 + kq] = ok ? d_pre[q] : 0.f;
    dp[pp * 55 + 27 + kq] = ok ? d_pre[q + 1] : 0.f;
  }
  __syncthreads();
  const int pl = tid / NCH, k = tid - pl * NCH;
  const bool act = pl < np;
  if (!act && !bn_part) return;
  float xv[EPC], o[EPC];
#pragma unroll
  for (int e = 0; e < EPC; ++e) { xv[e] = 0.f; o[e] = 0.f; }
  if (act) {
    const long pix = f * hw + p0 + pl;
    ld8<T>(x + pix * C + k * EPC, xv);
    // a chunk lies in one gate group unless it straddles F/2 (F/2 not a multiple of 8): per-element group then
    const bool one_g = (k * EPC + EPC <= Fh) || (k * EPC >= Fh);
    const float* d0 = dp + pl * 55;
    if (one_g) {
      const float* d = d0 + (k * EPC >= Fh ? 27 : 0);
#pragma unroll
      for (int kq = 0; kq < 27; ++kq) {
        const float dv = d[kq];
        const f32x4 w0 = *reinterpret_cast<const f32x4*>(sw + kq * Fp + k * EPC);
        const f32x4 w1 = *reinterpret_cast<const f32x4*>(sw + kq * Fp + k * EPC + 4);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          o[e] = fmaf(w0[e], dv, o[e]);
          o[4 + e] = fmaf(w1[e], dv, o[4 + e]);
        }
      }
    } else {
#pragma unroll
      for (int kq = 0; kq < 27; ++kq) {
        const float da = d0[kq], db = d0[27 + kq];
#pragma unroll
        for (int e = 0; e < EPC; ++e) o[e] = fmaf(sw[kq * Fp + k * EPC + e], (k * EPC + e >= Fh) ? db : da, o[e]);
      }
    }
#pragma unroll
    for (int e = 0; e < EPC; ++e) {
      const int c = k * EPC + e;
      if (c >= F || !(fmaf(xv[e], aff[c], aff[F + c]) > 0.f)) o[e] = 0.f;
    }
    st8<T>(d_bn + pix * Fp + k * EPC, o);
  }
  if (!bn_part) return;
  if (pl < PT) {
#pragma unroll
    for (int e = 0; e < EPC; ++e) {
      const int c = k * EPC + e;
      const float r = act ? round_to<T>(o[e]) : 0.f;
      bred[(pl * 2 + 0) * Fp + c] = r;
      bred[(pl * 2 + 1) * Fp + c] = (act && c < F) ? r * (xv[e] - bn_mean[c]) : 0.f;
    }
  }
  __syncthreads();
  for (int j = tid; j < 2 * Fp; j += 256) {
    float a = 0.f;
    for (int i = 0; i < PT; ++i) a += bred[i * 2 * Fp + j];
    bn_part[((f * gridDim.x + blockIdx.x) * 3) * Fp + j] = a;
  }
}

// ---- conv3d weight gradient: workgroup = frame f'; a[f'] (relu(bn(x))) tile in LDS, d_pre of frames f'-1..f'+1 with a
// zero ring; lane (c, tap) walks the pixels.  part[f'][F*27 + 2] (the last two: d bias = sum_p d_pre[f'][p][g])
template <typename T>
__global__ __launch_bounds__(256) void gsf_bwd_conv3d_dw_kernel(const T* __restrict__ x, const float* __restrict__ d_pre,
                                                                const float* __restrict__ sa, const float* __restrict__ sb,
                                                                int T_len, int h, int w, int C, int F,
                                                                float* __restrict__ part) {
  extern __shared__ float sm[];
  const int hw = h * w, WP = w + 2, HP = h + 2;
  float* at = sm;                                               // [hw][F]
  float* dpt = at + (size_t)hw * F;                             // [3][HP][WP][2]
  __shared__ float scratch[8];
  const long f = blockIdx.x;
  const int t = (int)(f % T_len);
  if (F % 8 == 0) {                                             // 8 channels per load (16 bytes of bf16)
    const int nck = F / 8;
    const IDiv dck(nck);
    for (int i0 = threadIdx.x; i0 < hw * nck; i0 += 256 * 4) {
      float v[4][8];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int i = min(i0 + u * 256, hw * nck - 1);
        int p, ck;
        dck.divmod(i, p, ck);
        ld8<T>(x + (f * hw + p) * C + ck * 8, v[u]);
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int i = i0 + u * 256;
        if (i < hw * nck) {
          int p, ck;
          dck.divmod(i, p, ck);
#pragma unroll
          for (int e = 0; e < 8; ++e) at[p * F + ck * 8 + e] = fmaxf(fmaf(v[u][e], sa[ck * 8 + e], sb[ck * 8 + e]), 0.f);
        }
      }
    }
  } else {
    for (int i = threadIdx.x; i < hw * F; i += 256) {
      const int p = i / F, c = i - p * F;
      at[i] = fmaxf(fmaf((float)x[(f * hw + p) * C + c], sa[c], sb[c]), 0.f);
    }
  }
  for (int i = threadIdx.x; i < 3 * HP * WP * 2; i += 256) {
    const int g = i & 1;
    const int r = i >> 1;
    const int xx = r % WP, yy = (r / WP) % HP, k = r / (WP * HP);
    const int t2 = t + k - 1, y2 = yy - 1, x2 = xx - 1;
    float v = 0.f;
    if (t2 >= 0 && t2 < T_len && y2 >= 0 && y2 < h && x2 >= 0 && x2 < w)
      v = d_pre[((f + (k - 1)) * hw + y2 * w + x2) * 2 + g];
    dpt[i] = v;
  }
  __syncthreads();
  const int Fh = F >> 1;
  // out frame t_out = t' - dt + 1 uses a[t'] with tap dt; its pixel = p - (dy-1, dx-1)
  for (int o = threadIdx.x; o < F * 27; o += 256) {
    const int c = o / 27, k = o - c * 27;
    const int dt = k / 9, dy = (k / 3) % 3, dx = k % 3;
    const int g = c >= Fh;
    const int slab = 2 - dt;                                    // frame t'-dt+1 sits in slab (t_out - t') + 1 = 2 - dt
    float a = 0.f;
    // rows and columns walked by two nested loops (no per-pixel division); ring coordinates: +1
    const float* dbase = dpt + (((slab * HP) + (2 - dy)) * WP + (2 - dx)) * 2 + g;
    for (int py = 0; py < h; ++py) {
      const float* arow = at + (py * w) * F + c;
      const float* drow = dbase + py * WP * 2;
#pragma unroll 4
      for (int px = 0; px < w; ++px) a = fmaf(arow[px * F], drow[px * 2], a);
    }
    part[f * (F * 27 + 2) + o] = a;
  }
  float b0 = 0.f, b1 = 0.f;
  for (int p = threadIdx.x; p < hw; p += 256) {
    const int py = p / w, px = p - py * w;
    b0 += dpt[(((1 * HP) + py + 1) * WP + px + 1) * 2 + 0];
    b1 += dpt[(((1 * HP) + py + 1) * WP + px + 1) * 2 + 1];
  }
  b0 = block_sum<4>(b0, scratch);
  b1 = block_sum<4>(b1, scratch);
  if (threadIdx.x == 0) {
    part[f * (F * 27 + 2) + F * 27] = b0;
    part[f * (F * 27 + 2) + F * 27 + 1] = b1;
  }
}

// ---- the same weight gradient on the MFMA pipe (bf16): per frame f' a GEMM  D[c][(g, tap)] = sum_p a[f'][p][c] * B[p][(g, tap)]
// with B[p][(g, tap)] = d_pre[g][f' - dt + 1][p - (dy - 1, dx - 1)].  The contraction index is the pixel: a (relu(bn(x)),
// bf16) goes to LDS as a row-major [pixel][channel] image and its operand -- one channel, 8 consecutive pixels -- comes
// from two transposing reads; the B operand (54 of 64 columns used) is gathered from the zero-ringed d_pre tile, 8 scalar
// LDS reads per lane and 32-pixel step, shared by all channel tiles.  Both gate groups are computed for a channel tile
// that straddles F/2; the output keeps each channel's own group.  Same partial layout as the kernel above.
__global__ __launch_bounds__(256) void gsf_bwd_conv3d_dw_mfma_kernel(const bf16_t* __restrict__ x,
                                                                     const float* __restrict__ d_pre,
                                                                     const float* __restrict__ sa,
                                                                     const float* __restrict__ sb, int T_len, int h, int w,
                                                                     int C, int F, int RS, int KST, float* __restrict__ part) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smraw[];
  const int hw = h * w, WP = w + 2, HP = h + 2;
  bf16_t* at = reinterpret_cast<bf16_t*>(smraw);                // [KST * 32][RS]
  float* dpt = reinterpret_cast<float*>(at + (size_t)KST * 32 * RS);      // [3][HP][WP][2]
  int* poff = reinterpret_cast<int*>(dpt + 3 * HP * WP * 2);    // [KST * 32]: ring offset of pixel p (or of a zero cell)
  __shared__ float scratch[8];
  const long f = blockIdx.x;
  const int t = (int)(f % T_len);
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int nck = (F + 7) >> 3, NP = KST * 32;
  {
    const IDiv dck(nck);
    for (int i0 = tid; i0 < NP * nck; i0 += 256 * 4) {
      u32x4 v[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int i = min(i0 + u * 256, NP * nck - 1);
        int p, ck;
        dck.divmod(i, p, ck);
        v[u] = *reinterpret_cast<const u32x4*>(x + (f * hw + min(p, hw - 1)) * C + ck * 8);
      }
      TD_ISSUE_FENCE();
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int i = i0 + u * 256;
        if (i < NP * nck) {
          int p, ck;
          dck.divmod(i, p, ck);
          const bf16x8 t8 = *reinterpret_cast<const bf16x8*>(&v[u]);
          bf16x8 o;
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            const int c = ck * 8 + e, cc = min(c, F - 1);
            const float a = fmaxf(fmaf((float)t8[e], sa[cc], sb[cc]), 0.f);
            o[e] = (p < hw && c < F) ? (bf16_t)a : (bf16_t)0.f;
          }
          *reinterpret_cast<bf16x8*>(at + (size_t)p * RS + ck * 8) = o;
        }
      }
    }
  }
  for (int i = tid; i < 3 * HP * WP * 2; i += 256) {
    const int g = i & 1;
    const int r = i >> 1;
    const int xx = r % WP, yy = (r / WP) % HP, k = r / (WP * HP);
    const int t2 = t + k - 1, y2 = yy - 1, x2 = xx - 1;
    float v = 0.f;
    if (t2 >= 0 && t2 < T_len && y2 >= 0 && y2 < h && x2 >= 0 && x2 < w)
      v = d_pre[((f + (k - 1)) * hw + y2 * w + x2) * 2 + g];
    dpt[i] = v;
  }
  for (int p = tid; p < NP; p += 256) {
    const int py = p / w, px = p - py * w;
    poff[p] = p < hw ? ((py + 1) * WP + px + 1) * 2 : 0;         // ring cell (0, 0) of slab 0..2 is zero: a pad pixel's a is 0 anyway
  }
  __syncthreads();
  const int Fh = F >> 1;
  const int MT = (F + 15) >> 4;
  const int g4 = lane >> 4, q4 = (lane >> 2) & 3, p4 = lane & 3, pl = lane & 15;
  // jobs = (channel tile, gate group the tile touches, half of that group's 32 columns), dealt round-robin to the 4 waves
  int njobs = 0;
  for (int mt = 0; mt < MT; ++mt) njobs += ((mt * 16 < Fh) ? 2 : 0) + ((mt * 16 + 15 >= Fh) ? 2 : 0);
  for (int job = wv; job < njobs; job += 4) {
    int mt = 0, g = 0, nh = 0, acc_j = job;
    for (mt = 0; mt < MT; ++mt) {
      const int n0 = (mt * 16 < Fh) ? 2 : 0, n1 = (mt * 16 + 15 >= Fh) ? 2 : 0;
      if (acc_j < n0) { g = 0; nh = acc_j; break; }
      acc_j -= n0;
      if (acc_j < n1) { g = 1; nh = acc_j; break; }
      acc_j -= n1;
    }
    const int tap = nh * 16 + pl;                                // this lane's B column
    const bool tok = tap < 27;
    const int tc = tok ? tap : 0;
    const int dt = tc / 9, dy = (tc / 3) % 3, dx = tc % 3;
    // B[p][(g, tap)] = dpt[slab 2 - dt][py + 2 - dy][px + 2 - dx][g] = dpt[base + poff[p]], poff = ((py + 1) WP + px + 1) 2
    const int base = (((2 - dt) * HP + (1 - dy)) * WP + (1 - dx)) * 2 + g;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int ks = 0; ks < KST; ++ks) {
      const int row = ks * 32 + g4 * 8;
      const bf16x8 af = td_tr_read8(at + (size_t)(row + q4) * RS + mt * 16 + p4 * 4, at + (size_t)(row + q4 + 4) * RS + mt * 16 + p4 * 4);
      bf16x8 bfr;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int p = row + e;
        const float v = dpt[(p < hw ? base : g) + poff[p]];
        bfr[e] = tok ? (bf16_t)v : (bf16_t)0.f;
      }
      acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af, bfr, acc, 0, 0, 0);
    }
    if (tok) {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int c = mt * 16 + 4 * g4 + e;
        if (c < F && (c >= Fh) == (g == 1)) part[f * (F * 27 + 2) + (long)c * 27 + tap] = acc[e];
      }
    }
  }
  float b0 = 0.f, b1 = 0.f;
  for (int p = tid; p < hw; p += 256) {
    b0 += dpt[HP * WP * 2 + poff[p]];
    b1 += dpt[HP * WP * 2 + poff[p] + 1];
  }
  b0 = block_sum<4>(b0, scratch);
  b1 = block_sum<4>(b1, scratch);
  if (tid == 0) {
    part[f * (F * 27 + 2) + F * 27] = b0;
    part[f * (F * 27 + 2) + F * 27 + 1] = b1;
  }
}

// ---- dx[m][c] += a[m][c] + b[m][c] for c < Fp (dx row stride C; a, b dense [M][Fp])
template <typename T>
__global__ __launch_bounds__(256) void gsf_add_cols_kernel(const T* __restrict__ a, const T* __restrict__ b, long M, int C,
                                                           int Fp, T* __restrict__ dx) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= M * Fp) return;
  const long m = i / Fp;
  const int c = (int)(i - m * Fp);
  dx[m * C + c] = (T)((float)dx[m * C + c] + (float)a[i] + (float)b[i]);
}

extern "C" int tdeed_gsf_slice(const void* x, long M, int C, int F, int Fp, void* xs, int dtype, void* stream) {
  TD_CHECK(x && xs && M > 0 && F > 0 && Fp >= F && Fp <= C, "gsf_slice: bad arguments");
  TD_CHECK(dtype == TDEED_F32 || dtype == TDEED_BF16, "gsf_slice: bad dtype %d", dtype);
  hipStream_t st = (hipStream_t)stream;
  const dim3 grid((unsigned)((M * Fp + 255) / 256));
  if (dtype == TDEED_F32) hipLaunchKernelGGL(gsf_slice_kernel<float>, grid, dim3(256), 0, st, (const float*)x, M, C, F, Fp, (float*)xs);
  else hipLaunchKernelGGL(gsf_slice_kernel<bf16_t>, grid, dim3(256), 0, st, (const bf16_t*)x, M, C, F, Fp, (bf16_t*)xs);
  TD_LAUNCH_CHECK("gsf_slice");
  return TDEED_OK;
}

constexpr int GSF_CW_Z = 8;       // workgroups per (clip, gate group) of the fusion-conv weight gradient
extern "C" long tdeed_gsf_bwd_scratch_floats(int B, int T, int hw, int F) {
  const long N = (long)B * T;
  // d_wgt [N][F], dpw [B][F][T], d_ym [N][F], d_rm [N][F], d_pre [N][hw][2], part_cw [B][2][19], part_w3 [N][F*27+2]
  return 4 * N * F + N * hw * 2 + (long)B * GSF_CW_Z * 38 + N * ((long)F * 27 + 2);
}

// Where the parameter-gradient partials of tdeed_gsf_bwd(d_w3 == NULL) lie inside `scratch` (float offsets), for a caller
// that folds them itself: out = {off_cw, rows_cw, stride_cw, off_w3, rows_w3, stride_w3}.  part_cw rows [rows_cw][38]:
// columns 0..17 channel_conv1 taps, 18 its bias, 19..36 channel_conv2 taps, 37 its bias; part_w3 rows [rows_w3][27 F + 2]:
// conv3D.weight as [F][27], then the two biases.
extern "C" int tdeed_gsf_bwd_part_layout(int B, int T, int hw, int F, long* out) {
  TD_CHECK(out && B > 0 && T > 0 && hw > 0 && F > 0, "gsf_bwd_part_layout: bad arguments");
  const long N = (long)B * T;
  out[0] = 4 * N * F + N * hw * 2;
  out[1] = (long)B * GSF_CW_Z;
  out[2] = 38;
  out[3] = out[0] + out[1] * 38;
  out[4] = N;
  out[5] = (long)F * 27 + 2;
  return TDEED_OK;
}

template <typename T>
static int gsf_bwd_launch(const void* x_, const float* gate, const float* fw, const float* ysum, const float* xsum,
                          const void* dA_, int B, int T_len, int h, int w, int C, int F, int Fp, const float* w3,
                          const float* sa, const float* sb, const float* cw1, const float* cw2, float* scratch,
                          void* d_xs_, void* d_bn_, const float* bn_mean, float* bn_part, hipStream_t st) {
  const T* x = (const T*)x_;
  const T* dA = (const T*)dA_;
  T* d_xs = (T*)d_xs_;
  T* d_bn = (T*)d_bn_;
  const int hw = h * w;
  const long N = (long)B * T_len;
  float* d_wgt = scratch;
  float* dpw = d_wgt + N * F;
  float* d_ym = dpw + N * F;
  float* d_rm = d_ym + N * F;
  float* d_pre = d_rm + N * F;
  float* part_cw = d_pre + N * hw * 2;
  float* part_w3 = part_cw + (long)B * GSF_CW_Z * 38;
  const int S = 256 / F > 0 ? 256 / F : 1;
  if (fw) {          // the fusion-weight path exists in gate-shift-FUSE only
    hipLaunchKernelGGL(gsf_bwd_dw_kernel<T>, dim3((unsigned)N), dim3(256), (size_t)S * F * sizeof(float), st, x, gate, dA,
                       T_len, hw, C, F, Fp, d_wgt);
    TD_LAUNCH_CHECK("gsf_bwd_dw");
    const long nft = N * F;
    hipLaunchKernelGGL(gsf_bwd_dpw_kernel, dim3((unsigned)((nft + 255) / 256)), dim3(256), 0, st, d_wgt, fw, T_len, F, nft,
                       dpw);
    hipLaunchKernelGGL(gsf_bwd_planes_kernel, dim3((unsigned)((nft + 255) / 256)), dim3(256), 0, st, dpw, T_len, F, cw1,
                       cw2, nft, d_ym, d_rm);
    hipLaunchKernelGGL(gsf_bwd_cw_kernel, dim3(B, 2, GSF_CW_Z), dim3(256), 0, st, dpw, ysum, xsum, T_len, F, 1.0f / (float)hw,
                       part_cw);
    TD_LAUNCH_CHECK("gsf_bwd fuse");
  }
  const dim3 gpix(cdiv(hw, 256), (unsigned)N);
  if (Fp % 8 == 0 && Fp <= C && Fp / 8 <= 64) {
    // coalesced forms: PT pixels x (Fp / 8) channel chunks per workgroup
    const int NCH = Fp / 8;
    const int PT = 256 / NCH;
    const dim3 g2(cdiv(hw, PT), (unsigned)N);
    const size_t sm_g = (size_t)3 * PT * Fp * sizeof(T) + (size_t)(4 * F + PT * NCH * 2) * sizeof(float);
    hipLaunchKernelGGL(gsf_bwd_gate2_kernel<T>, g2, dim3(256), sm_g, st, x, gate, fw, dA, d_ym, d_rm, T_len, hw, C, F, Fp, PT,
                       d_xs, d_pre);
    TD_LAUNCH_CHECK("gsf_bwd_gate2");
    const size_t sm_d = (size_t)(Fp * 27 + PT * 55 + 2 * F + (bn_part ? 2 * PT * Fp : 0)) * sizeof(float);
    hipLaunchKernelGGL(gsf_bwd_conv3d_dx2_kernel<T>, g2, dim3(256), sm_d, st, x, d_pre, w3, sa, sb, T_len, h, w, C, F, Fp,
                       PT, d_bn, bn_mean, bn_part);
    TD_LAUNCH_CHECK("gsf_bwd_conv3d_dx2");
  } else {
    TD_CHECK(!bn_part, "gsf_bwd: the BatchNorm3d statistics partials come from the coalesced kernels only (tdeed_gsf_bwd_bn_parts)");
    hipLaunchKernelGGL(gsf_bwd_gate_kernel<T>, gpix, dim3(256), 0, st, x, gate, fw, dA, d_ym, d_rm, T_len, hw, C, F, Fp, d_xs,
                       d_pre);
    TD_LAUNCH_CHECK("gsf_bwd_gate");
    hipLaunchKernelGGL(gsf_bwd_conv3d_dx_kernel<T>, gpix, dim3(256), (size_t)F * 27 * sizeof(float), st, x, d_pre, w3, sa,
                       sb, T_len, h, w, C, F, Fp, d_bn);
    TD_LAUNCH_CHECK("gsf_bwd_conv3d_dx");
  }
  if constexpr (sizeof(T) == 2) {
    const int KST = (hw + 31) / 32, MTn = (F + 15) / 16;
    const int RS = MTn * 16 + 16;                               // row stride (elements): whole channel tiles + 32 bytes
    const size_t smm = (size_t)KST * 32 * RS * 2 + (size_t)3 * (h + 2) * (w + 2) * 2 * 4 + (size_t)KST * 32 * 4;
    if (smm <= 150 * 1024 && ((F + 7) / 8) * 8 <= C) {
      hipError_t e2 = hipFuncSetAttribute((const void*)gsf_bwd_conv3d_dw_mfma_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                          150 * 1024);
      if (e2 != hipSuccess) { tdeed_set_error("gsf_bwd: hipFuncSetAttribute: %s", hipGetErrorString(e2)); return TDEED_ERR_RUNTIME; }
      hipLaunchKernelGGL(gsf_bwd_conv3d_dw_mfma_kernel, dim3((unsigned)N), dim3(256), smm, st, (const bf16_t*)x, d_pre, sa, sb,
                         T_len, h, w, C, F, RS, KST, part_w3);
      TD_LAUNCH_CHECK("gsf_bwd_conv3d_dw_mfma");
      return TDEED_OK;
    }
  }
  const size_t smw = ((size_t)hw * F + (size_t)3 * (h + 2) * (w + 2) * 2) * sizeof(float);
  TD_CHECK(smw <= 150 * 1024, "gsf_bwd: frame %dx%d x %d channels does not fit LDS", h, w, F);
  hipError_t e = hipFuncSetAttribute((const void*)gsf_bwd_conv3d_dw_kernel<T>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                     150 * 1024);
  if (e != hipSuccess) { tdeed_set_error("gsf_bwd: hipFuncSetAttribute: %s", hipGetErrorString(e)); return TDEED_ERR_RUNTIME; }
  hipLaunchKernelGGL(gsf_bwd_conv3d_dw_kernel<T>, dim3((unsigned)N), dim3(256), smw, st, x, d_pre, sa, sb, T_len, h, w, C, F,
                     part_w3);
  TD_LAUNCH_CHECK("gsf_bwd_conv3d_dw");
  return TDEED_OK;
}

// Everything between the module output gradient dA and (a) the direct part of d x (d_xs, dense [M][Fp]) and (b) the
// gradient entering the BatchNorm3d output after the ReLU mask (d_bn, dense [M][Fp]); parameter gradients:
// d_w3 [F][27] (= conv3D.weight (2, F/2, 3,3,3) flattened), d_b3 [2], d_cw [2][18] + d_cb [2] (channel_conv1 | 2).
// sa/sb: the BatchNorm3d affine used by the forward of this step.  scratch: tdeed_gsf_bwd_scratch_floats() fp32.
static int gsf_bwd_entry(const void* x, const float* gate, const float* fw, const float* ysum, const float* xsum,
                         const void* dA, int B, int T, int h, int w, int C, int F, int Fp, const float* w3,
                         const float* sa, const float* sb, const float* cw1, const float* cw2, float* scratch,
                         void* d_xs, void* d_bn, float* d_w3, float* d_b3, float* d_cw, float* d_cb, const float* bn_mean,
                         float* bn_part, int dtype, void* stream) {
  // fw == NULL selects the plain gate-shift module (_GSM, impl/gsm.py:89-116: out = shift(gate*x) + (x - gate*x), no
  // fusion conv): ysum / xsum / cw1 / cw2 / d_cw / d_cb are then unused and may be NULL
  // d_w3 == NULL: the parameter gradients stay partials in `scratch` (part_cw [B * 8][38] at float offset 4 N F + 2 N hw, then
  // part_w3 [N][27 F + 2]; column layout in the fold calls below) for a caller that folds them itself (gradient write-out)
  TD_CHECK(x && gate && dA && w3 && sa && sb && scratch && d_xs && d_bn && (!d_w3 == !d_b3), "gsf_bwd: null pointer");
  TD_CHECK(!fw || (ysum && xsum && cw1 && cw2 && (!d_w3 || (d_cw && d_cb))), "gsf_bwd: the fuse path needs ysum, xsum, cw1, cw2, d_cw, d_cb");
  TD_CHECK(B > 0 && T > 0 && h > 0 && w > 0 && F > 0 && F % 4 == 0 && Fp >= F && Fp <= C && F <= 256, "gsf_bwd: bad sizes");
  TD_CHECK(dtype == TDEED_F32 || dtype == TDEED_BF16, "gsf_bwd: bad dtype %d", dtype);
  hipStream_t st = (hipStream_t)stream;
  int rc = dtype == TDEED_F32
               ? gsf_bwd_launch<float>(x, gate, fw, ysum, xsum, dA, B, T, h, w, C, F, Fp, w3, sa, sb, cw1, cw2, scratch, d_xs,
                                       d_bn, bn_mean, bn_part, st)
               : gsf_bwd_launch<bf16_t>(x, gate, fw, ysum, xsum, dA, B, T, h, w, C, F, Fp, w3, sa, sb, cw1, cw2, scratch,
                                        d_xs, d_bn, bn_mean, bn_part, st);
  if (rc != TDEED_OK || !d_w3) return rc;
  const long N = (long)B * T;
  float* part_cw = scratch + 4 * N * F + N * h * w * 2;
  float* part_w3 = part_cw + (long)B * GSF_CW_Z * 38;
  // fold the per-frame / per-clip partials (rows hold several parameter groups side by side)
  const long row = (long)F * 27 + 2;
  rc = tdeed_reduce_strided(part_w3, (int)N, row, (long)F * 27, d_w3, stream);
  if (rc == TDEED_OK) rc = tdeed_reduce_strided(part_w3 + (long)F * 27, (int)N, row, 2, d_b3, stream);
  if (!fw) return rc;
  if (rc == TDEED_OK) rc = tdeed_reduce_strided(part_cw, B * GSF_CW_Z, 38, 18, d_cw, stream);                 // channel_conv1 taps
  if (rc == TDEED_OK) rc = tdeed_reduce_strided(part_cw + 19, B * GSF_CW_Z, 38, 18, d_cw + 18, stream);       // channel_conv2 taps
  if (rc == TDEED_OK) rc = tdeed_reduce_strided(part_cw + 18, B * GSF_CW_Z, 38, 1, d_cb, stream);
  if (rc == TDEED_OK) rc = tdeed_reduce_strided(part_cw + 37, B * GSF_CW_Z, 38, 1, d_cb + 1, stream);
  return rc;
}

extern "C" int tdeed_gsf_bwd(const void* x, const float* gate, const float* fw, const float* ysum, const float* xsum,
                             const void* dA, int B, int T, int h, int w, int C, int F, int Fp, const float* w3,
                             const float* sa, const float* sb, const float* cw1, const float* cw2, float* scratch,
                             void* d_xs, void* d_bn, float* d_w3, float* d_b3, float* d_cw, float* d_cb, int dtype,
                             void* stream) {
  return gsf_bwd_entry(x, gate, fw, ysum, xsum, dA, B, T, h, w, C, F, Fp, w3, sa, sb, cw1, cw2, scratch, d_xs, d_bn, d_w3, d_b3,
                       d_cw, d_cb, nullptr, nullptr, dtype, stream);
}
// rows of bn_part (fp32 [rows][3][Fp]) tdeed_gsf_bwd_stats writes; 0: this geometry has no statistics epilogue
extern "C" int tdeed_gsf_bwd_bn_parts(int B, int T, int h, int w, int C, int Fp) {
  if (Fp % 8 != 0 || Fp > C || Fp / 8 > 64) return 0;
  const int PT = 256 / (Fp / 8);
  return cdiv(h * w, PT) * B * T;
}
// tdeed_gsf_bwd that also leaves the statistics of the module's BatchNorm3d backward (bn_mean: its batch means, fp32 [>= F]):
// bn_part[p][0][c] = sum d_bn, bn_part[p][1][c] = sum d_bn * (x - mean) over the pixels of workgroup p (row 2 unused), the
// layout tdeed_bn_bwd_from_parts folds (q = 1): no column-statistics pass over (d_bn, x)
extern "C" int tdeed_gsf_bwd_stats(const void* x, const float* gate, const float* fw, const float* ysum, const float* xsum,
                                   const void* dA, int B, int T, int h, int w, int C, int F, int Fp, const float* w3,
                                   const float* sa, const float* sb, const float* cw1, const float* cw2, float* scratch,
                                   void* d_xs, void* d_bn, float* d_w3, float* d_b3, float* d_cw, float* d_cb,
                                   const float* bn_mean, float* bn_part, int dtype, void* stream) {
  TD_CHECK(bn_mean && bn_part && tdeed_gsf_bwd_bn_parts(B, T, h, w, C, Fp) > 0, "gsf_bwd_stats: no statistics epilogue here");
  return gsf_bwd_entry(x, gate, fw, ysum, xsum, dA, B, T, h, w, C, F, Fp, w3, sa, sb, cw1, cw2, scratch, d_xs, d_bn, d_w3, d_b3,
                       d_cw, d_cb, bn_mean, bn_part, dtype, stream);
}

// dx[m][0:Fp] += a + b (dx row stride C): the module's input gradient joins conv1's pass-through gradient
extern "C" int tdeed_gsf_add_cols(const void* a, const void* b, long M, int C, int Fp, void* dx, int dtype, void* stream) {
  TD_CHECK(a && b && dx && M > 0 && Fp > 0 && Fp <= C, "gsf_add_cols: bad arguments");
  TD_CHECK(dtype == TDEED_F32 || dtype == TDEED_BF16, "gsf_add_cols: bad dtype %d", dtype);
  hipStream_t st = (hipStream_t)stream;
  const dim3 grid((unsigned)((M * Fp + 255) / 256));
  if (dtype == TDEED_F32) hipLaunchKernelGGL(gsf_add_cols_kernel<float>, grid, dim3(256), 0, st, (const float*)a, (const float*)b, M, C, Fp, (float*)dx);
  else hipLaunchKernelGGL(gsf_add_cols_kernel<bf16_t>, grid, dim3(256), 0, st, (const bf16_t*)a, (const bf16_t*)b, M, C, Fp, (bf16_t*)dx);
  TD_LAUNCH_CHECK("gsf_add_cols");
  return TDEED_OK;
}
