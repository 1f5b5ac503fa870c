// Gate-Shift-Fuse / Gate-Shift module in eval mode, channels-last.
// Reference: /root/reference/model/impl/gsf.py:38-93, gsm.py:89-116 (restated, not translated):
//   gate  = tanh(conv3d_{3x3x3, groups 2}(relu(bn3d(x))))                  (B,2,T,h,w)
//   y = gate_g * x_g, r = x_g - y;  y shifted by one frame (group 1 left, group 2 right, zero fill)
//   GSF: fw = sigmoid(conv2d_{2->1,3x3}([mean_hw y_shift ; mean_hw r]) over the (channel,time) plane)
//        out = y_shift*fw + r*(1-fw);      GSM: out = y_shift + r
//   channel interleave inside each half: c = i*(F/4)+j -> 2j+i.
// Three launches: (1) gates + spatial sums, one block per frame, temporal halo t-1,t,t+1 read
// straight from L2 (a 3-frame x fold slab is <= 110 KB, L2 resident); (2) the tiny (c,t)-plane conv;
// (3) blend + shift + interleave, written as the first Fp columns of conv1's A operand.
#include "common.h"

template <typename T> struct Pair;   // 2 consecutive elements (F/2 is always even)
template <> struct Pair<float> {
  static __device__ __forceinline__ void load(const float* p, float& a, float& b) {
    f32x2 t = *reinterpret_cast<const f32x2*>(p);
    a = t[0]; b = t[1];
  }
};
template <> struct Pair<bf16_t> {
  static __device__ __forceinline__ void load(const bf16_t* p, float& a, float& b) {
    unsigned int u = *reinterpret_cast<const unsigned int*>(p);
    a = __uint_as_float(u << 16);
    b = __uint_as_float(u & 0xffff0000u);
  }
};

// wq: [27][F] tap-major repack of conv3D.weight ([2][F/2][3][3][3]); channel c = g*F/2 + cl.
template <typename T>
__global__ __launch_bounds__(256) void gsf_gate_kernel(const T* __restrict__ x, int T_len, int h, int w, int C,
                                                       int F, const float* __restrict__ bn_scale,
                                                       const float* __restrict__ bn_shift,
                                                       const float* __restrict__ wq,
                                                       const float* __restrict__ b3d, float* __restrict__ gate,
                                                       float* __restrict__ ysum, float* __restrict__ xsum) {
  extern __shared__ float sm[];        // gates [hw][2], then partial sums [2][S][F]
  const int f = blockIdx.x;
  const int t = f % T_len;
  const int hw = h * w;
  const int Fh = F >> 1;
  float* sg = sm;
  float* part = sm + 2 * hw;
  for (int p = threadIdx.x; p < hw; p += 256) {
    const int py = p / w, px = p - py * w;
    float g0 = b3d[0], g1 = b3d[1];
    for (int dt = 0; dt < 3; ++dt) {
      const int tt = t + dt - 1;
      if (tt < 0 || tt >= T_len) continue;
      const T* xf = x + (long)(f + dt - 1) * hw * C;
      for (int dy = 0; dy < 3; ++dy) {
        const int yy = py + dy - 1;
        if (yy < 0 || yy >= h) continue;
        for (int dx = 0; dx < 3; ++dx) {
          const int xx = px + dx - 1;
          if (xx < 0 || xx >= w) continue;
          const T* src = xf + ((long)yy * w + xx) * C;
          const float* wt = wq + ((dt * 3 + dy) * 3 + dx) * F;
          float a0 = 0.f, a1 = 0.f;
          for (int c = 0; c < Fh; c += 2) {
            float v0, v1;
            Pair<T>::load(src + c, v0, v1);
            a0 = fmaf(fmaxf(fmaf(v0, bn_scale[c], bn_shift[c]), 0.f), wt[c], a0);
            a0 = fmaf(fmaxf(fmaf(v1, bn_scale[c + 1], bn_shift[c + 1]), 0.f), wt[c + 1], a0);
          }
          for (int c = Fh; c < F; c += 2) {
            float v0, v1;
            Pair<T>::load(src + c, v0, v1);
            a1 = fmaf(fmaxf(fmaf(v0, bn_scale[c], bn_shift[c]), 0.f), wt[c], a1);
            a1 = fmaf(fmaxf(fmaf(v1, bn_scale[c + 1], bn_shift[c + 1]), 0.f), wt[c + 1], a1);
          }
          g0 += a0;
          g1 += a1;
        }
      }
    }
    g0 = tanhf(g0);
    g1 = tanhf(g1);
    sg[2 * p] = g0;
    sg[2 * p + 1] = g1;
    gate[((long)f * hw + p) * 2] = g0;
    gate[((long)f * hw + p) * 2 + 1] = g1;
  }
  __syncthreads();
  // spatial sums of y = gate*x and of x, deterministic: S pixel slices per channel, ordered reduce
  const int S = 256 / F;               // F <= 256 checked on the host
  {
    const int c = threadIdx.x % F;
    const int s = threadIdx.x / F;
    if (s < S) {
      float ys = 0.f, xs = 0.f;
      const T* xf = x + (long)f * hw * C + c;
      const int g = c >= Fh;
      for (int p = s; p < hw; p += S) {
        float v = (float)xf[(long)p * C];
        xs += v;
        ys += v * sg[2 * p + g];
      }
      part[s * F + c] = ys;
      part[(S + s) * F + c] = xs;
    }
  }
  __syncthreads();
  for (int c = threadIdx.x; c < F; c += 256) {
    float ys = 0.f, xs = 0.f;
    for (int s = 0; s < S; ++s) {
      ys += part[s * F + c];
      xs += part[(S + s) * F + c];
    }
    ysum[(long)f * F + c] = ys;
    xsum[(long)f * F + c] = xs;
  }
}

extern "C" int tdeed_gsf_gate_fwd(const void* x, int B, int T, int h, int w, int C, int F,
                                  const float* bn_scale, const float* bn_shift, const float* wq,
                                  const float* b3d, float* gate, float* ysum, float* xsum, int dtype,
                                  void* stream) {
  TD_CHECK(x && bn_scale && bn_shift && wq && b3d && gate && ysum && xsum, "gsf_gate: null pointer");
  TD_CHECK(B > 0 && T > 0 && h > 0 && w > 0 && F > 0 && F % 4 == 0 && F <= C && F <= 256,
           "gsf_gate: bad sizes B=%d T=%d h=%d w=%d C=%d F=%d", B, T, h, w, C, F);
  const int hw = h * w;
  const int S = 256 / F;
  size_t smem = (size_t)(2 * hw + 2 * S * F) * sizeof(float);
  TD_CHECK(smem <= 64 * 1024, "gsf_gate: frame too large for LDS (%d px)", hw);
  hipStream_t st = (hipStream_t)stream;
  if (dtype == TDEED_F32)
    hipLaunchKernelGGL(gsf_gate_kernel<float>, dim3(B * T), dim3(256), smem, st, (const float*)x, T, h, w, C, F,
                       bn_scale, bn_shift, wq, b3d, gate, ysum, xsum);
  else if (dtype == TDEED_BF16)
    hipLaunchKernelGGL(gsf_gate_kernel<bf16_t>, dim3(B * T), dim3(256), smem, st, (const bf16_t*)x, T, h, w, C, F,
                       bn_scale, bn_shift, wq, b3d, gate, ysum, xsum);
  else { tdeed_set_error("gsf_gate: bad dtype %d", dtype); return TDEED_ERR_ARG; }
  TD_LAUNCH_CHECK("gsf_gate");
  return TDEED_OK;
}

// --------------------------------------------------------------------------- fusion weights
// cw: conv weight [2][3][3] (in-channel 0 = shifted-y mean, 1 = r mean; kernel over (channel, time)).
__global__ void gsf_weight_kernel(const float* __restrict__ ysum, const float* __restrict__ xsum, int T, int F,
                                  float inv_hw, const float* __restrict__ cw1, const float* __restrict__ cb1,
                                  const float* __restrict__ cw2, const float* __restrict__ cb2,
                                  float* __restrict__ fw) {
  const int b = blockIdx.x;
  const int Fh = F >> 1;
  for (int i = threadIdx.x; i < F * T; i += blockDim.x) {
    const int c = i / T, t = i - c * T;
    const int g = c >= Fh;
    const int cl = c - g * Fh;
    const float* cw = g ? cw2 : cw1;
    float a = g ? cb2[0] : cb1[0];
#pragma unroll
    for (int dc = -1; dc <= 1; ++dc) {
      const int c2 = cl + dc;
      if (c2 < 0 || c2 >= Fh) continue;
      const int cc = g * Fh + c2;
#pragma unroll
      for (int dt = -1; dt <= 1; ++dt) {
        const int t2 = t + dt;
        if (t2 < 0 || t2 >= T) continue;
        const long row = (long)(b * T + t2) * F + cc;
        const float ym = ysum[row] * inv_hw;
        const float rm = (xsum[row] - ysum[row]) * inv_hw;
        // shifted y: group 1 reads frame t2+1, group 2 frame t2-1, zero outside the clip
        const int ts = g ? t2 - 1 : t2 + 1;
        const float ysh = (ts >= 0 && ts < T) ? ysum[(long)(b * T + ts) * F + cc] * inv_hw : 0.f;
        (void)ym;
        a = fmaf(cw[(dc + 1) * 3 + (dt + 1)], ysh, a);
        a = fmaf(cw[9 + (dc + 1) * 3 + (dt + 1)], rm, a);
      }
    }
    fw[((long)b * F + c) * T + t] = sigmoidf_(a);
  }
}

extern "C" int tdeed_gsf_weight_fwd(const float* ysum, const float* xsum, int B, int T, int F, int hw,
                                    const float* cw1, const float* cb1, const float* cw2, const float* cb2,
                                    float* fw, void* stream) {
  TD_CHECK(ysum && xsum && cw1 && cb1 && cw2 && cb2 && fw, "gsf_weight: null pointer");
  TD_CHECK(B > 0 && T > 0 && F > 0 && hw > 0, "gsf_weight: bad sizes");
  hipLaunchKernelGGL(gsf_weight_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, ysum, xsum, T, F,
                     1.0f / (float)hw, cw1, cb1, cw2, cb2, fw);
  TD_LAUNCH_CHECK("gsf_weight");
  return TDEED_OK;
}

// --------------------------------------------------------------------------- blend + shift + interleave
template <typename T>
__global__ __launch_bounds__(256) void gsf_apply_kernel(const T* __restrict__ x, const float* __restrict__ gate,
                                                        const float* __restrict__ fw, int T_len, int hw, int C,
                                                        int F, int Fp, T* __restrict__ out, long total) {
  const int Fh = F >> 1, Fq = F >> 2;
  const int qpr = Fp >> 2;             // 4-channel groups per pixel
  for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
    const int q = (int)(idx % qpr);
    const long pix = idx / qpr;        // global pixel index = f*hw + p
    const long f = pix / hw;
    const int t = (int)(f % T_len);
    const long b = f / T_len;
    const T* xp = x + pix * C;
    float o[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int co = q * 4 + e;        // output channel
      if (co >= F) {
        o[e] = (float)xp[co];          // pass-through padding columns [F, Fp)
        continue;
      }
      const int g = co >= Fh;
      const int col = co - g * Fh;     // = 2*j + i
      const int j = col >> 1, i = col & 1;
      const int ci = g * Fh + i * Fq + j;   // source channel
      const float gt = gate[pix * 2 + g];
      const float xv = (float)xp[ci];
      const float r = xv - gt * xv;
      const int ts = g ? t - 1 : t + 1;
      float ysh = 0.f;
      if (ts >= 0 && ts < T_len) {
        const long pix2 = pix + (long)(ts - t) * hw;
        ysh = gate[pix2 * 2 + g] * (float)x[pix2 * C + ci];
      }
      if (fw) {
        const float wv = fw[(b * F + ci) * T_len + t];
        o[e] = ysh * wv + r * (1.0f - wv);
      } else {
        o[e] = ysh + r;
      }
    }
    T* dst = out + pix * Fp + q * 4;
    if constexpr (sizeof(T) == 4) {
      Chunk<float>::store(reinterpret_cast<float*>(dst), o);
    } else {
      bf16x4 v;
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = (bf16_t)o[e];
      *reinterpret_cast<bf16x4*>(dst) = v;
    }
  }
}

extern "C" int tdeed_gsf_apply_fwd(const void* x, const float* gate, const float* fw, int B, int T, int h, int w,
                                   int C, int F, int Fp, void* out, int dtype, void* stream) {
  TD_CHECK(x && gate && out, "gsf_apply: null pointer");
  TD_CHECK(F % 4 == 0 && Fp % 8 == 0 && Fp >= F && Fp <= C, "gsf_apply: bad fold F=%d Fp=%d C=%d", F, Fp, C);
  const int hw = h * w;
  const long total = (long)B * T * hw * (Fp / 4);
  const int grid = (int)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192);
  hipStream_t st = (hipStream_t)stream;
  if (dtype == TDEED_F32)
    hipLaunchKernelGGL(gsf_apply_kernel<float>, dim3(grid), dim3(256), 0, st, (const float*)x, gate, fw, T, hw, C, F,
                       Fp, (float*)out, total);
  else if (dtype == TDEED_BF16)
    hipLaunchKernelGGL(gsf_apply_kernel<bf16_t>, dim3(grid), dim3(256), 0, st, (const bf16_t*)x, gate, fw, T, hw, C,
                       F, Fp, (bf16_t*)out, total);
  else { tdeed_set_error("gsf_apply: bad dtype %d", dtype); return TDEED_ERR_ARG; }
  TD_LAUNCH_CHECK("gsf_apply");
  return TDEED_OK;
}
