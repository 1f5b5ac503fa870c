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

template <typename T> __device__ __forceinline__ void load4(const T* p, float (&v)[4]);
template <> __device__ __forceinline__ void load4<float>(const float* p, float (&v)[4]) { Chunk<float>::load(p, v); }
template <> __device__ __forceinline__ void load4<bf16_t>(const bf16_t* p, float (&v)[4]) {
  const u32x2 u = *reinterpret_cast<const u32x2*>(p);
  v[0] = __uint_as_float(u[0] << 16); v[1] = __uint_as_float(u[0] & 0xffff0000u);
  v[2] = __uint_as_float(u[1] << 16); v[3] = __uint_as_float(u[1] & 0xffff0000u);
}

// ---- launch 1a: per-frame partial gate sums Q
// conv3d(a)[t] = sum_j conv2d(a[t+j-1], w[:,:,j]), so every frame is read ONCE: its block computes
// Q[f][p][j][g] = conv2d_3x3(relu(bn(x[f])), w3d[g][:, j]) for the three temporal taps j and both
// gate groups g.  The BN+ReLU'd frame band (+1 halo row each side) and the 27 x F weights sit in LDS;
// work items are (pixel, j, g), pixel fastest, so activation reads are conflict-free ds_read_b64
// (row stride F+2 floats) and weight reads are broadcasts.
// wq: [27][F] tap-major repack of conv3D.weight ([2][F/2][3][3][3]); channel c = g*F/2 + cl.
template <typename T>
__global__ __launch_bounds__(256) void gsf_q_kernel(const T* __restrict__ x, int h, int w, int C, int F,
                                                    int band, const float* __restrict__ bn_scale,
                                                    const float* __restrict__ bn_shift,
                                                    const float* __restrict__ wq, float* __restrict__ Q) {
  extern __shared__ float sm[];
  const int f = blockIdx.x;
  const int y0 = blockIdx.y * band;
  const int y1 = min(h, y0 + band);
  const int rows = y1 - y0 + 2;               // with halo rows y0-1 and y1
  const int LD = F + 2;
  float* wl = sm;                              // [27][F]
  float* a = sm + 27 * F;                      // [rows*w][LD]
  for (int i = threadIdx.x; i < 27 * F; i += 256) wl[i] = wq[i];
  const int nq = F >> 2;                       // channel quads (fold % 4 == 0)
  const int total = rows * w * nq;
  for (int i0 = threadIdx.x; i0 < total; i0 += 256 * 4) {     // 4 independent loads in flight per thread
    float v[4][4];
#pragma unroll
    for (int b4 = 0; b4 < 4; ++b4) {
      const int i = i0 + b4 * 256;
#pragma unroll
      for (int e = 0; e < 4; ++e) v[b4][e] = 0.f;
      if (i < total) {
        const int cq = i % nq;
        const int pix = i / nq;
        const int ry = pix / w, px = pix - ry * w;
        const int yy = y0 - 1 + ry;
        if (yy >= 0 && yy < h) load4<T>(x + ((long)f * h * w + (long)yy * w + px) * C + 4 * cq, v[b4]);
      }
    }
#pragma unroll
    for (int b4 = 0; b4 < 4; ++b4) {
      const int i = i0 + b4 * 256;
      if (i < total) {
        const int cq = i % nq;
        const int pix = i / nq;
        const int yy2 = y0 - 1 + pix / w;
        const bool outside = yy2 < 0 || yy2 >= h;               // halo rows beyond the image stay exactly zero
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float r = fmaxf(fmaf(v[b4][e], bn_scale[4 * cq + e], bn_shift[4 * cq + e]), 0.f);
          a[pix * LD + 4 * cq + e] = outside ? 0.f : r;
        }
      }
    }
  }
  __syncthreads();
  const int Fh = F >> 1;
  const int npix = (y1 - y0) * w;
  for (int it = threadIdx.x; it < 6 * npix; it += 256) {
    const int jg = it / npix;                 // 0..5 = j*2 + g
    const int p = it - jg * npix;
    const int j = jg >> 1, g = jg & 1;
    const int py = p / w, px = p - py * w;    // py relative to y0
    float acc = 0.f;
#pragma unroll
    for (int dy = 0; dy < 3; ++dy) {
#pragma unroll
      for (int dx = 0; dx < 3; ++dx) {
        const int xx = px + dx - 1;
        if (xx < 0 || xx >= w) continue;      // halo rows are zero-filled, columns are bounds-checked
        const float* ap = a + ((py + dy) * w + xx) * LD + g * Fh;
        const float* wp = wl + ((j * 3 + dy) * 3 + dx) * F + g * Fh;
        float a2 = 0.f;
#pragma unroll 4
        for (int c = 0; c < Fh; c += 2) {
          const f32x2 av = *reinterpret_cast<const f32x2*>(ap + c);
          const f32x2 wv = *reinterpret_cast<const f32x2*>(wp + c);
          acc = fmaf(av[0], wv[0], acc);
          a2 = fmaf(av[1], wv[1], a2);
        }
        acc += a2;
      }
    }
    Q[((long)f * h * w + (long)(y0 + py) * w + px) * 6 + jg] = acc;
  }
}

// ---- launch 1a, bf16 throughput mode: the same partial sums as an implicit GEMM on the MFMA pipe.
// D[jg][pixel] = sum_k Wt[jg][k] A[k][pixel], jg = (temporal tap j, gate g) = 6 of the 16 MFMA rows,
// k = (spatial tap, 8-channel chunk): a lane's 8 k-values are one 16-byte read of the BN+ReLU'd frame
// (bf16, zero halo, pixel stride an odd number of 16-byte slots => conflict-free ds_read_b128).
// wqf: [KS][64] fragments (engine.pack_gsf_q_frags), rows jg >= 6 and channels of the other gate group are 0.
__global__ __launch_bounds__(256) void gsf_q_mfma_kernel(const bf16_t* __restrict__ x, int h, int w, int C, int F,
                                                         int band, int nch, int PSQ, int KS,
                                                         const float* __restrict__ bn_scale,
                                                         const float* __restrict__ bn_shift,
                                                         const bf16x8* __restrict__ wqf, float* __restrict__ Q) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smq[];
  const int f = blockIdx.x;
  const int y0 = blockIdx.y * band;
  const int y1 = min(h, y0 + band);
  const int rows = y1 - y0 + 2, WP = w + 2;
  bf16x8* wl = reinterpret_cast<bf16x8*>(smq);                  // [KS][64]
  unsigned char* a = smq + (size_t)KS * 64 * 16;               // [rows][WP][PSQ]
  const int tid = threadIdx.x;
  for (int i = tid; i < KS * 64; i += 256) wl[i] = wqf[i];
  // stage: relu(bn(x)) as bf16, 8-channel chunks; halo ring, channels >= F and the stride pad are zero
  const int cpp = PSQ >> 4;                                     // 16-byte pieces per pixel incl. pad
  const int total = rows * WP * cpp;
  for (int i0 = tid; i0 < total; i0 += 256 * 4) {
    u32x4 v[4];
    int meta[4];
#pragma unroll
    for (int b4 = 0; b4 < 4; ++b4) {
      const int i = i0 + b4 * 256;
      meta[b4] = -1;
      v[b4] = (u32x4){0u, 0u, 0u, 0u};
      if (i < total) {
        const int j = i % cpp, pix = i / cpp;
        const int ry = pix / WP, rx = pix - ry * WP;
        const int yy = y0 - 1 + ry, xx = rx - 1;
        meta[b4] = -2 - j;                                      // zero piece
        if (j < nch && yy >= 0 && yy < h && xx >= 0 && xx < w) {
          const bf16_t* src = x + ((long)f * h * w + (long)yy * w + xx) * C + j * 8;
          if (j * 8 + 8 <= F) {
            v[b4] = *reinterpret_cast<const u32x4*>(src);
          } else {                                              // last chunk of a fold that is not a multiple of 8
            const u32x2 lo = *reinterpret_cast<const u32x2*>(src);
            v[b4] = (u32x4){lo[0], lo[1], 0u, 0u};
          }
          meta[b4] = j;
        }
      }
    }
#pragma unroll
    for (int b4 = 0; b4 < 4; ++b4) {
      const int i = i0 + b4 * 256;
      if (i >= total) continue;
      u32x4 o = {0u, 0u, 0u, 0u};
      if (meta[b4] >= 0) {
        const int c0 = meta[b4] * 8;
        float fv[8];
        Chunk<bf16_t>::load(reinterpret_cast<const bf16_t*>(&v[b4]), fv);
        bf16x8 r;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const int c = c0 + e;
          r[e] = (c < F) ? (bf16_t)fmaxf(fmaf(fv[e], bn_scale[c], bn_shift[c]), 0.f) : (bf16_t)0.f;
        }
        o = *reinterpret_cast<u32x4*>(&r);
      }
      *reinterpret_cast<u32x4*>(a + (long)i * 16) = o;
    }
  }
  __syncthreads();
  const int lane = tid & 63, wv = tid >> 6, pl = lane & 15, q = lane >> 4;
  const int npix = (y1 - y0) * w;
  const int ntl = (npix + 15) >> 4;
  for (int mt = wv; mt < ntl; mt += 4) {
    const int p = mt * 16 + pl;
    const bool pok = p < npix;
    const int pc = pok ? p : 0;
    const int py = pc / w, px = pc - py * w;
    const unsigned char* base = a + ((long)py * WP + px) * PSQ;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int ks = 0; ks < KS; ++ks) {
      const int s_ = 4 * ks + q;
      const int tap = s_ / nch, ck = s_ - tap * nch;
      const bool sok = tap < 9;
      const int dy = tap / 3, dx = tap - dy * 3;
      const bf16x8 af = *reinterpret_cast<const bf16x8*>(base + (sok ? ((dy * WP + dx) * PSQ + ck * 16) : 0));
      acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wl[ks * 64 + lane], af, acc, 0, 0, 0);
    }
    if (pok) {
      float* dst = Q + ((long)f * h * w + (long)(y0 + py) * w + px) * 6;
      if (q == 0) { dst[0] = acc[0]; dst[1] = acc[1]; dst[2] = acc[2]; dst[3] = acc[3]; }
      else if (q == 1) { dst[4] = acc[0]; dst[5] = acc[1]; }
    }
  }
}

// ---- launch 1b: gate = tanh(b + Q[t-1][0] + Q[t][1] + Q[t+1][2]) and the spatial sums of gate*x and x
template <typename T>
__global__ __launch_bounds__(256) void gsf_gate_sums_kernel(const T* __restrict__ x, const float* __restrict__ Q,
                                                            int T_len, int hw, int C, int F,
                                                            const float* __restrict__ b3d,
                                                            float* __restrict__ gate, float* __restrict__ ysum,
                                                            float* __restrict__ xsum) {
  extern __shared__ float sm[];        // gates [hw][2], then partial sums [2][S][F]
  const int f = blockIdx.x;
  const int t = f % T_len;
  const int Fh = F >> 1;
  float* sg = sm;
  float* part = sm + 2 * hw;
  for (int i = threadIdx.x; i < 2 * hw; i += 256) {
    const int p = i >> 1, g = i & 1;
    float v = b3d[g] + Q[((long)f * hw + p) * 6 + 2 + g];
    if (t > 0) v += Q[((long)(f - 1) * hw + p) * 6 + g];
    if (t < T_len - 1) v += Q[((long)(f + 1) * hw + p) * 6 + 4 + g];
    v = tanhf(v);
    sg[i] = v;
    gate[(long)f * hw * 2 + i] = v;
  }
  __syncthreads();
  // deterministic spatial sums: channel pairs across lanes (coalesced), S pixel slices, ordered reduce
  const int nq = F >> 1;
  const int S = 256 / nq;
  {
    const int cp = threadIdx.x % nq;
    const int s = threadIdx.x / nq;
    if (s < S) {
      float y0 = 0.f, y1 = 0.f, x0 = 0.f, x1 = 0.f;
      const T* xf = x + (long)f * hw * C + 2 * cp;
      const int g = (2 * cp) >= Fh;
      for (int p = s; p < hw; p += S) {
        float v0, v1;
        Pair<T>::load(xf + (long)p * C, v0, v1);
        const float gt = sg[2 * p + g];
        x0 += v0; x1 += v1;
        y0 += v0 * gt; y1 += v1 * gt;
      }
      part[s * F + 2 * cp] = y0;
      part[s * F + 2 * cp + 1] = y1;
      part[(S + s) * F + 2 * cp] = x0;
      part[(S + s) * F + 2 * cp + 1] = x1;
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
                                  const float* bn_scale, const float* bn_shift, const float* wq, const void* wqf,
                                  const float* b3d, float* Q, float* gate, float* ysum, float* xsum, int dtype,
                                  void* stream) {
  TD_CHECK(x && bn_scale && bn_shift && (wq || wqf) && b3d && Q && gate && ysum && xsum, "gsf_gate: null pointer");
  TD_CHECK(B > 0 && T > 0 && h > 0 && w > 0 && F > 0 && F % 4 == 0 && F <= C && F <= 256,
           "gsf_gate: bad sizes B=%d T=%d h=%d w=%d C=%d F=%d", B, T, h, w, C, F);
  TD_CHECK(dtype == TDEED_F32 || dtype == TDEED_BF16, "gsf_gate: bad dtype %d", dtype);
  const int hw = h * w;
  // rows per band so that weights + (band+2) rows of (F+2) floats fit 60 KB of LDS
  const long budget = 60 * 1024 / 4 - 27L * F;
  int band = (int)(budget / ((long)w * (F + 2))) - 2;
  TD_CHECK(band >= 1, "gsf_gate: row of %d px x %d ch does not fit LDS", w, F);
  if (band > h) band = h;
  const int nb = cdiv(h, band);
  TD_CHECK(nb <= 65535, "gsf_gate: too many bands");
  size_t smem1 = (size_t)(27 * F + (band + 2) * w * (F + 2)) * sizeof(float);
  const int S = 256 / (F / 2);
  size_t smem2 = (size_t)(2 * hw + 2 * S * F) * sizeof(float);
  TD_CHECK(smem2 <= 64 * 1024, "gsf_gate: frame too large for the gate/sum pass (%d px)", hw);
  hipStream_t st = (hipStream_t)stream;
  bool mfma_done = false;
  if (dtype == TDEED_BF16 && wqf) {
    const int nch = (F + 7) / 8;
    int ps16 = nch + 1;
    if ((ps16 & 1) == 0) ++ps16;
    const int PSQ = ps16 * 16, KSq = (9 * nch + 3) / 4;
    const long wbytes = (long)KSq * 64 * 16;
    int bq = (int)((60 * 1024 - wbytes) / ((long)(w + 2) * PSQ)) - 2;
    if (bq >= 1) {
      if (bq > h) bq = h;
      const int nbq = cdiv(h, bq);
      const size_t smq = (size_t)wbytes + (size_t)(bq + 2) * (w + 2) * PSQ;
      hipLaunchKernelGGL(gsf_q_mfma_kernel, dim3(B * T, nbq), dim3(256), smq, st, (const bf16_t*)x, h, w, C, F, bq,
                         nch, PSQ, KSq, bn_scale, bn_shift, (const bf16x8*)wqf, Q);
      mfma_done = true;
    }
  }
  if (!mfma_done) TD_CHECK(wq, "gsf_gate: fp32 tap weights needed for the VALU path");
  if (mfma_done) {
    hipLaunchKernelGGL(gsf_gate_sums_kernel<bf16_t>, dim3(B * T), dim3(256), smem2, st, (const bf16_t*)x, Q, T, hw,
                       C, F, b3d, gate, ysum, xsum);
  } else if (dtype == TDEED_F32) {
    hipLaunchKernelGGL(gsf_q_kernel<float>, dim3(B * T, nb), dim3(256), smem1, st, (const float*)x, h, w, C, F, band,
                       bn_scale, bn_shift, wq, Q);
    hipLaunchKernelGGL(gsf_gate_sums_kernel<float>, dim3(B * T), dim3(256), smem2, st, (const float*)x, Q, T, hw, C,
                       F, b3d, gate, ysum, xsum);
  } else {
    hipLaunchKernelGGL(gsf_q_kernel<bf16_t>, dim3(B * T, nb), dim3(256), smem1, st, (const bf16_t*)x, h, w, C, F,
                       band, bn_scale, bn_shift, wq, Q);
    hipLaunchKernelGGL(gsf_gate_sums_kernel<bf16_t>, dim3(B * T), dim3(256), smem2, st, (const bf16_t*)x, Q, T, hw,
                       C, F, b3d, gate, ysum, xsum);
  }
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
  for (int i = blockIdx.y * blockDim.x + threadIdx.x; i < F * T; i += gridDim.y * blockDim.x) {
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
  hipLaunchKernelGGL(gsf_weight_kernel, dim3(B, cdiv((long)F * T, 256)), dim3(256), 0, (hipStream_t)stream, ysum, xsum, T, F,
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

// frame-per-block variant that also evaluates the fusion weights of its frame (the (channel,time)-plane
// conv of launch 2) in its prologue: one launch less per site, no fw round trip through memory.
template <typename T>
__global__ __launch_bounds__(256) void gsf_apply_fused_kernel(const T* __restrict__ x, const float* __restrict__ gate,
                                                              const float* __restrict__ ysum,
                                                              const float* __restrict__ xsum, float inv_hw,
                                                              const float* __restrict__ cw1,
                                                              const float* __restrict__ cb1,
                                                              const float* __restrict__ cw2,
                                                              const float* __restrict__ cb2, int T_len, int hw,
                                                              int C, int F, int Fp, T* __restrict__ out) {
  extern __shared__ float fwl[];                  // [F] fusion weight of this frame, indexed by source channel
  const long f = blockIdx.x;
  const int t = (int)(f % T_len);
  const long b = f / T_len;
  const int Fh = F >> 1, Fq = F >> 2;
  for (int c = threadIdx.x; c < F; c += 256) {
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
        if (t2 < 0 || t2 >= T_len) continue;
        const long row = (b * T_len + t2) * F + cc;
        const float rm = (xsum[row] - ysum[row]) * inv_hw;
        const int ts = g ? t2 - 1 : t2 + 1;
        const float ysh = (ts >= 0 && ts < T_len) ? ysum[(b * T_len + ts) * F + cc] * inv_hw : 0.f;
        a = fmaf(cw[(dc + 1) * 3 + (dt + 1)], ysh, a);
        a = fmaf(cw[9 + (dc + 1) * 3 + (dt + 1)], rm, a);
      }
    }
    fwl[c] = sigmoidf_(a);
  }
  __syncthreads();
  const int qpr = Fp >> 2;
  for (int idx = threadIdx.x; idx < hw * qpr; idx += 256) {
    const int q = idx % qpr;
    const long pix = f * hw + idx / qpr;
    const T* xp = x + pix * C;
    float o[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int co = q * 4 + e;
      if (co >= F) { o[e] = (float)xp[co]; continue; }
      const int g = co >= Fh;
      const int col = co - g * Fh;
      const int j = col >> 1, i = col & 1;
      const int ci = g * Fh + i * Fq + j;
      const float gt = gate[pix * 2 + g];
      const float xv = (float)xp[ci];
      const float r = xv - gt * xv;
      const int ts = g ? t - 1 : t + 1;
      float ysh = 0.f;
      if (ts >= 0 && ts < T_len) {
        const long pix2 = pix + (long)(ts - t) * hw;
        ysh = gate[pix2 * 2 + g] * (float)x[pix2 * C + ci];
      }
      const float wv = fwl[ci];
      o[e] = ysh * wv + r * (1.0f - wv);
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

extern "C" int tdeed_gsf_apply_fused_fwd(const void* x, const float* gate, const float* ysum, const float* xsum,
                                         const float* cw1, const float* cb1, const float* cw2, const float* cb2,
                                         int B, int T, int h, int w, int C, int F, int Fp, void* out, int dtype,
                                         void* stream) {
  TD_CHECK(x && gate && ysum && xsum && cw1 && cb1 && cw2 && cb2 && out, "gsf_apply_fused: null pointer");
  TD_CHECK(F % 4 == 0 && Fp % 8 == 0 && Fp >= F && Fp <= C, "gsf_apply_fused: bad fold F=%d Fp=%d C=%d", F, Fp, C);
  TD_CHECK(B > 0 && T > 0 && (long)B * T <= 0x7fffffffL, "gsf_apply_fused: bad sizes");
  const int hw = h * w;
  hipStream_t st = (hipStream_t)stream;
  const size_t smem = (size_t)F * sizeof(float);
  if (dtype == TDEED_F32)
    hipLaunchKernelGGL(gsf_apply_fused_kernel<float>, dim3(B * T), dim3(256), smem, st, (const float*)x, gate, ysum,
                       xsum, 1.0f / (float)hw, cw1, cb1, cw2, cb2, T, hw, C, F, Fp, (float*)out);
  else if (dtype == TDEED_BF16)
    hipLaunchKernelGGL(gsf_apply_fused_kernel<bf16_t>, dim3(B * T), dim3(256), smem, st, (const bf16_t*)x, gate, ysum,
                       xsum, 1.0f / (float)hw, cw1, cb1, cw2, cb2, T, hw, C, F, Fp, (bf16_t*)out);
  else { tdeed_set_error("gsf_apply_fused: bad dtype %d", dtype); return TDEED_ERR_ARG; }
  TD_LAUNCH_CHECK("gsf_apply_fused");
  return TDEED_OK;
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
