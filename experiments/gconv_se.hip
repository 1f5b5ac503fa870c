// Parked negative result (round 1): conv2 + SE in one launch for maps <= 14x14.  Faster back to back (31 vs 37 us at 7x7),
// slower inside the overlapped two-stream graph; not built into libtdeed_hip.so.  Needs common.h and the staging helpers of
// t-deed_amd/csrc/conv.hip.
#include "common.h"


// =========================================================================== grouped 3x3 + SE in one launch
// Small maps (s3: 14x14x152, s4: 7x7x368 of RegNetY-200MF): ONE workgroup owns a whole frame, so the SE squeeze
// completes inside it and the excitation runs in the same launch.  The conv output never touches LDS or HBM
// unscaled: each wave keeps its (unit, pixel-tile) accumulators in registers until the gate is known, then stores
// y2 * gate.  conv3 downstream is then a plain contraction (no per-frame operand re-scale) and the separate SE
// launch disappears.  bf16, stride 1; weights: MFMA fragments of pack_gconv_frags, SE weights of pack_se_bf16.
template <int NUW, int NPT>     // units per wave, pixel tiles per frame
__global__ __launch_bounds__(256) void gconv_se_kernel(const bf16_t* __restrict__ x, int h, int w, int C,
                                                       const bf16x8* __restrict__ wfrag,
                                                       const float* __restrict__ scale,
                                                       const float* __restrict__ shift,
                                                       const bf16_t* __restrict__ se_w1p,
                                                       const float* __restrict__ se_b1,
                                                       const bf16_t* __restrict__ se_w2p,
                                                       const float* __restrict__ se_b2, int R,
                                                       bf16_t* __restrict__ y) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int hw = h * w, WP = w + 2;
  const int PSF = C * 2 + 16;                                   // pixel stride: all channels + 16 B skew
  unsigned char* tile = smem;                                   // [(h+2)][(w+2)][PSF], zero halo / pad
  float* pooled = reinterpret_cast<float*>(tile + (size_t)(h + 2) * WP * PSF);   // [C]
  const int R8 = (R + 7) & ~7;
  float* hid = pooled + C;                                      // [R8]
  float* gate = hid + R8;                                       // [C]
  float* part = gate + C;                                       // SE partial sums
  const int n = blockIdx.x;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, pl = lane & 15, q = lane >> 4;
  // ---- stage the frame with a zero ring (and zero pad piece behind the channels)
  {
    const int cpp = PSF >> 4;                                   // 16-byte pieces per pixel incl. the pad piece
    const int cch = C >> 3;
    const bf16_t* xin = x + (long)n * hw * C;
    const int total = (h + 2) * WP * cpp;
    for (int i0 = tid; i0 < total; i0 += 256 * 8) {
      u32x4 v[8];
#pragma unroll
      for (int b8 = 0; b8 < 8; ++b8) {
        const int i = i0 + b8 * 256;
        v[b8] = (u32x4){0u, 0u, 0u, 0u};
        if (i < total) {
          const int j = i % cpp, pix = i / cpp;
          const int r = pix / WP, xx = pix - r * WP;
          const int iy = r - 1, ix = xx - 1;
          if (j < cch && iy >= 0 && iy < h && ix >= 0 && ix < w)
            v[b8] = *reinterpret_cast<const u32x4*>(xin + ((long)iy * w + ix) * C + j * 8);
        }
      }
#pragma unroll
      for (int b8 = 0; b8 < 8; ++b8) {
        const int i = i0 + b8 * 256;
        if (i < total) *reinterpret_cast<u32x4*>(tile + (long)(i / cpp) * PSF + (i % cpp) * 16) = v[b8];
      }
    }
  }
  __syncthreads();
  // ---- conv2: wave wv owns units wv, wv+4, ...; accumulators stay in registers
  const int NU = (C + 15) >> 4;
  f32x4 acc[NUW][NPT];
  int poff[NPT];
  bool pok[NPT];
#pragma unroll
  for (int pt = 0; pt < NPT; ++pt) {
    const int p = pt * 16 + pl;
    pok[pt] = p < hw;
    const int pc = pok[pt] ? p : 0;
    const int oy = pc / w, ox = pc - oy * w;
    poff[pt] = (oy * WP + ox) * PSF;
  }
  int toff[5];
#pragma unroll
  for (int ks = 0; ks < 5; ++ks) {
    const int sidx = 4 * ks + q;
    const int half = sidx / 9, tap = sidx - half * 9;
    const int dy = tap / 3, dx = tap - dy * 3;
    toff[ks] = sidx < 18 ? (dy * WP + dx) * PSF + half * 16 : 0;
  }
#pragma unroll
  for (int ui = 0; ui < NUW; ++ui) {
    const int U = wv + 4 * ui;
    if (U < NU) {
      bf16x8 wf[5];
#pragma unroll
      for (int ks = 0; ks < 5; ++ks) wf[ks] = wfrag[((long)U * 5 + ks) * 64 + lane];
      const int ch0 = U * 16 + 4 * q;
      float sc[4], sh[4], psum[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const bool ok = ch0 + r < C;
        sc[r] = ok ? scale[ch0 + r] : 0.f;
        sh[r] = ok ? shift[ch0 + r] : 0.f;
        psum[r] = 0.f;
      }
#pragma unroll
      for (int pt = 0; pt < NPT; ++pt) {
        f32x4 a = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < 5; ++ks) {
          const bf16x8 yf = *reinterpret_cast<const bf16x8*>(tile + poff[pt] + toff[ks] + U * 32);
          a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[ks], yf, a, 0, 0, 0);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float vv = (float)(bf16_t)fmaxf(a[r] * sc[r] + sh[r], 0.f);   // the squeeze sees the bf16 activation
          a[r] = vv;
          if (pok[pt]) psum[r] += vv;
        }
        acc[ui][pt] = a;
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float v = psum[r];
        v += __shfl_xor(v, 1, 64);
        v += __shfl_xor(v, 2, 64);
        v += __shfl_xor(v, 4, 64);
        v += __shfl_xor(v, 8, 64);
        if (pl == 0 && ch0 + r < C) pooled[ch0 + r] = v;
      }
    }
  }
  __syncthreads();
  // ---- SE excitation (bf16 weights, one batch of wide loads per phase)
  {
    const float inv = 1.0f / (float)hw;
    constexpr int MAXB = 24;
    {
      const int NJ = R8 >> 3, nsl = 256 / NJ;
      const int jo = tid % NJ, sl = tid / NJ;
      if (sl < nsl) {
        float a[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        const int cper = (C + nsl - 1) / nsl;
        const int c0 = sl * cper, c1 = min(C, c0 + cper);
        for (int cb = c0; cb < c1; cb += MAXB) {
          bf16x8 wv8[MAXB];
#pragma unroll
          for (int i = 0; i < MAXB; ++i)
            if (cb + i < c1) wv8[i] = *reinterpret_cast<const bf16x8*>(se_w1p + (long)(cb + i) * R8 + jo * 8);
#pragma unroll
          for (int i = 0; i < MAXB; ++i)
            if (cb + i < c1) {
              const float pv = pooled[cb + i];
#pragma unroll
              for (int e = 0; e < 8; ++e) a[e] = fmaf(pv, (float)wv8[i][e], a[e]);
            }
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) part[sl * R8 + jo * 8 + e] = a[e];
      }
      __syncthreads();
      if (tid < R8) {
        float v = 0.f;
        for (int s_ = 0; s_ < nsl; ++s_) v += part[s_ * R8 + tid];
        hid[tid] = tid < R ? fmaxf(v * inv + se_b1[tid], 0.f) : 0.f;
      }
      __syncthreads();
    }
    {
      const int NC = C >> 3;
      const int nsl = 256 / NC > 0 ? 256 / NC : 1;
      const int co = tid % NC, sl = tid / NC;
      const bool act = sl < nsl && tid < NC * nsl;
      if (act) {
        float a[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        const int jper = (R + nsl - 1) / nsl;
        const int j0 = sl * jper, j1 = min(R, j0 + jper);
        for (int jb = j0; jb < j1; jb += MAXB) {
          bf16x8 wv8[MAXB];
#pragma unroll
          for (int i = 0; i < MAXB; ++i)
            if (jb + i < j1) wv8[i] = *reinterpret_cast<const bf16x8*>(se_w2p + (long)(jb + i) * C + co * 8);
#pragma unroll
          for (int i = 0; i < MAXB; ++i)
            if (jb + i < j1) {
              const float hv = hid[jb + i];
#pragma unroll
              for (int e = 0; e < 8; ++e) a[e] = fmaf(hv, (float)wv8[i][e], a[e]);
            }
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) part[sl * C + co * 8 + e] = a[e];
      }
      __syncthreads();
      for (int c = tid; c < C; c += 256) {
        float g = 0.f;
        for (int s_ = 0; s_ < nsl; ++s_) g += part[s_ * C + c];
        gate[c] = sigmoidf_(g + se_b2[c]);
      }
      __syncthreads();
    }
  }
  // ---- store y2 * gate straight from the accumulators
  bf16_t* yout = y + (long)n * hw * C;
#pragma unroll
  for (int ui = 0; ui < NUW; ++ui) {
    const int U = wv + 4 * ui;
    const int ch0 = U * 16 + 4 * q;
    if (U < NU && ch0 < C) {
      float g4[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) g4[r] = gate[ch0 + r];
#pragma unroll
      for (int pt = 0; pt < NPT; ++pt) {
        if (!pok[pt]) continue;
        bf16x4 o;
#pragma unroll
        for (int r = 0; r < 4; ++r) o[r] = (bf16_t)(acc[ui][pt][r] * g4[r]);
        *reinterpret_cast<bf16x4*>(yout + (long)(pt * 16 + pl) * C + ch0) = o;
      }
    }
  }
}

static size_t gconv_se_smem(int h, int w, int C, int R) {
  const int R8 = (R + 7) & ~7;
  const size_t p1 = (size_t)(256 / (R8 / 8)) * R8, p2 = (size_t)(256 / (C / 8) > 0 ? 256 / (C / 8) : 1) * C;
  return (size_t)(h + 2) * (w + 2) * (C * 2 + 16) + ((size_t)2 * C + R8 + (p1 > p2 ? p1 : p2)) * sizeof(float);
}

// 0 = unsupported, else a variant id: 1 = (6 units/wave, 4 pixel tiles): C<=384, hw<=64; 2 = (3, 13): C<=192, hw<=208
extern "C" int tdeed_gconv_se_fits(int h, int w, int C, int R) {
  const int hw = h * w, NU = (C + 15) / 16;
  if (C % 8 != 0 || R > 2048 || gconv_se_smem(h, w, C, R) > 120 * 1024) return 0;
  if (hw <= 64 && NU <= 24) return 1;
  if (hw <= 208 && NU <= 12) return 2;
  return 0;
}

extern "C" int tdeed_gconv_se_fwd(const void* x, int N, int h, int w, int C, const void* wfrag, const float* scale,
                                  const float* shift, const void* se_w1p, const float* se_b1, const void* se_w2p,
                                  const float* se_b2, int R, void* y, void* stream) {
  TD_CHECK(x && wfrag && scale && shift && se_w1p && se_b1 && se_w2p && se_b2 && y, "gconv_se: null pointer");
  const int var = tdeed_gconv_se_fits(h, w, C, R);
  TD_CHECK(N > 0 && var != 0, "gconv_se: geometry h=%d w=%d C=%d R=%d unsupported", h, w, C, R);
  const size_t smem = gconv_se_smem(h, w, C, R);
  hipStream_t st = (hipStream_t)stream;
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute((const void*)gconv_se_kernel<6, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, 120 * 1024);
    if (e == hipSuccess) e = hipFuncSetAttribute((const void*)gconv_se_kernel<3, 13>, hipFuncAttributeMaxDynamicSharedMemorySize, 120 * 1024);
    if (e != hipSuccess) { tdeed_set_error("gconv_se: hipFuncSetAttribute: %s", hipGetErrorString(e)); return TDEED_ERR_RUNTIME; }
    attr_set = true;
  }
  if (var == 1)
    hipLaunchKernelGGL((gconv_se_kernel<6, 4>), dim3(N), dim3(256), smem, st, (const bf16_t*)x, h, w, C,
                       (const bf16x8*)wfrag, scale, shift, (const bf16_t*)se_w1p, se_b1, (const bf16_t*)se_w2p, se_b2, R,
                       (bf16_t*)y);
  else
    hipLaunchKernelGGL((gconv_se_kernel<3, 13>), dim3(N), dim3(256), smem, st, (const bf16_t*)x, h, w, C,
                       (const bf16x8*)wfrag, scale, shift, (const bf16_t*)se_w1p, se_b1, (const bf16_t*)se_w2p, se_b2, R,
                       (bf16_t*)y);
  TD_LAUNCH_CHECK("gconv_se");
  return TDEED_OK;
}
