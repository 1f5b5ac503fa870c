// Train-time augmentation of `TDEEDModel.Impl.forward` (/root/reference/model/model.py:76-83, 154-157): per clip, with the
// clip's own random parameters, ColorJitter(hue) -> ColorJitter(saturation) -> ColorJitter(brightness) ->
// ColorJitter(contrast) -> GaussianBlur(5) on the already cropped 0..1 frames (RandomHorizontalFlip, the last transform of
// the Compose, is a per-frame flag of the stem's load: tdeed_stem_fwd).  The arithmetic restates torchvision 0.18.1's
// float-tensor functional ops (torchvision is a pip dependency of the reference, requirements.txt:41, not vendored):
//   hue:        rgb -> hsv, h = (h + f) mod 1, hsv -> rgb                      (_rgb2hsv / _hsv2rgb)
//   saturation: clamp(f * x + (1 - f) * gray(x), 0, 1), gray = 0.2989 r + 0.587 g + 0.114 b
//   brightness: clamp(f * x, 0, 1)
//   contrast:   clamp(f * x + (1 - f) * mean_frame(gray(x)), 0, 1)              (mean over ONE frame)
//   blur:       5x5 separable gaussian, sigma per clip, reflect padding
// Per-clip parameters prm[b][8] = {hue shift, saturation, brightness, contrast, blur sigma (0 = off), 3 unused}; identity
// values (0, 1, 1, 1, 0) switch a stage off exactly (no arithmetic is applied for it).
// HBM-bound streaming kernels: input 1 B/px (uint8) or 4 B/px (mixup batches), output fp32 0..255 frames of the crop
// window that tdeed_stem_fwd(frames_f32 = 1) reads.
#include "common.h"

namespace {

struct ClipPrm { float hue, sat, bri, con, sigma; };

__device__ __forceinline__ ClipPrm load_prm(const float* prm, int b) {
  const float* p = prm + (long)b * 8;
  return ClipPrm{p[0], p[1], p[2], p[3], p[4]};
}

__device__ __forceinline__ float clamp01(float v) { return fminf(fmaxf(v, 0.f), 1.f); }
__device__ __forceinline__ float gray_of(float r, float g, float b) { return 0.2989f * r + 0.587f * g + 0.114f * b; }

// torchvision _rgb2hsv + hue shift + _hsv2rgb on one pixel
__device__ __forceinline__ void hue_shift(float& r, float& g, float& b, float f) {
  const float maxc = fmaxf(r, fmaxf(g, b)), minc = fminf(r, fminf(g, b));
  const bool eqc = maxc == minc;
  const float cr = maxc - minc;
  const float s = cr / (eqc ? 1.f : maxc);
  const float crd = eqc ? 1.f : cr;
  const float rc = (maxc - r) / crd, gc = (maxc - g) / crd, bc = (maxc - b) / crd;
  float h;
  if (maxc == r) h = bc - gc;
  else if (maxc == g) h = 2.0f + rc - bc;
  else h = 4.0f + gc - rc;
  h = fmodf(h / 6.0f + 1.0f, 1.0f);
  h = h + f;
  h = h - floorf(h);                                             // python's % 1.0 (result in [0, 1))
  const float v = maxc;
  const float i_f = floorf(h * 6.0f);
  const float fr = h * 6.0f - i_f;
  int i = (int)i_f % 6;
  const float p = clamp01(v * (1.0f - s));
  const float q = clamp01(v * (1.0f - s * fr));
  const float t = clamp01(v * (1.0f - s * (1.0f - fr)));
  switch (i) {
    case 0: r = v; g = t; b = p; break;
    case 1: r = q; g = v; b = p; break;
    case 2: r = p; g = v; b = t; break;
    case 3: r = p; g = q; b = v; break;
    case 4: r = t; g = p; b = v; break;
    default: r = v; g = p; b = q; break;
  }
}

// hue, saturation, brightness of one 0..1 pixel (the stages in front of the contrast mean)
__device__ __forceinline__ void color_head(float& r, float& g, float& b, const ClipPrm& c) {
  if (c.hue != 0.f) hue_shift(r, g, b, c.hue);
  if (c.sat != 1.f) {
    const float gr = gray_of(r, g, b);
    r = clamp01(c.sat * r + (1.f - c.sat) * gr);
    g = clamp01(c.sat * g + (1.f - c.sat) * gr);
    b = clamp01(c.sat * b + (1.f - c.sat) * gr);
  }
  if (c.bri != 1.f) {
    r = clamp01(c.bri * r);
    g = clamp01(c.bri * g);
    b = clamp01(c.bri * b);
  }
}

template <typename IN>
__device__ __forceinline__ void load_px(const IN* src, long plane, long off, float& r, float& g, float& b) {
  r = (float)src[off] / 255.f;
  g = (float)src[plane + off] / 255.f;
  b = (float)src[2 * plane + off] / 255.f;
}

constexpr int AUG_SLABS = 32;      // partial sums per frame for the contrast mean

// pass 1 (only frames of clips with contrast != 1 do work): partial sums of gray(color_head(x)) over the crop window
template <typename IN>
__global__ __launch_bounds__(256) void aug_mean_kernel(const IN* __restrict__ frames, int T, int H, int W, int top, int left,
                                                       int ch, int cw, const float* __restrict__ prm,
                                                       float* __restrict__ part) {
  __shared__ float scratch[8];
  const int n = blockIdx.x, slab = blockIdx.y;
  const ClipPrm c = load_prm(prm, n / T);
  if (c.con == 1.f) return;
  const IN* src = frames + (long)n * 3 * H * W;
  const long plane = (long)H * W;
  const int npx = ch * cw;
  const int per = (npx + AUG_SLABS - 1) / AUG_SLABS;
  const int p0 = slab * per, p1 = min(npx, p0 + per);
  const IDiv dw(cw);
  float a = 0.f;
  for (int p = p0 + threadIdx.x; p < p1; p += 256) {
    int y, x;
    dw.divmod(p, y, x);
    float r, g, b;
    load_px(src, plane, (long)(top + y) * W + left + x, r, g, b);
    color_head(r, g, b, c);
    a += gray_of(r, g, b);
  }
  a = block_sum<4>(a, scratch);
  if (threadIdx.x == 0) part[(long)n * AUG_SLABS + slab] = a;
}

// pass 2: all colour stages; writes fp32 0..255 crop-window frames into out (clips without blur) or tmp (clips with blur)
template <typename IN>
__global__ __launch_bounds__(256) void aug_color_kernel(const IN* __restrict__ frames, int T, int H, int W, int top, int left,
                                                        int ch, int cw, const float* __restrict__ prm,
                                                        const float* __restrict__ part, float* __restrict__ out,
                                                        float* __restrict__ tmp) {
  const int n = blockIdx.y;
  const ClipPrm c = load_prm(prm, n / T);
  const IN* src = frames + (long)n * 3 * H * W;
  const long plane = (long)H * W, oplane = (long)ch * cw;
  float* dst = (c.sigma > 0.f ? tmp : out) + (long)n * 3 * oplane;
  float mean = 0.f;
  if (c.con != 1.f) {
    for (int i = 0; i < AUG_SLABS; ++i) mean += part[(long)n * AUG_SLABS + i];   // fixed order: deterministic
    mean /= (float)oplane;
  }
  const IDiv dw(cw);
  for (int p = blockIdx.x * 256 + threadIdx.x; p < (int)oplane; p += gridDim.x * 256) {
    int y, x;
    dw.divmod(p, y, x);
    float r, g, b;
    load_px(src, plane, (long)(top + y) * W + left + x, r, g, b);
    color_head(r, g, b, c);
    if (c.con != 1.f) {
      r = clamp01(c.con * r + (1.f - c.con) * mean);
      g = clamp01(c.con * g + (1.f - c.con) * mean);
      b = clamp01(c.con * b + (1.f - c.con) * mean);
    }
    dst[p] = r * 255.f;
    dst[oplane + p] = g * 255.f;
    dst[2 * oplane + p] = b * 255.f;
  }
}

// pass 3 (clips with sigma > 0): 5x5 gaussian, reflect padding, tmp -> out.  One 32x32 output tile per block, 36x36
// input patch in LDS, separable: rows first into a second LDS image, then columns.
__global__ __launch_bounds__(256) void aug_blur_kernel(const float* __restrict__ tmp, int T, int ch, int cw,
                                                       const float* __restrict__ prm, float* __restrict__ out) {
  __shared__ float patch[36][37];
  __shared__ float rowp[36][33];
  const int n = blockIdx.z;
  const float sigma = prm[(long)(n / T) * 8 + 4];
  if (!(sigma > 0.f)) return;
  float k[5];
  {
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 5; ++i) {
      const float x = (float)(i - 2) / sigma;
      k[i] = expf(-0.5f * x * x);
      s += k[i];
    }
#pragma unroll
    for (int i = 0; i < 5; ++i) k[i] /= s;
  }
  const int oy0 = blockIdx.y * 32, ox0 = blockIdx.x * 32;
  const long oplane = (long)ch * cw;
  for (int c = 0; c < 3; ++c) {
    const float* src = tmp + ((long)n * 3 + c) * oplane;
    __syncthreads();
    for (int i = threadIdx.x; i < 36 * 36; i += 256) {
      const int py = i / 36, px = i - py * 36;
      int y = oy0 + py - 2, x = ox0 + px - 2;
      y = y < 0 ? -y : (y >= ch ? 2 * ch - 2 - y : y);            // reflect (no edge repeat), torch 'reflect' padding
      x = x < 0 ? -x : (x >= cw ? 2 * cw - 2 - x : x);
      y = min(max(y, 0), ch - 1);
      x = min(max(x, 0), cw - 1);
      patch[py][px] = src[(long)y * cw + x];
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 36 * 32; i += 256) {
      const int py = i >> 5, px = i & 31;
      float a = 0.f;
#pragma unroll
      for (int j = 0; j < 5; ++j) a = fmaf(k[j], patch[py][px + j], a);
      rowp[py][px] = a;
    }
    __syncthreads();
    float* dst = out + ((long)n * 3 + c) * oplane;
    for (int i = threadIdx.x; i < 32 * 32; i += 256) {
      const int py = i >> 5, px = i & 31;
      const int y = oy0 + py, x = ox0 + px;
      if (y < ch && x < cw) {
        float a = 0.f;
#pragma unroll
        for (int j = 0; j < 5; ++j) a = fmaf(k[j], rowp[py + j][px], a);
        dst[(long)y * cw + x] = a;
      }
    }
  }
}

}  // namespace

extern "C" long tdeed_augment_scratch_floats(int N) { return (long)N * AUG_SLABS; }

// frames: uint8 (or fp32 0..255 when frames_f32) [B*T][3][H][W]; prm fp32 [B][8] on the device; out / tmp fp32
// [B*T][3][crop_h][crop_w] (tmp only read/written for clips with sigma > 0; may alias nothing else); part fp32
// [tdeed_augment_scratch_floats(B*T)].
extern "C" int tdeed_augment_clips(const void* frames, int frames_f32, int B, int T, int H, int W, int crop_top, int crop_left,
                                   int crop_h, int crop_w, const float* prm, float* part, float* out, float* tmp,
                                   void* stream) {
  TD_CHECK(frames && prm && part && out && tmp, "augment: null pointer");
  TD_CHECK(B > 0 && T > 0 && crop_h >= 3 && crop_w >= 3 && crop_top >= 0 && crop_left >= 0 && crop_top + crop_h <= H &&
               crop_left + crop_w <= W, "augment: bad geometry");
  const int N = B * T;
  TD_CHECK(N <= 65535, "augment: at most 65535 frames per launch (got %d)", N);
  hipStream_t st = (hipStream_t)stream;
  const int gx = min(cdiv((long)crop_h * crop_w, 256), 64);
  if (frames_f32) {
    hipLaunchKernelGGL(aug_mean_kernel<float>, dim3(N, AUG_SLABS), dim3(256), 0, st, (const float*)frames, T, H, W, crop_top,
                       crop_left, crop_h, crop_w, prm, part);
    hipLaunchKernelGGL(aug_color_kernel<float>, dim3(gx, N), dim3(256), 0, st, (const float*)frames, T, H, W, crop_top,
                       crop_left, crop_h, crop_w, prm, part, out, tmp);
  } else {
    hipLaunchKernelGGL(aug_mean_kernel<uint8_t>, dim3(N, AUG_SLABS), dim3(256), 0, st, (const uint8_t*)frames, T, H, W,
                       crop_top, crop_left, crop_h, crop_w, prm, part);
    hipLaunchKernelGGL(aug_color_kernel<uint8_t>, dim3(gx, N), dim3(256), 0, st, (const uint8_t*)frames, T, H, W, crop_top,
                       crop_left, crop_h, crop_w, prm, part, out, tmp);
  }
  hipLaunchKernelGGL(aug_blur_kernel, dim3(cdiv(crop_w, 32), cdiv(crop_h, 32), N), dim3(256), 0, st, tmp, T, crop_h, crop_w,
                     prm, out);
  TD_LAUNCH_CHECK("augment");
  return TDEED_OK;
}
