// Fused front of the RegNetY trunk (bf16 throughput mode):
//   uint8 frame -> /255, crop, flip, standardise -> stem conv3x3 s2 (3->32)+BN+ReLU
//               -> s1.b1.conv1 1x1 (32->C1)+BN+ReLU -> s1.b1.conv2 grouped 3x3 s2 (+BN+ReLU, SE squeeze)
//               -> s1.b1.downsample 1x1 s2 (32->C1)+BN        (the block's shortcut)
// One block = one frame x one band of conv2 output rows.  The 112^2 x 32 stem map and the 112^2 x C1 conv1
// map (0.64 + 0.48 GB per 8 clips in bf16, written once and read 2.25x by the unfused path) never leave
// the CU: the normalised input patch and the conv1 band live in LDS, the stem result goes from MFMA
// accumulators straight into the next MFMA as its B operand (k order permuted consistently in the
// pre-packed weights).  HBM traffic per step drops from 2.77 GB to 0.36 GB for these four layers.
//
// MFMA formulation (v_mfma_f32_16x16x32_bf16, weights = A operand, 16 pixels = B operand columns):
//   stem : K = (ky, kx 0..3, c 0..3) = 48 of 64: the LDS patch is [row][col][4 bf16] (RGB0), so a lane's 8
//          k-values (two horizontally adjacent taps) are ONE aligned 16-byte LDS read. 2 k-steps x 2 n-tiles.
//   conv1/downsample : K = 32 stem channels, taken from the stem accumulators (rows 4q+r of both n-tiles).
//   conv2: as gconv3x3_mfma_kernel (conv.hip): 5 k-steps per 16-channel unit from the LDS conv1 band.
#include "common.h"
#include <stdlib.h>

// ReLU on packed bf16: a negative value has its sign bit set, i.e. is negative as a signed 16-bit integer, so one
// v_pk_max_i16 against 0 clamps two elements (the kernel is bound by vector-ALU issue: this halves its ReLU instructions)
typedef short short4v __attribute__((ext_vector_type(4)));
typedef short short8v __attribute__((ext_vector_type(8)));
__device__ __forceinline__ bf16x8 relu_bf16x8(bf16x8 v) {
  short8v s = __builtin_bit_cast(short8v, v);
  s = __builtin_elementwise_max(s, (short8v)(0));
  return __builtin_bit_cast(bf16x8, s);
}
__device__ __forceinline__ bf16x4 relu_bf16x4(bf16x4 v) {
  short4v s = __builtin_bit_cast(short4v, v);
  s = __builtin_elementwise_max(s, (short4v)(0));
  return __builtin_bit_cast(bf16x4, s);
}

struct FrontP {
  const uint8_t* frames; int H, W, top, left, ch, cw, flip;
  const bf16x8* stem_wf;  const float* stem_sc; const float* stem_sh;     // [2][2][64]
  const bf16x8* w1f;      const float* sc1;     const float* sh1;         // [C1P/16][64]
  const bf16x8* wdf;      const float* scd;     const float* shd;
  const bf16x8* w2f;      const float* sc2;     const float* sh2;         // [ceil4(C1/16)][5][64]
  bf16_t* y2; bf16_t* shortcut; float* pooled;
  int C1, CSP, PS, band, nbands, Hs, Ws, Ho, Wo;     // stem map Hs x Ws, block output Ho x Wo
  int vec16;                                          // rows are 16-byte aligned: vector uint8 loads
};

template <int NT1>
__global__ __launch_bounds__(256) void s1_front_kernel(const FrontP p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  __shared__ float red[4][16];
  // bands of one frame share halo rows: keep them on one XCD (one L2)
  const long lid = xcd_logical_id(blockIdx.x, gridDim.x);
  int bnd, n;
  td_split(lid, p.nbands, n, bnd);
  const int oy0 = bnd * p.band;
  const int nrows_out = min(p.band, p.Ho - oy0);
  const int y1r0 = 2 * oy0 - 1;                         // first conv1-map row held (may be -1)
  const int ny1 = 2 * (nrows_out - 1) + 3;              // conv1 rows held
  const int in_r0 = 2 * y1r0 - 1;                       // first input row held
  const int nin = 2 * (ny1 - 1) + 3;
  const int INW = p.cw + 2;                             // patch columns: input col -1 .. cw (zero padded)
  const int Y1W = p.Ws + 2;                             // conv1 columns -1 .. Ws
  bf16_t* inp = reinterpret_cast<bf16_t*>(smem);                       // [nin][INW][4]
  const int in_bytes = ((2 * (2 * (p.band - 1) + 3 - 1) + 3) * INW * 8 + 15) & ~15;
  unsigned char* y1t = smem + in_bytes;                                 // [ny1][Y1W][PS]
  // the wave index as a scalar: tile loops and their index arithmetic then run on the scalar unit (the kernel is bound by
  // vector-ALU issue)
  const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int px = lane & 15, q = lane >> 4;

  // ---- A. fill the patch: normalised input as bf16 [row][col][RGB0]; zero halo columns / rows outside the
  //         image; zero the parts of the conv1 band that phase B does not write (halo columns, rows outside the map)
  {
    // (u/255 - mean)/std as one fp32 FMA; the result is rounded to bf16 anyway
    const float na[3] = {1.0f / (255.0f * 0.229f), 1.0f / (255.0f * 0.224f), 1.0f / (255.0f * 0.225f)};
    const float nb[3] = {-0.485f / 0.229f, -0.456f / 0.224f, -0.406f / 0.225f};
    const uint8_t* src = p.frames + (long)n * 3 * p.H * p.W;
    const long plane = (long)p.H * p.W;
    const bf16x4 z4 = {(bf16_t)0.f, (bf16_t)0.f, (bf16_t)0.f, (bf16_t)0.f};
    if (p.vec16) {
      // a thread owns 16 pixels of one row: three 16-byte loads (R,G,B planes) in flight, 16 8-byte LDS stores
      const int nch = p.cw >> 4;
      const int total = nin * nch;
      const IDiv dnch(nch);
      for (int i = tid; i < total; i += 256) {
        int k, r;
        dnch.divmod(i, r, k);
        const int iy = in_r0 + r;
        bf16x4* dst = reinterpret_cast<bf16x4*>(inp + ((long)r * INW + 16 * k + 1) * 4);
        if (iy >= 0 && iy < p.ch) {
          const int scol = p.flip ? (p.cw - 16 - 16 * k) : 16 * k;
          const uint8_t* s0 = src + (long)(p.top + iy) * p.W + p.left + scol;
          const u32x4 v0 = *reinterpret_cast<const u32x4*>(s0);
          const u32x4 v1 = *reinterpret_cast<const u32x4*>(s0 + plane);
          const u32x4 v2 = *reinterpret_cast<const u32x4*>(s0 + 2 * plane);
#pragma unroll
          for (int e = 0; e < 16; ++e) {
            const unsigned int sh8 = 8 * (e & 3);
            bf16x4 o;
            o[0] = (bf16_t)fmaf((float)((v0[e >> 2] >> sh8) & 0xffu), na[0], nb[0]);
            o[1] = (bf16_t)fmaf((float)((v1[e >> 2] >> sh8) & 0xffu), na[1], nb[1]);
            o[2] = (bf16_t)fmaf((float)((v2[e >> 2] >> sh8) & 0xffu), na[2], nb[2]);
            o[3] = (bf16_t)0.f;
            dst[p.flip ? (15 - e) : e] = o;
          }
        } else {
#pragma unroll
          for (int e = 0; e < 16; ++e) dst[e] = z4;
        }
      }
    } else {
      for (int i = tid; i < nin * p.cw; i += 256) {
        const int r = i / p.cw, ix = i - r * p.cw;
        const int iy = in_r0 + r;
        bf16x4 v4 = z4;
        if (iy >= 0 && iy < p.ch) {
          const int sx = p.flip ? (p.cw - 1 - ix) : ix;
          const long o = (long)(p.top + iy) * p.W + (p.left + sx);
#pragma unroll
          for (int c3 = 0; c3 < 3; ++c3) v4[c3] = (bf16_t)fmaf((float)src[c3 * plane + o], na[c3], nb[c3]);
        }
        *reinterpret_cast<bf16x4*>(inp + ((long)r * INW + ix + 1) * 4) = v4;
      }
    }
    for (int i = tid; i < nin * 2; i += 256)          // patch halo columns (input col -1 and cw)
      *reinterpret_cast<bf16x4*>(inp + ((long)(i >> 1) * INW + ((i & 1) ? (p.cw + 1) : 0)) * 4) = z4;
    // conv1 band: halo columns of every row, and whole rows that fall outside the stem map
    const int cpp = p.PS >> 4;                         // 16-byte pieces per pixel (incl. the pad piece)
    for (int i = tid; i < ny1 * 2 * cpp; i += 256) {
      const int j = i % cpp, rc = i / cpp;
      const int rr = rc >> 1, col = (rc & 1) ? (p.Ws + 1) : 0;
      *reinterpret_cast<u32x4*>(y1t + ((long)rr * Y1W + col) * p.PS + j * 16) = (u32x4){0u, 0u, 0u, 0u};
    }
    for (int rr = 0; rr < ny1; ++rr) {
      const int r = y1r0 + rr;
      if (r >= 0 && r < p.Hs) continue;
      for (int i = tid; i < Y1W * cpp; i += 256)
        *reinterpret_cast<u32x4*>(y1t + (long)rr * Y1W * p.PS + (long)i * 16) = (u32x4){0u, 0u, 0u, 0u};
    }
  }
  __syncthreads();

  // ---- B. stem -> conv1 (-> LDS band) and downsample (-> HBM), 16 stem pixels per MFMA tile
  {
    bf16x8 swf[2][2];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) swf[t][ks] = p.stem_wf[(t * 2 + ks) * 64 + lane];
    float ssc[8], ssh[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int c = (e < 4) ? (4 * q + e) : (16 + 4 * q + e - 4);
      ssc[e] = p.stem_sc[c];
      ssh[e] = p.stem_sh[c];
    }
    // conv1 / downsample weights and their BN affine for this lane's channels stay in registers
    bf16x8 w1r[NT1], wdr[NT1];
    float c1s[NT1][4], c1h[NT1][4], cds[NT1][4], cdh[NT1][4];
#pragma unroll
    for (int t = 0; t < NT1; ++t) {
      w1r[t] = p.w1f[t * 64 + lane];
      wdr[t] = p.wdf[t * 64 + lane];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int chn = t * 16 + 4 * q + e;
        const bool ok = chn < p.C1;
        c1s[t][e] = ok ? p.sc1[chn] : 0.f; c1h[t][e] = ok ? p.sh1[chn] : 0.f;
        cds[t][e] = ok ? p.scd[chn] : 0.f; cdh[t][e] = ok ? p.shd[chn] : 0.f;
      }
    }
    // stem rows to produce: conv1 rows y1r0 .. y1r0+ny1-1 that lie inside the map
    const int r_lo = max(y1r0, 0), r_hi = min(y1r0 + ny1, p.Hs);
    const int tiles_per_row = (p.Ws + 15) >> 4;
    const int ntiles = (r_hi - r_lo) * tiles_per_row;
    const IDiv dtpr(tiles_per_row);
    for (int tI = wv; tI < ntiles; tI += 4) {
      int rq, rm;
      dtpr.divmod(tI, rq, rm);
      const int r = r_lo + rq;
      const int c0 = rm * 16;
      const int c = c0 + px;
      const bool cok = c < p.Ws;
      const int cc = cok ? c : (p.Ws - 1);
      // stem B fragments: slot s = 4ks+q -> (ky = s>>1, half = s&1): 16 B at patch[(2r-1+ky) - in_r0][2c + 2half]
      f32x4 sa[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        const int s = 4 * ks + q;
        const int ky = s >> 1, half = s & 1;
        const int prow = (s < 6) ? (2 * r - 1 + ky - in_r0) : 0;
        const int pcol = (s < 6) ? (2 * cc + 2 * half) : 0;
        const bf16x8 xf = *reinterpret_cast<const bf16x8*>(inp + (prow * INW + pcol) * 4);       // LDS offsets: 32-bit
        sa[0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(swf[0][ks], xf, sa[0], 0, 0, 0);
        sa[1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(swf[1][ks], xf, sa[1], 0, 0, 0);
      }
      // BN + ReLU -> bf16: this lane's 8 stem channels of pixel (r, c) = k-slot q of the next contraction
      bf16x8 sf;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        sf[e] = (bf16_t)(sa[0][e] * ssc[e] + ssh[e]);
        sf[4 + e] = (bf16_t)(sa[1][e] * ssc[4 + e] + ssh[4 + e]);
      }
      sf = relu_bf16x8(sf);
      unsigned char* y1p = y1t + ((r - y1r0) * Y1W + (cc + 1)) * p.PS;
      const bool do_ds = ((r & 1) == 0) && (r >> 1) >= oy0 && (r >> 1) < oy0 + nrows_out;
#pragma unroll
      for (int t = 0; t < NT1; ++t) {
        const int ch0 = t * 16 + 4 * q;
        f32x4 a1 = {0.f, 0.f, 0.f, 0.f};
        a1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w1r[t], sf, a1, 0, 0, 0);
        if (cok) {          // channels >= C1 get exact zeros (their scale/shift are 0): the band needs no pre-clear
          bf16x4 o;
#pragma unroll
          for (int e = 0; e < 4; ++e) o[e] = (bf16_t)(a1[e] * c1s[t][e] + c1h[t][e]);
          *reinterpret_cast<bf16x4*>(y1p + ch0 * 2) = relu_bf16x4(o);
        }
        if (do_ds) {
          f32x4 ad = {0.f, 0.f, 0.f, 0.f};
          ad = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wdr[t], sf, ad, 0, 0, 0);
          if (cok && (c & 1) == 0 && ch0 < p.C1) {
            bf16x4 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = (bf16_t)(ad[e] * cds[t][e] + cdh[t][e]);
            *reinterpret_cast<bf16x4*>(p.shortcut + (((long)n * p.Ho + (r >> 1)) * p.Wo + (c >> 1)) * p.C1 + ch0) = o;
          }
        }
      }
    }
  }
  __syncthreads();

  // ---- C. grouped 3x3 stride 2 from the LDS conv1 band (same scheme as gconv3x3_mfma_kernel)
  {
    const int units = p.CSP >> 4;
    const int unit = wv % units;
    const int mstep = 4 / units;
    bf16x8 wf[5];
#pragma unroll
    for (int ks = 0; ks < 5; ++ks) wf[ks] = p.w2f[(unit * 5 + ks) * 64 + lane];
    int off[5];
#pragma unroll
    for (int ks = 0; ks < 5; ++ks) {
      const int sidx = 4 * ks + q;
      const int half = sidx / 9, tap = sidx - half * 9;
      const int dy = tap / 3, dx = tap - dy * 3;
      off[ks] = sidx < 18 ? (dy * Y1W + dx) * p.PS + half * 16 + unit * 32 : unit * 32;
    }
    const int ch0 = unit * 16 + q * 4;
    float sc[4], sh[4], psum[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const bool ok = ch0 + r < p.C1;
      sc[r] = ok ? p.sc2[ch0 + r] : 0.f;
      sh[r] = ok ? p.sh2[ch0 + r] : 0.f;
      psum[r] = 0.f;
    }
    const int npix = nrows_out * p.Wo;
    const int ntl = (npix + 15) >> 4;
    bf16_t* yout = p.y2 + ((long)n * p.Ho + oy0) * p.Wo * p.C1;
    const IDiv dwo(p.Wo);
    for (int mt = wv / units; mt < ntl; mt += mstep) {
      const int pp = mt * 16 + px;
      const bool pok = pp < npix;
      const int pc = pok ? pp : 0;
      int oyl, ox;
      dwo.divmod(pc, oyl, ox);
      const unsigned char* base = y1t + ((oyl * 2) * Y1W + ox * 2) * p.PS;
      f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < 5; ++ks) {
        const bf16x8 xf = *reinterpret_cast<const bf16x8*>(base + off[ks]);
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[ks], xf, acc, 0, 0, 0);
      }
      if (pok && ch0 < p.C1) {
        bf16x4 o;
#pragma unroll
        for (int r = 0; r < 4; ++r) o[r] = (bf16_t)(acc[r] * sc[r] + sh[r]);
        o = relu_bf16x4(o);
#pragma unroll
        for (int r = 0; r < 4; ++r) psum[r] += (float)o[r];
        *reinterpret_cast<bf16x4*>(yout + (long)pc * p.C1 + ch0) = o;
      }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      float v = psum[r];
      v += __shfl_xor(v, 1, 64);
      v += __shfl_xor(v, 2, 64);
      v += __shfl_xor(v, 4, 64);
      v += __shfl_xor(v, 8, 64);
      if (px == 0) red[wv][q * 4 + r] = v;
    }
    __syncthreads();
    if (tid < units * 16) {
      const int u = tid >> 4, cc = tid & 15;
      float sres = 0.f;
      for (int w2 = u; w2 < 4; w2 += units) sres += red[w2][cc];
      const int chn = u * 16 + cc;
      if (chn < p.C1) p.pooled[((long)n * p.nbands + bnd) * p.C1 + chn] = sres;
    }
  }
}

#define FRONT_LDS_CAP (80 * 1024)

// ---------------------------------------------------------------------------------------------------------------------
// Rolling form of the same launch.  s1_front_kernel above gives every conv2 output row band its own workgroup, which with
// one-row bands (what 48 KB of LDS = three workgroups per CU allows) recomputes half of the conv1 rows and three of every
// seven input rows.  Here a workgroup walks a STRIP of S consecutive output rows: the conv1 rows live in a ring of three
// LDS rows and the normalised input rows in a ring of eight, so a step adds four input rows and two conv1 rows to what the
// previous step left (S = 8: 35 input / 17 conv1 rows per strip instead of 56 / 24).  Same arithmetic, same order per
// output element; the squeeze sums come as one partial row per strip.
template <int NT1>
__global__ __launch_bounds__(256) void s1_front_roll_kernel(const FrontP p, int S) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  __shared__ float red[4][16];
  __shared__ __attribute__((aligned(16))) bf16_t zero16[8];           // operand of the two unused k-slots of the stem patch
  const long lid = xcd_logical_id(blockIdx.x, gridDim.x);
  const int nstrips = (p.Ho + S - 1) / S;
  int bnd, n;
  td_split(lid, nstrips, n, bnd);
  const int oy0 = bnd * S;
  const int nrows_out = min(S, p.Ho - oy0);
  const int INW = p.cw + 2, Y1W = p.Ws + 2;
  bf16_t* inp = reinterpret_cast<bf16_t*>(smem);                       // [8][INW][4] ring of normalised input rows
  const int in_bytes = (8 * INW * 8 + 15) & ~15;
  unsigned char* y1t = smem + in_bytes;                                 // [3][Y1W][PS] ring of conv1 rows
  const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int px = lane & 15, q = lane >> 4;
  const int cpp = p.PS >> 4;
  // ---- once: halo columns of both rings
  {
    const bf16x4 z4 = {(bf16_t)0.f, (bf16_t)0.f, (bf16_t)0.f, (bf16_t)0.f};
    if (tid < 8) zero16[tid] = (bf16_t)0.f;
    for (int i = tid; i < 16; i += 256)
      *reinterpret_cast<bf16x4*>(inp + ((long)(i >> 1) * INW + ((i & 1) ? (p.cw + 1) : 0)) * 4) = z4;
    for (int i = tid; i < 3 * 2 * cpp; i += 256) {
      const int j = i % cpp, rc = i / cpp;
      const int rr = rc >> 1, col = (rc & 1) ? (p.Ws + 1) : 0;
      *reinterpret_cast<u32x4*>(y1t + ((long)rr * Y1W + col) * p.PS + j * 16) = (u32x4){0u, 0u, 0u, 0u};
    }
  }
  // ---- weights and BatchNorm affines (as in s1_front_kernel)
  bf16x8 swf[2][2];
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) swf[t][ks] = p.stem_wf[(t * 2 + ks) * 64 + lane];
  float ssc[8], ssh[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const int c = (e < 4) ? (4 * q + e) : (16 + 4 * q + e - 4);
    ssc[e] = p.stem_sc[c];
    ssh[e] = p.stem_sh[c];
  }
  bf16x8 w1r[NT1], wdr[NT1];
  float c1s[NT1][4], c1h[NT1][4], cds[NT1][4], cdh[NT1][4];
#pragma unroll
  for (int t = 0; t < NT1; ++t) {
    w1r[t] = p.w1f[t * 64 + lane];
    wdr[t] = p.wdf[t * 64 + lane];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int chn = t * 16 + 4 * q + e;
      const bool ok = chn < p.C1;
      c1s[t][e] = ok ? p.sc1[chn] : 0.f; c1h[t][e] = ok ? p.sh1[chn] : 0.f;
      cds[t][e] = ok ? p.scd[chn] : 0.f; cdh[t][e] = ok ? p.shd[chn] : 0.f;
    }
  }
  const int units = p.CSP >> 4;
  const int unit = wv % units;
  const int mstep = 4 / units;
  bf16x8 wf[5];
#pragma unroll
  for (int ks = 0; ks < 5; ++ks) wf[ks] = p.w2f[(unit * 5 + ks) * 64 + lane];
  int tdy[5], toff[5];                      // per k-slot: tap row (0..2) and the offset of the tap inside a conv1 row
#pragma unroll
  for (int ks = 0; ks < 5; ++ks) {
    const int sidx = 4 * ks + q;
    const int half = sidx / 9, tap = sidx - half * 9;
    const int dy = tap / 3, dx = tap - dy * 3;
    tdy[ks] = sidx < 18 ? dy : 0;
    toff[ks] = sidx < 18 ? dx * p.PS + half * 16 + unit * 32 : unit * 32;
  }
  const int ch0 = unit * 16 + q * 4;
  float sc2[4], sh2[4], psum[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const bool ok = ch0 + r < p.C1;
    sc2[r] = ok ? p.sc2[ch0 + r] : 0.f;
    sh2[r] = ok ? p.sh2[ch0 + r] : 0.f;
    psum[r] = 0.f;
  }
  const float na[3] = {1.0f / (255.0f * 0.229f), 1.0f / (255.0f * 0.224f), 1.0f / (255.0f * 0.225f)};
  const float nb[3] = {-0.485f / 0.229f, -0.456f / 0.224f, -0.406f / 0.225f};
  const uint8_t* src = p.frames + (long)n * 3 * p.H * p.W;
  const long plane = (long)p.H * p.W;
  const bf16x4 z4 = {(bf16_t)0.f, (bf16_t)0.f, (bf16_t)0.f, (bf16_t)0.f};
  const int tiles_per_row = (p.Ws + 15) >> 4;
  const IDiv dtpr(tiles_per_row);
  const int nch = p.cw >> 4;
  const IDiv dnch(max(nch, 1)), dcw(p.cw);
  __syncthreads();

  for (int k = 0; k < nrows_out; ++k) {
    const int oy = oy0 + k;
    // ---- A. new input rows -> ring (slot = (row + 8) & 7); rows outside the crop are zeros
    {
      const int ir0 = k == 0 ? 4 * oy - 3 : 4 * oy;
      const int nir = k == 0 ? 7 : 4;
      if (p.vec16) {
        // four pixels per lane: a step's 4 rows x (cw / 4) quads spread over all four waves (16 pixels per lane left the
        // conversion to one wave: 354 -> 332 us).  Requesting the next step's quads one step ahead as the pipelined kernel
        // does was measured too: 325 us and 16 more VGPRs, no change at the 800MF widths; not kept.
        const int nq4 = p.cw >> 2;
        const IDiv dq4(nq4);
        for (int i = tid; i < nir * nq4; i += 256) {
          int kq, r;
          dq4.divmod(i, r, kq);
          const int iy = ir0 + r;
          bf16x4* dst = reinterpret_cast<bf16x4*>(inp + ((long)((iy + 8) & 7) * INW + 4 * kq + 1) * 4);
          if (iy >= 0 && iy < p.ch) {
            const int scol = p.flip ? (p.cw - 4 - 4 * kq) : 4 * kq;
            const uint8_t* s0 = src + (long)(p.top + iy) * p.W + p.left + scol;
            const unsigned int v0 = *reinterpret_cast<const unsigned int*>(s0);
            const unsigned int v1 = *reinterpret_cast<const unsigned int*>(s0 + plane);
            const unsigned int v2 = *reinterpret_cast<const unsigned int*>(s0 + 2 * plane);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              const unsigned int sh8 = 8 * e;
              bf16x4 o;
              o[0] = (bf16_t)fmaf((float)((v0 >> sh8) & 0xffu), na[0], nb[0]);
              o[1] = (bf16_t)fmaf((float)((v1 >> sh8) & 0xffu), na[1], nb[1]);
              o[2] = (bf16_t)fmaf((float)((v2 >> sh8) & 0xffu), na[2], nb[2]);
              o[3] = (bf16_t)0.f;
              dst[p.flip ? (3 - e) : e] = o;
            }
          } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) dst[e] = z4;
          }
        }
      } else {
        for (int i = tid; i < nir * p.cw; i += 256) {
          int r, ix;
          dcw.divmod(i, r, ix);
          const int iy = ir0 + r;
          bf16x4 v4 = z4;
          if (iy >= 0 && iy < p.ch) {
            const int sx = p.flip ? (p.cw - 1 - ix) : ix;
            const long o = (long)(p.top + iy) * p.W + (p.left + sx);
#pragma unroll
            for (int c3 = 0; c3 < 3; ++c3) v4[c3] = (bf16_t)fmaf((float)src[c3 * plane + o], na[c3], nb[c3]);
          }
          *reinterpret_cast<bf16x4*>(inp + ((long)((iy + 8) & 7) * INW + ix + 1) * 4) = v4;
        }
      }
    }
    __syncthreads();
    // ---- B. new conv1 rows (stem -> conv1 -> ring slot (r + 3) % 3) and the downsample of the even one
    {
      const int r_first = k == 0 ? 2 * oy - 1 : 2 * oy;
      const int nr = k == 0 ? 3 : 2;
      // rows outside the stem map are the grouped conv's zero padding
      for (int j = 0; j < nr; ++j) {
        const int r = r_first + j;
        if (r >= 0 && r < p.Hs) continue;
        unsigned char* rowp = y1t + (long)((r + 3) % 3) * Y1W * p.PS;
        for (int i = tid; i < Y1W * cpp; i += 256) *reinterpret_cast<u32x4*>(rowp + (long)i * 16) = (u32x4){0u, 0u, 0u, 0u};
      }
      const int r_lo = max(r_first, 0), r_hi = min(r_first + nr, p.Hs);
      const int ntiles = max(r_hi - r_lo, 0) * tiles_per_row;
      // (addressing as in the pipelined kernel below: ring slots and row bases are scalars of the step, lanes past the map's
      // edge duplicate the edge column - the same value to the same address - instead of being masked)
      const int inv_tpr = 65536 / tiles_per_row + 1;                     // tile / tiles_per_row for tile < 64
      const int pitch_in = INW * 8, pitch_y1 = Y1W * p.PS;
      const bool q_hi = (q & 2) != 0, q_lt2 = q < 2;
      const int qcol = 16 * (q & 1);
      bf16_t* sc_row = p.shortcut + ((long)n * p.Ho + oy) * p.Wo * p.C1;
      const unsigned char* inb = reinterpret_cast<const unsigned char*>(inp);
      for (int tI = wv; tI < ntiles; tI += 4) {
        const int rq = (tI * inv_tpr) >> 16, ct = tI - rq * tiles_per_row;
        const int r = r_lo + rq;
        const int ro0 = ((2 * r + 7) & 7) * pitch_in, ro1 = ((2 * r + 8) & 7) * pitch_in, ro2 = ((2 * r + 9) & 7) * pitch_in;
        const int y1row = ((r + 3) % 3) * pitch_y1;
        const bool do_ds = (r & 1) == 0 && (r >> 1) == oy;
        const int cc = min(ct * 16 + px, p.Ws - 1);
        const int colb = cc * 16 + qcol;
        const unsigned char* xp0 = inb + colb + (q_hi ? ro1 : ro0);
        const unsigned char* xp1 = q_lt2 ? inb + colb + ro2 : reinterpret_cast<const unsigned char*>(zero16);
        const bf16x8 xf0 = *reinterpret_cast<const bf16x8*>(xp0);
        const bf16x8 xf1 = *reinterpret_cast<const bf16x8*>(xp1);
        f32x4 sa0 = {0.f, 0.f, 0.f, 0.f}, sa1 = {0.f, 0.f, 0.f, 0.f};
        sa0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(swf[0][0], xf0, sa0, 0, 0, 0);
        sa1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(swf[1][0], xf0, sa1, 0, 0, 0);
        sa0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(swf[0][1], xf1, sa0, 0, 0, 0);
        sa1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(swf[1][1], xf1, sa1, 0, 0, 0);
        bf16x8 sf;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          sf[e] = (bf16_t)(sa0[e] * ssc[e] + ssh[e]);
          sf[4 + e] = (bf16_t)(sa1[e] * ssc[4 + e] + ssh[4 + e]);
        }
        sf = relu_bf16x8(sf);
        unsigned char* y1p = y1t + y1row + __mul24(cc + 1, p.PS) + 8 * q;
#pragma unroll
        for (int t = 0; t < NT1; ++t) {
          f32x4 a1 = {0.f, 0.f, 0.f, 0.f};
          a1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w1r[t], sf, a1, 0, 0, 0);
          bf16x4 o;
#pragma unroll
          for (int e = 0; e < 4; ++e) o[e] = (bf16_t)(a1[e] * c1s[t][e] + c1h[t][e]);
          *reinterpret_cast<bf16x4*>(y1p + t * 32) = relu_bf16x4(o);
        }
        if (do_ds) {
          bf16_t* scp = sc_row + __mul24(cc >> 1, p.C1) + 4 * q;
          const bool even = (cc & 1) == 0;
#pragma unroll
          for (int t = 0; t < NT1; ++t) {
            f32x4 ad = {0.f, 0.f, 0.f, 0.f};
            ad = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wdr[t], sf, ad, 0, 0, 0);
            bf16x4 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = (bf16_t)(ad[e] * cds[t][e] + cdh[t][e]);
            if (even && t * 16 + 4 * q < p.C1) *reinterpret_cast<bf16x4*>(scp + t * 16) = o;
          }
        }
      }
    }
    __syncthreads();
    // ---- C. grouped 3x3 stride 2 for output row oy from the ring (conv1 rows 2 oy - 1 .. 2 oy + 1)
    {
      int off[5];
#pragma unroll
      for (int ks = 0; ks < 5; ++ks) off[ks] = ((2 * oy - 1 + tdy[ks] + 3) % 3) * Y1W * p.PS + toff[ks];
      const int ntl = (p.Wo + 15) >> 4;
      bf16_t* yout = p.y2 + ((long)n * p.Ho + oy) * p.Wo * p.C1;
      for (int mt = wv / units; mt < ntl; mt += mstep) {
        const int ox = mt * 16 + px;
        const bool pok = ox < p.Wo;
        const unsigned char* base = y1t + (long)((pok ? ox : 0) * 2) * p.PS;
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < 5; ++ks) {
          const bf16x8 xf = *reinterpret_cast<const bf16x8*>(base + off[ks]);
          acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[ks], xf, acc, 0, 0, 0);
        }
        if (pok && ch0 < p.C1) {
          bf16x4 o;
#pragma unroll
          for (int r = 0; r < 4; ++r) o[r] = (bf16_t)(acc[r] * sc2[r] + sh2[r]);
          o = relu_bf16x4(o);
#pragma unroll
          for (int r = 0; r < 4; ++r) psum[r] += (float)o[r];
          *reinterpret_cast<bf16x4*>(yout + (long)ox * p.C1 + ch0) = o;
        }
      }
    }
    __syncthreads();
  }
  // ---- squeeze sums of the strip
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    float v = psum[r];
    v += __shfl_xor(v, 1, 64);
    v += __shfl_xor(v, 2, 64);
    v += __shfl_xor(v, 4, 64);
    v += __shfl_xor(v, 8, 64);
    if (px == 0) red[wv][q * 4 + r] = v;
  }
  __syncthreads();
  if (tid < units * 16) {
    const int u = tid >> 4, cc = tid & 15;
    float sres = 0.f;
    for (int w2 = u; w2 < 4; w2 += units) sres += red[w2][cc];
    const int chn = u * 16 + cc;
    if (chn < p.C1) p.pooled[((long)n * nstrips + bnd) * p.C1 + chn] = sres;
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// Pipelined form of the strip walk: the three phases of a step depend on each other but not on the SAME step of the other
// phases, so eight waves split into roles and run skewed by one step each - wave 0 fetches and normalises the input rows of
// step j (its global loads for step j + 1 stay in flight across the barrier: the barrier fences LDS only), waves 1-5 turn
// the rows of step j - 1 into conv1 rows, waves 6-7 contract the conv1 rows of step j - 2 into an output row - with ONE
// barrier per step.  Each role holds only its own weights in registers (119 VGPRs: two workgroups per CU).  Rings: 16 input
// rows, 5 conv1 rows.  CSP <= 32 and 16-byte-aligned crops; the rolling kernel above takes everything else.
// Measured (800 frames 224 x 224, regnety_002 widths): band form 434 us, rolling 389 us, this 247 us; the role split
// 2 / 4 / 2 gives 252 us, ten waves (2 / 6 / 2) do not fit twice on a CU and give 315 us.
#define LDS_BARRIER()                                                  \
  do {                                                                 \
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");    \
    __builtin_amdgcn_s_barrier();                                      \
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");    \
  } while (0)

typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

constexpr int PIPE_NWA = 1, PIPE_NWB = 5;                                // waves of roles A and B (C: 2)
constexpr int PIPE_NT = 64 * (PIPE_NWA + PIPE_NWB + 2);
template <int NT1>
__global__ __launch_bounds__(PIPE_NT, 4) void s1_front_pipe_kernel(const FrontP p, int S) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  __shared__ float red[2][16];
  const long lid = xcd_logical_id(blockIdx.x, gridDim.x);
  const int nstrips = (p.Ho + S - 1) / S;
  int bnd, n;
  td_split(lid, nstrips, n, bnd);
  const int oy0 = bnd * S;
  const int nrows_out = min(S, p.Ho - oy0);
  const int NS = nrows_out + 2;                                         // super-steps: the roles run skewed by one each
  const int INW = p.cw + 2, Y1W = p.Ws + 2;
  // smem[0..16) stays zero: the operand of the two unused k-slots of the 3 x 3 x 2 stem patch
  bf16_t* inp = reinterpret_cast<bf16_t*>(smem + 16);                  // [16][INW][4] ring of normalised input rows
  const int in_bytes = (16 * INW * 8 + 15) & ~15;
  unsigned char* y1t = smem + 16 + in_bytes;                                 // [5][Y1W][PS] ring of conv1 rows
  const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int px = lane & 15, q = lane >> 4;
  const int cpp = p.PS >> 4;
  {
    const bf16x4 z4 = {(bf16_t)0.f, (bf16_t)0.f, (bf16_t)0.f, (bf16_t)0.f};
    if (tid < 2) reinterpret_cast<bf16x4*>(smem)[tid] = z4;
    for (int i = tid; i < 32; i += PIPE_NT)
      *reinterpret_cast<bf16x4*>(inp + ((long)(i >> 1) * INW + ((i & 1) ? (p.cw + 1) : 0)) * 4) = z4;
    for (int i = tid; i < 5 * 2 * cpp; i += PIPE_NT) {
      const int j = i % cpp, rc = i / cpp;
      const int rr = rc >> 1, col = (rc & 1) ? (p.Ws + 1) : 0;
      *reinterpret_cast<u32x4*>(y1t + ((long)rr * Y1W + col) * p.PS + j * 16) = (u32x4){0u, 0u, 0u, 0u};
    }
  }
  __syncthreads();

  if (wv < PIPE_NWA) {
    // =============================== role A: uint8 planes -> normalised bf16 pixels in the ring (slot = (row + 16) & 15)
    const float na[3] = {1.0f / (255.0f * 0.229f), 1.0f / (255.0f * 0.224f), 1.0f / (255.0f * 0.225f)};
    const float nb[3] = {-0.485f / 0.229f, -0.456f / 0.224f, -0.406f / 0.225f};
    const uint8_t* src = p.frames + (long)n * 3 * p.H * p.W;
    const long plane = (long)p.H * p.W;
    const bf16x4 z4 = {(bf16_t)0.f, (bf16_t)0.f, (bf16_t)0.f, (bf16_t)0.f};
    const int nch8 = p.cw >> 3;
    const IDiv dn8(nch8);
    constexpr int NTA = PIPE_NWA * 64, MAXP = 256 / NTA;               // 4 rows x (cw / 8) chunks <= MAXP x NTA lanes: cw <= 512
    auto fetch = [&](int iy, int c8, u32x2* v) {
      const int scol = p.flip ? (p.cw - 8 - 8 * c8) : 8 * c8;
      const uint8_t* s0 = src + (long)(p.top + iy) * p.W + p.left + scol;
      v[0] = *reinterpret_cast<const u32x2*>(s0);
      v[1] = *reinterpret_cast<const u32x2*>(s0 + plane);
      v[2] = *reinterpret_cast<const u32x2*>(s0 + 2 * plane);
    };
    auto put = [&](int iy, int c8, bool ok, const u32x2* v) {
      bf16x4* dst = reinterpret_cast<bf16x4*>(inp + ((long)((iy + 16) & 15) * INW + 8 * c8 + 1) * 4);
      if (ok) {
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const unsigned int sh8 = 8 * (e & 3);
          bf16x4 o;
          o[0] = (bf16_t)fmaf((float)((v[0][e >> 2] >> sh8) & 0xffu), na[0], nb[0]);
          o[1] = (bf16_t)fmaf((float)((v[1][e >> 2] >> sh8) & 0xffu), na[1], nb[1]);
          o[2] = (bf16_t)fmaf((float)((v[2][e >> 2] >> sh8) & 0xffu), na[2], nb[2]);
          o[3] = (bf16_t)0.f;
          dst[p.flip ? (7 - e) : e] = o;
        }
      } else {
#pragma unroll
        for (int e = 0; e < 8; ++e) dst[e] = z4;
      }
    };
    // steady state: the four rows of step k are fetched one super-step ahead into `pre`
    u32x2 preA[MAXP][3], preB[MAXP][3];
    // (loads are unconditional - clamped lanes and rows re-read a valid chunk - so that the compiler can count them: a load
    // under a branch makes every later wait a vmcnt(0), which would drain the fetch-ahead)
    auto issue = [&](int k, u32x2 (*pre)[3]) {
      const int ir0 = 4 * (oy0 + k);
#pragma unroll
      for (int ps = 0; ps < MAXP; ++ps) {
        const int i = min(tid + ps * NTA, 4 * nch8 - 1);
        int r, c8;
        dn8.divmod(i, r, c8);
        fetch(min(ir0 + r, p.ch - 1), c8, pre[ps]);
      }
    };
    auto drain = [&](int k, u32x2 (*pre)[3]) {
      const int ir0 = 4 * (oy0 + k);
#pragma unroll
      for (int ps = 0; ps < MAXP; ++ps) {
        const int i = tid + ps * NTA;
        int r, c8;
        dn8.divmod(i, r, c8);
        const int iy = ir0 + r;
        if (i < 4 * nch8) put(iy, c8, iy < p.ch, pre[ps]);
      }
    };
    auto stepA = [&](int j, u32x2 (*cur)[3], u32x2 (*nxt)[3]) {
      if (j < nrows_out) {
        issue(min(j + 1, nrows_out - 1), nxt);
        if (j == 0) {
          const int ir0 = 4 * oy0 - 3;
          for (int i = tid; i < 7 * nch8; i += NTA) {
            int r, c8;
            dn8.divmod(i, r, c8);
            const int iy = ir0 + r;
            const bool ok = iy >= 0 && iy < p.ch;
            u32x2 v[3];
            if (ok) fetch(iy, c8, v);
            put(iy, c8, ok, v);
          }
        } else {
          drain(j, cur);
        }
      }
      LDS_BARRIER();
    };
    for (int j = 0; j < NS; j += 2) {
      stepA(j, preA, preB);
      if (j + 1 < NS) stepA(j + 1, preB, preA);
    }
  } else if (wv < PIPE_NWA + PIPE_NWB) {
    // =============================== role B: stem -> conv1 rows into the ring (slot = (r + 5) % 5), downsample of the even row
    // The SIMDs' instruction issue is what this role is short of: everything that depends only on the step (ring slots) is
    // scalar, a tile's per-lane addressing is a handful of adds and selects, and lanes past the map's edge duplicate the edge
    // column (same value to the same address) instead of being masked.
    constexpr int NWB = PIPE_NWB;
    const int bw = wv - PIPE_NWA, btid = tid - PIPE_NWA * 64;
    bf16x8 swf[2][2];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) swf[t][ks] = p.stem_wf[(t * 2 + ks) * 64 + lane];
    float ssc[8], ssh[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int c = (e < 4) ? (4 * q + e) : (16 + 4 * q + e - 4);
      ssc[e] = p.stem_sc[c];
      ssh[e] = p.stem_sh[c];
    }
    bf16x8 w1r[NT1], wdr[NT1];
    float c1s[NT1][4], c1h[NT1][4], cds[NT1][4], cdh[NT1][4];
#pragma unroll
    for (int t = 0; t < NT1; ++t) {
      w1r[t] = p.w1f[t * 64 + lane];
      wdr[t] = p.wdf[t * 64 + lane];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int chn = t * 16 + 4 * q + e;
        const bool ok = chn < p.C1;
        c1s[t][e] = ok ? p.sc1[chn] : 0.f; c1h[t][e] = ok ? p.sh1[chn] : 0.f;
        cds[t][e] = ok ? p.scd[chn] : 0.f; cdh[t][e] = ok ? p.shd[chn] : 0.f;
      }
    }
    const int tiles_per_row = (p.Ws + 15) >> 4;
    const int inv_tpr = 65536 / tiles_per_row + 1;                       // item / tiles_per_row for item < 64 (scalar)
    const int pitch_in = INW * 8, pitch_y1 = Y1W * p.PS;
    const bool q_hi = (q & 2) != 0, q_lt2 = q < 2;
    const int qcol = 16 * (q & 1);                                        // second pixel pair of the 3-wide patch
    const int st_lane = 8 * q;                                            // this lane's 4 channels inside a pixel of the ring
    for (int j = 0; j < NS; ++j) {
      const int k = j - 1;
      if (k >= 0 && k < nrows_out) {
        const int oy = oy0 + k;
        const int r_first = k == 0 ? 2 * oy - 1 : 2 * oy;
        const int nr = k == 0 ? 3 : 2;
        for (int jr = 0; jr < nr; ++jr) {                                // rows outside the stem map: the 3x3's zero padding
          const int r = r_first + jr;
          if (r >= 0 && r < p.Hs) continue;
          unsigned char* rowp = y1t + (long)((r + 5) % 5) * pitch_y1;
          for (int i = btid; i < Y1W * cpp; i += NWB * 64) *reinterpret_cast<u32x4*>(rowp + (long)i * 16) = (u32x4){0u, 0u, 0u, 0u};
        }
        const int r_lo = max(r_first, 0), r_hi = min(r_first + nr, p.Hs);
        const int ntiles = max(r_hi - r_lo, 0) * tiles_per_row;
        bf16_t* sc_row = p.shortcut + ((long)n * p.Ho + oy) * p.Wo * p.C1;
        for (int tI = bw; tI < ntiles; tI += NWB) {
          const int rq = (tI * inv_tpr) >> 16, ct = tI - rq * tiles_per_row;
          const int r = r_lo + rq;
          const int ro0 = ((2 * r + 15) & 15) * pitch_in, ro1 = ((2 * r + 16) & 15) * pitch_in, ro2 = ((2 * r + 17) & 15) * pitch_in;
          const int y1row = ((r + 5) % 5) * pitch_y1;
          const bool do_ds = (r & 1) == 0 && (r >> 1) == oy;
          const int cc = min(ct * 16 + px, p.Ws - 1);
          const int colb = cc * 16 + qcol;
          const int a0 = 16 + colb + (q_hi ? ro1 : ro0);
          const int a1 = q_lt2 ? 16 + colb + ro2 : 0;                     // k-slots 6, 7 of the patch: the zero block
          const bf16x8 xf0 = *reinterpret_cast<const bf16x8*>(smem + a0);
          const bf16x8 xf1 = *reinterpret_cast<const bf16x8*>(smem + a1);
          f32x4 sa0 = {0.f, 0.f, 0.f, 0.f}, sa1 = {0.f, 0.f, 0.f, 0.f};
          sa0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(swf[0][0], xf0, sa0, 0, 0, 0);
          sa1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(swf[1][0], xf0, sa1, 0, 0, 0);
          sa0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(swf[0][1], xf1, sa0, 0, 0, 0);
          sa1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(swf[1][1], xf1, sa1, 0, 0, 0);
          bf16x8 sf;
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            sf[e] = (bf16_t)(sa0[e] * ssc[e] + ssh[e]);
            sf[4 + e] = (bf16_t)(sa1[e] * ssc[4 + e] + ssh[4 + e]);
          }
          sf = relu_bf16x8(sf);
          unsigned char* y1p = y1t + y1row + __mul24(cc + 1, p.PS) + st_lane;
          f32x4 a1v[NT1], adv[NT1];
#pragma unroll
          for (int t = 0; t < NT1; ++t) {
            a1v[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
            a1v[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w1r[t], sf, a1v[t], 0, 0, 0);
          }
          if (do_ds) {
#pragma unroll
            for (int t = 0; t < NT1; ++t) {
              adv[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
              adv[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wdr[t], sf, adv[t], 0, 0, 0);
            }
          }
#pragma unroll
          for (int t = 0; t < NT1; ++t) {
            bf16x4 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = (bf16_t)(a1v[t][e] * c1s[t][e] + c1h[t][e]);
            *reinterpret_cast<bf16x4*>(y1p + t * 32) = relu_bf16x4(o);
          }
          if (do_ds) {
            bf16_t* scp = sc_row + __mul24(cc >> 1, p.C1) + 4 * q;
            const bool even = (cc & 1) == 0;
#pragma unroll
            for (int t = 0; t < NT1; ++t) {
              bf16x4 o;
#pragma unroll
              for (int e = 0; e < 4; ++e) o[e] = (bf16_t)(adv[t][e] * cds[t][e] + cdh[t][e]);
              if (even && t * 16 + 4 * q < p.C1) *reinterpret_cast<bf16x4*>(scp + t * 16) = o;
            }
          }
        }
      }
      LDS_BARRIER();
    }
  } else {
    // =============================== role C: grouped 3x3 stride 2 for one output row from the conv1 ring + squeeze sums
    const int cwv = wv - PIPE_NWA - PIPE_NWB;
    const int unit = NT1 == 2 ? cwv : 0;
    const int mt0 = NT1 == 2 ? 0 : cwv, mstep = NT1 == 2 ? 1 : 2;
    bf16x8 wf[5];
#pragma unroll
    for (int ks = 0; ks < 5; ++ks) wf[ks] = p.w2f[(unit * 5 + ks) * 64 + lane];
    const int pitch_y1 = Y1W * p.PS;
    int tdy[5], toff[5];                                                  // per k-slot: tap row (0..2) and the tap's offset in a row
#pragma unroll
    for (int ks = 0; ks < 5; ++ks) {
      const int sidx = 4 * ks + q;
      const int half = sidx / 9, tap = sidx - half * 9;
      const int dy = tap / 3, dx = tap - dy * 3;
      tdy[ks] = sidx < 18 ? dy : 0;
      toff[ks] = (sidx < 18 ? dx * p.PS + half * 16 : 0) + unit * 32;
    }
    const int ch0 = unit * 16 + q * 4;
    float sc2[4], sh2[4], psum[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const bool ok = ch0 + r < p.C1;
      sc2[r] = ok ? p.sc2[ch0 + r] : 0.f;
      sh2[r] = ok ? p.sh2[ch0 + r] : 0.f;
      psum[r] = 0.f;
    }
    const int ntl = (p.Wo + 15) >> 4;
    const bool chok = ch0 < p.C1;
    for (int j = 0; j < NS; ++j) {
      const int k = j - 2;
      if (k >= 0) {
        const int oy = oy0 + k;
        const int rs0 = ((2 * oy + 4) % 5) * pitch_y1, rs1 = ((2 * oy + 5) % 5) * pitch_y1, rs2 = ((2 * oy + 6) % 5) * pitch_y1;
        int off[5];
#pragma unroll
        for (int ks = 0; ks < 5; ++ks) off[ks] = toff[ks] + (tdy[ks] == 0 ? rs0 : (tdy[ks] == 1 ? rs1 : rs2));
        bf16_t* yout = p.y2 + ((long)n * p.Ho + oy) * p.Wo * p.C1 + ch0;
        for (int mt = mt0; mt < ntl; mt += mstep) {
          const int ox = mt * 16 + px;
          const bool pok = ox < p.Wo;
          const unsigned char* base = y1t + __mul24(min(ox, p.Wo - 1), 2 * p.PS);
          f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int ks = 0; ks < 5; ++ks) {
            const bf16x8 xf = *reinterpret_cast<const bf16x8*>(base + off[ks]);
            acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[ks], xf, acc, 0, 0, 0);
          }
          if (pok && chok) {
            bf16x4 o;
#pragma unroll
            for (int r = 0; r < 4; ++r) o[r] = (bf16_t)(acc[r] * sc2[r] + sh2[r]);
            o = relu_bf16x4(o);
#pragma unroll
            for (int r = 0; r < 4; ++r) psum[r] += (float)o[r];
            *reinterpret_cast<bf16x4*>(yout + __mul24(ox, p.C1)) = o;
          }
        }
      }
      LDS_BARRIER();
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      float v = psum[r];
      v += __shfl_xor(v, 1, 64);
      v += __shfl_xor(v, 2, 64);
      v += __shfl_xor(v, 4, 64);
      v += __shfl_xor(v, 8, 64);
      if (px == 0) red[cwv][q * 4 + r] = v;
    }
  }
  __syncthreads();
  if (tid < NT1 * 16) {
    const int u = tid >> 4, cc = tid & 15;
    const float sres = NT1 == 2 ? red[u][cc] : red[0][cc] + red[1][cc];
    const int chn = u * 16 + cc;
    if (chn < p.C1) p.pooled[((long)n * nstrips + bnd) * p.C1 + chn] = sres;
  }
}

// the pipelined form is the default where it applies; TDEED_FRONT_PIPE=0 -> the rolling kernel
static bool front_pipe() {
  static int s = -1;
  if (s < 0) { const char* e = getenv("TDEED_FRONT_PIPE"); s = e ? (atoi(e) != 0) : 1; }
  return s != 0;
}
static size_t front_pipe_smem(int crop_w, int Ws, int PS) {
  return 16 + (((size_t)16 * (crop_w + 2) * 8 + 15) & ~(size_t)15) + (size_t)5 * (Ws + 2) * PS;
}
static bool front_pipe_shape(int crop_w, int Ws, int CSP) {
  return front_pipe() && CSP <= 32 && crop_w <= 512 && crop_w % 16 == 0 && front_pipe_smem(crop_w, Ws, CSP * 2 + 16) <= FRONT_LDS_CAP;
}

// rows per strip of the strip-walking forms (0: the band form at the top); TDEED_FRONT_ROLL overrides.  A function of the
// SHAPE only - tdeed_s1_front_parts must agree with the launch whatever the alignment of the frames turns out to be.
static bool front_pipe_shape(int crop_w, int Ws, int CSP);
static int front_roll(int crop_w, int Ws, int CSP) {
  static int s = -2;
  if (s == -2) { const char* e = getenv("TDEED_FRONT_ROLL"); s = e ? atoi(e) : -1; if (e && s < 0) s = 0; }
  if (s >= 0) return s;
  return (front_pipe_shape(crop_w, Ws, CSP) || CSP > 32) ? 28 : 8;        // measured: A/B lines in profiles/r03_front_strips.txt
}
static size_t front_roll_smem(int crop_w, int Ws, int PS) {
  return (((size_t)8 * (crop_w + 2) * 8 + 15) & ~(size_t)15) + (size_t)3 * (Ws + 2) * PS;
}

static long front_cap() {
  static long cap = -1;
  if (cap < 0) {
    cap = 48 * 1024;      // 48 KB: three workgroups per CU, measured best with three batches in flight
                                               // (64 KB / two per CU was best with two: DESIGN section 4)
    if (cap > FRONT_LDS_CAP) cap = FRONT_LDS_CAP;
  }
  return cap;
}
static int front_band(int cw, int Ws, int PS, int Ho) {
  int best = 0;
  for (int band = 1; band <= Ho && band <= 8; ++band) {
    const int ny1 = 2 * (band - 1) + 3, nin = 2 * (ny1 - 1) + 3;
    const long bytes = (((long)nin * (cw + 2) * 8 + 15) & ~15L) + (long)ny1 * (Ws + 2) * PS;
    if (bytes <= front_cap() || band == 1) best = band;
  }
  return best;
}

static size_t front_smem(int crop_w, int Ws, int PS, int band) {
  const int ny1 = 2 * (band - 1) + 3, nin = 2 * (ny1 - 1) + 3;
  return (((size_t)nin * (crop_w + 2) * 8 + 15) & ~(size_t)15) + (size_t)ny1 * (Ws + 2) * PS;
}

// number of partial-sum rows of `pooled` per frame, or 0 when even a one-row band does not fit LDS
// (very wide frames): the caller then uses the unfused stem / conv1 / conv2 / downsample kernels.
extern "C" int tdeed_s1_front_parts(int crop_h, int crop_w, int C1) {
  const int Hs = (crop_h + 1) / 2, Ws = (crop_w + 1) / 2, Ho = (Hs + 1) / 2;
  if (C1 % 8 != 0 || C1 < 8 || C1 > 64) return 0;
  const int CSP = C1 > 32 ? 64 : (C1 > 16 ? 32 : 16);
  if (const int S = front_roll(crop_w, Ws, CSP); S > 0 && front_roll_smem(crop_w, Ws, CSP * 2 + 16) <= FRONT_LDS_CAP) return (Ho + S - 1) / S;
  const int band = front_band(crop_w, Ws, CSP * 2 + 16, Ho);
  if (band <= 0 || front_smem(crop_w, Ws, CSP * 2 + 16, band) > FRONT_LDS_CAP) return 0;
  return (Ho + band - 1) / band;
}

extern "C" int tdeed_s1_front_fwd(const uint8_t* frames, int N, int H, int W, int crop_top, int crop_left,
                                  int crop_h, int crop_w, int flip, const void* stem_wf, const float* stem_sc,
                                  const float* stem_sh, int C1, const void* w1f, const float* sc1,
                                  const float* sh1, const void* wdf, const float* scd, const float* shd,
                                  const void* w2f, const float* sc2, const float* sh2, void* y2, void* shortcut,
                                  float* pooled, void* stream) {
  TD_CHECK(frames && stem_wf && stem_sc && stem_sh && w1f && sc1 && sh1 && wdf && scd && shd && w2f && sc2 && sh2 &&
               y2 && shortcut && pooled, "s1_front: null pointer");
  TD_CHECK(N > 0 && N <= 65535 && crop_h > 0 && crop_w > 0 && crop_top >= 0 && crop_left >= 0 &&
               crop_top + crop_h <= H && crop_left + crop_w <= W, "s1_front: bad geometry");
  TD_CHECK(C1 % 8 == 0 && C1 >= 8 && C1 <= 64, "s1_front: C1=%d unsupported", C1);
  FrontP p;
  p.frames = frames; p.H = H; p.W = W; p.top = crop_top; p.left = crop_left; p.ch = crop_h; p.cw = crop_w; p.flip = flip;
  p.stem_wf = (const bf16x8*)stem_wf; p.stem_sc = stem_sc; p.stem_sh = stem_sh;
  p.w1f = (const bf16x8*)w1f; p.sc1 = sc1; p.sh1 = sh1;
  p.wdf = (const bf16x8*)wdf; p.scd = scd; p.shd = shd;
  p.w2f = (const bf16x8*)w2f; p.sc2 = sc2; p.sh2 = sh2;
  p.y2 = (bf16_t*)y2; p.shortcut = (bf16_t*)shortcut; p.pooled = pooled;
  p.C1 = C1;
  p.CSP = C1 > 32 ? 64 : (C1 > 16 ? 32 : 16);
  p.PS = p.CSP * 2 + 16;
  p.Hs = (crop_h + 1) / 2; p.Ws = (crop_w + 1) / 2;
  p.Ho = (p.Hs + 1) / 2; p.Wo = (p.Ws + 1) / 2;
  p.vec16 = (W % 16 == 0) && (crop_w % 16 == 0) && (crop_left % 16 == 0) && (((long)H * W) % 16 == 0) &&
            (((uintptr_t)frames & 15) == 0);
  if (const int S = front_roll(crop_w, p.Ws, p.CSP); S > 0 && front_roll_smem(crop_w, p.Ws, p.PS) <= FRONT_LDS_CAP) {
    const int nstrips = (p.Ho + S - 1) / S;
    const size_t smr = front_roll_smem(crop_w, p.Ws, p.PS);
    p.band = S; p.nbands = nstrips;
    static TdDevOnce attr_r;
    if (!attr_r.get()) {
      hipError_t e = hipFuncSetAttribute((const void*)s1_front_roll_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, FRONT_LDS_CAP);
      if (e == hipSuccess) e = hipFuncSetAttribute((const void*)s1_front_roll_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, FRONT_LDS_CAP);
      if (e == hipSuccess) e = hipFuncSetAttribute((const void*)s1_front_roll_kernel<4>, hipFuncAttributeMaxDynamicSharedMemorySize, FRONT_LDS_CAP);
      if (e != hipSuccess) { tdeed_set_error("s1_front: hipFuncSetAttribute: %s", hipGetErrorString(e)); return TDEED_ERR_RUNTIME; }
      attr_r.set();
    }
    const int nt1r = p.CSP >> 4;
    if (p.vec16 && front_pipe_shape(crop_w, p.Ws, p.CSP)) {
      static TdDevOnce attr_p;
      if (!attr_p.get()) {
        hipError_t e = hipFuncSetAttribute((const void*)s1_front_pipe_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, FRONT_LDS_CAP);
        if (e == hipSuccess) e = hipFuncSetAttribute((const void*)s1_front_pipe_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, FRONT_LDS_CAP);
        if (e != hipSuccess) { tdeed_set_error("s1_front: hipFuncSetAttribute: %s", hipGetErrorString(e)); return TDEED_ERR_RUNTIME; }
        attr_p.set();
      }
      const size_t smp = front_pipe_smem(crop_w, p.Ws, p.PS);
      if (nt1r == 1) hipLaunchKernelGGL(s1_front_pipe_kernel<1>, dim3(nstrips * N), dim3(PIPE_NT), smp, (hipStream_t)stream, p, S);
      else hipLaunchKernelGGL(s1_front_pipe_kernel<2>, dim3(nstrips * N), dim3(PIPE_NT), smp, (hipStream_t)stream, p, S);
      TD_LAUNCH_CHECK("s1_front_pipe");
      return TDEED_OK;
    }
    if (nt1r == 1) hipLaunchKernelGGL(s1_front_roll_kernel<1>, dim3(nstrips * N), dim3(256), smr, (hipStream_t)stream, p, S);
    else if (nt1r == 2) hipLaunchKernelGGL(s1_front_roll_kernel<2>, dim3(nstrips * N), dim3(256), smr, (hipStream_t)stream, p, S);
    else hipLaunchKernelGGL(s1_front_roll_kernel<4>, dim3(nstrips * N), dim3(256), smr, (hipStream_t)stream, p, S);
    TD_LAUNCH_CHECK("s1_front_roll");
    return TDEED_OK;
  }
  p.band = front_band(crop_w, p.Ws, p.PS, p.Ho);
  TD_CHECK(p.band > 0 && front_smem(crop_w, p.Ws, p.PS, p.band) <= FRONT_LDS_CAP,
           "s1_front: a row band of %d px does not fit LDS (use the unfused kernels)", crop_w);
  p.nbands = (p.Ho + p.band - 1) / p.band;
  const int ny1 = 2 * (p.band - 1) + 3, nin = 2 * (ny1 - 1) + 3;
  const size_t smem = (((size_t)nin * (crop_w + 2) * 8 + 15) & ~(size_t)15) + (size_t)ny1 * (p.Ws + 2) * p.PS;
  static TdDevOnce attr_set;
  if (!attr_set.get()) {
    hipError_t e = hipFuncSetAttribute((const void*)s1_front_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, FRONT_LDS_CAP);
    if (e == hipSuccess) e = hipFuncSetAttribute((const void*)s1_front_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, FRONT_LDS_CAP);
    if (e == hipSuccess) e = hipFuncSetAttribute((const void*)s1_front_kernel<4>, hipFuncAttributeMaxDynamicSharedMemorySize, FRONT_LDS_CAP);
    if (e != hipSuccess) { tdeed_set_error("s1_front: hipFuncSetAttribute: %s", hipGetErrorString(e)); return TDEED_ERR_RUNTIME; }
    attr_set.set();
  }
  const int nt1 = p.CSP >> 4;
  if (nt1 == 1) hipLaunchKernelGGL(s1_front_kernel<1>, dim3(p.nbands * N), dim3(256), smem, (hipStream_t)stream, p);
  else if (nt1 == 2) hipLaunchKernelGGL(s1_front_kernel<2>, dim3(p.nbands * N), dim3(256), smem, (hipStream_t)stream, p);
  else hipLaunchKernelGGL(s1_front_kernel<4>, dim3(p.nbands * N), dim3(256), smem, (hipStream_t)stream, p);
  TD_LAUNCH_CHECK("s1_front");
  return TDEED_OK;
}

// =============================================================================================
// Training stem: the same pre-processing + 3x3 stride-2 conv on the MFMA pipe as phase A / B of s1_front_kernel, as its
// own launch that writes the RAW conv output z0 (bf16; BatchNorm runs on batch statistics in training) and, per workgroup,
// the per-channel sum / sum of squares of what it stored -- the BatchNorm statistics come out of the conv's epilogue instead
// of a second pass over the 112^2 x 32 map.  One workgroup = one frame x a band of output rows; the normalised input band
// ([row][col][RGB0] bf16, zero halo) sits in LDS, a lane's 8 k-values (two horizontally adjacent taps) are one 16-byte
// read, 2 k-steps x 2 channel tiles per 16 pixels.  The VALU stem (train.hip) spends 27 x 32 FMAs per pixel and ran at a
// quarter of what writing z0 costs (519 vs ~120 us for 800 frames of 224^2).
// Precision: both operands are split into a bf16 head and a bf16 tail (x = xh + xl, w = wh + wl) and the product is taken as
// wh xh + wh xl + wl xh (three MFMAs, fp32 accumulation; the dropped wl xl term is 2^-16 of the result), so z0 is the fp32
// conv rounded once to bf16 like the VALU stem's -- rounding the operands themselves to bf16 (as the inference front may)
// measurably degrades the gradient direction of the noise-prone small tensors (SE weights) in training.
struct StemGeo { int H, W, top, left, ch, cw, vec16; };

// normalised input rows [in_r0, in_r0 + nin) of one frame -> LDS as [row][col -1 .. cw][RGB0] bf16 (heads in inp, the bf16
// tails of the fp32 values in inl when SPLIT), zero outside the image, h-flip applied; ends with the two slack pixels behind
// the last row zeroed (odd widths read one pixel past the row).  Caller synchronises.
template <typename IN, bool SPLIT>
__device__ __forceinline__ void stem_fill_patch(bf16_t* inp, bf16_t* inl, const IN* src, const StemGeo& g, int in_r0, int nin,
                                                int flip) {
  const int tid = threadIdx.x;
  const int INW = g.cw + 2;
  const float na[3] = {1.0f / (255.0f * 0.229f), 1.0f / (255.0f * 0.224f), 1.0f / (255.0f * 0.225f)};
  const float nb[3] = {-0.485f / 0.229f, -0.456f / 0.224f, -0.406f / 0.225f};
  const long plane = (long)g.H * g.W;
  const bf16x4 z4 = {(bf16_t)0.f, (bf16_t)0.f, (bf16_t)0.f, (bf16_t)0.f};
  bool done = false;
  if constexpr (sizeof(IN) == 1) {
    if (g.vec16) {
      // a thread owns 16 pixels of one row: three 16-byte loads (R, G, B planes) in flight, 16 8-byte LDS stores
      const int nch = g.cw >> 4;
      const int total = nin * nch;
      const IDiv dnch(nch);
      for (int i = tid; i < total; i += 256) {
        int k, r;
        dnch.divmod(i, r, k);
        const int iy = in_r0 + r;
        bf16x4* dst = reinterpret_cast<bf16x4*>(inp + ((long)r * INW + 16 * k + 1) * 4);
        bf16x4* dsl = reinterpret_cast<bf16x4*>(inl + ((long)r * INW + 16 * k + 1) * 4);
        if (iy >= 0 && iy < g.ch) {
          const int scol = flip ? (g.cw - 16 - 16 * k) : 16 * k;
          const uint8_t* s0 = reinterpret_cast<const uint8_t*>(src) + (long)(g.top + iy) * g.W + g.left + scol;
          const u32x4 v0 = *reinterpret_cast<const u32x4*>(s0);
          const u32x4 v1 = *reinterpret_cast<const u32x4*>(s0 + plane);
          const u32x4 v2 = *reinterpret_cast<const u32x4*>(s0 + 2 * plane);
#pragma unroll
          for (int e = 0; e < 16; ++e) {
            const unsigned int sh8 = 8 * (e & 3);
            const float f0 = fmaf((float)((v0[e >> 2] >> sh8) & 0xffu), na[0], nb[0]);
            const float f1 = fmaf((float)((v1[e >> 2] >> sh8) & 0xffu), na[1], nb[1]);
            const float f2 = fmaf((float)((v2[e >> 2] >> sh8) & 0xffu), na[2], nb[2]);
            bf16x4 o;
            o[0] = (bf16_t)f0; o[1] = (bf16_t)f1; o[2] = (bf16_t)f2; o[3] = (bf16_t)0.f;
            dst[flip ? (15 - e) : e] = o;
            if constexpr (SPLIT) {
              bf16x4 l;
              l[0] = (bf16_t)(f0 - (float)o[0]); l[1] = (bf16_t)(f1 - (float)o[1]); l[2] = (bf16_t)(f2 - (float)o[2]);
              l[3] = (bf16_t)0.f;
              dsl[flip ? (15 - e) : e] = l;
            }
          }
        } else {
#pragma unroll
          for (int e = 0; e < 16; ++e) {
            dst[e] = z4;
            if constexpr (SPLIT) dsl[e] = z4;
          }
        }
      }
      done = true;
    }
  }
  if (!done) {
    const IDiv dcw(g.cw);
    for (int i = tid; i < nin * g.cw; i += 256) {
      int r, ix;
      dcw.divmod(i, r, ix);
      const int iy = in_r0 + r;
      bf16x4 v4 = z4, l4 = z4;
      if (iy >= 0 && iy < g.ch) {
        const int sx = flip ? (g.cw - 1 - ix) : ix;
        const long o = (long)(g.top + iy) * g.W + (g.left + sx);
#pragma unroll
        for (int c3 = 0; c3 < 3; ++c3) {
          const float f = fmaf((float)src[c3 * plane + o], na[c3], nb[c3]);
          v4[c3] = (bf16_t)f;
          l4[c3] = (bf16_t)(f - (float)v4[c3]);
        }
      }
      *reinterpret_cast<bf16x4*>(inp + ((long)r * INW + ix + 1) * 4) = v4;
      if constexpr (SPLIT) *reinterpret_cast<bf16x4*>(inl + ((long)r * INW + ix + 1) * 4) = l4;
    }
  }
  for (int i = tid; i < nin * 2; i += 256) {        // halo columns (input col -1 and cw)
    const long o = ((long)(i >> 1) * INW + ((i & 1) ? (g.cw + 1) : 0)) * 4;
    *reinterpret_cast<bf16x4*>(inp + o) = z4;
    if constexpr (SPLIT) *reinterpret_cast<bf16x4*>(inl + o) = z4;
  }
  if (tid < 2) {                                     // slack
    *reinterpret_cast<bf16x4*>(inp + ((long)nin * INW + tid) * 4) = z4;
    if constexpr (SPLIT) *reinterpret_cast<bf16x4*>(inl + ((long)nin * INW + tid) * 4) = z4;
  }
}

struct StemP {
  const void* frames; StemGeo g; int flip;
  const unsigned char* flip_mask;
  const float* wf;                                  // [2][2][64][8] fp32 fragments (engine.stem_frags_on_device)
  bf16_t* z; float* colpart;                        // z [N][Hs][Ws][32]; colpart [N * nbands][2][32]
  int Hs, Ws, band, nbands;
};

template <typename IN>
__global__ __launch_bounds__(256) void stem_mfma_kernel(const StemP p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  __shared__ float red[4][64];
  const long lid = xcd_logical_id(blockIdx.x, gridDim.x);
  int bnd, n;
  td_split(lid, p.nbands, n, bnd);
  const int r0 = bnd * p.band;
  const int nrows = min(p.band, p.Hs - r0);
  const int in_r0 = 2 * r0 - 1, nin = 2 * nrows + 1;
  const int INW = p.g.cw + 2;
  bf16_t* inp = reinterpret_cast<bf16_t*>(smem);                        // heads [nin][INW][4] (+ 16 B of slack behind)
  const int plane_el = (nin * INW + 2) * 4;                             // elements per plane
  bf16_t* inl = inp + plane_el;                                         // tails, same layout
  const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int px = lane & 15, q = lane >> 4;
  const int flip = p.flip_mask ? (int)p.flip_mask[n] : p.flip;
  stem_fill_patch<IN, true>(inp, inl, reinterpret_cast<const IN*>(p.frames) + (long)n * 3 * p.g.H * p.g.W, p.g, in_r0, nin, flip);
  __syncthreads();

  bf16x8 swf[2][2], swl[2][2];
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const float* wp = p.wf + ((long)((t * 2 + ks) * 64 + lane)) * 8;
      const f32x4 w0 = *reinterpret_cast<const f32x4*>(wp), w1 = *reinterpret_cast<const f32x4*>(wp + 4);
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float f = e < 4 ? w0[e] : w1[e - 4];
        swf[t][ks][e] = (bf16_t)f;
        swl[t][ks][e] = (bf16_t)(f - (float)swf[t][ks][e]);
      }
    }
  float s1[8], s2[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) s1[e] = s2[e] = 0.f;
  const int tiles_per_row = (p.Ws + 15) >> 4;
  const int ntiles = nrows * tiles_per_row;
  const IDiv dtpr(tiles_per_row);
  bf16_t* zf = p.z + (long)n * p.Hs * p.Ws * 32;
  for (int tI = wv; tI < ntiles; tI += 4) {
    int rq, rm;
    dtpr.divmod(tI, rq, rm);
    const int c = rm * 16 + px;
    const bool cok = c < p.Ws;
    const int cc = cok ? c : (p.Ws - 1);
    f32x4 sa[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const int s = 4 * ks + q;                          // slot -> (ky = s >> 1, half = s & 1); slots 6, 7 carry zero weights
      const int ky = s >> 1, half = s & 1;
      const int prow = (s < 6) ? (2 * rq + ky) : 0;
      const int pcol = (s < 6) ? (2 * cc + 2 * half) : 0;
      const bf16x8 xf = *reinterpret_cast<const bf16x8*>(inp + (prow * INW + pcol) * 4);
      const bf16x8 xl = *reinterpret_cast<const bf16x8*>(inl + (prow * INW + pcol) * 4);
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        sa[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(swl[t][ks], xf, sa[t], 0, 0, 0);
        sa[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(swf[t][ks], xl, sa[t], 0, 0, 0);
        sa[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(swf[t][ks], xf, sa[t], 0, 0, 0);
      }
    }
    bf16x4 o0, o1;
#pragma unroll
    for (int e = 0; e < 4; ++e) { o0[e] = (bf16_t)sa[0][e]; o1[e] = (bf16_t)sa[1][e]; }
    if (cok) {
      bf16_t* zp = zf + ((long)(r0 + rq) * p.Ws + c) * 32 + 4 * q;
      *reinterpret_cast<bf16x4*>(zp) = o0;
      *reinterpret_cast<bf16x4*>(zp + 16) = o1;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float a = (float)o0[e], b = (float)o1[e];
        s1[e] += a; s2[e] = fmaf(a, a, s2[e]);
        s1[4 + e] += b; s2[4 + e] = fmaf(b, b, s2[4 + e]);
      }
    }
  }
  // fold: the 16 pixel lanes of a k-slot group, then the 4 waves; channel of element e: 4q + e (e < 4), 16 + 4q + e - 4
#pragma unroll
  for (int e = 0; e < 8; ++e) {
#pragma unroll
    for (int o = 1; o < 16; o <<= 1) {
      s1[e] += __shfl_xor(s1[e], o, 64);
      s2[e] += __shfl_xor(s2[e], o, 64);
    }
  }
  if (px == 0) {
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int chn = (e < 4) ? (4 * q + e) : (16 + 4 * q + e - 4);
      red[wv][chn] = s1[e];
      red[wv][32 + chn] = s2[e];
    }
  }
  __syncthreads();
  if (tid < 64) p.colpart[lid * 64 + tid] = red[0][tid] + red[1][tid] + red[2][tid] + red[3][tid];
}

static int stem_band(int cw, int Hs) {
  const long rowb = (long)(cw + 2) * 16;                 // head + tail planes
  static long cap = -1;                                  // 48 KB: three workgroups per CU
  if (cap < 0) cap = 48 * 1024;
  int band = (int)((cap / rowb - 1) / 2);
  if (band > 16) band = 16;
  if (band < 1) band = 1;
  if (band > Hs) band = Hs;
  return band;
}

// row bands per frame = rows of colpart per frame
extern "C" int tdeed_stem_mfma_parts(int crop_h, int crop_w) {
  const int Hs = (crop_h + 1) / 2;
  const int band = stem_band(crop_w, Hs);
  if (2 * ((size_t)(2 * band + 1) * (crop_w + 2) * 8 + 16) > FRONT_LDS_CAP) return 0;
  return (Hs + band - 1) / band;
}

extern "C" int tdeed_stem_mfma_fwd(const void* frames, int frames_f32, int N, int H, int W, int crop_top, int crop_left,
                                   int crop_h, int crop_w, int flip, const unsigned char* flip_mask, const void* wfrag,
                                   void* z, float* colpart, void* stream) {
  TD_CHECK(frames && wfrag && z && colpart, "stem_mfma: null pointer");
  TD_CHECK(N > 0 && crop_h > 0 && crop_w > 0 && crop_top >= 0 && crop_left >= 0 && crop_top + crop_h <= H &&
               crop_left + crop_w <= W, "stem_mfma: bad geometry");
  StemP p;
  p.frames = frames; p.g.H = H; p.g.W = W; p.g.top = crop_top; p.g.left = crop_left; p.g.ch = crop_h; p.g.cw = crop_w;
  p.flip = flip; p.flip_mask = flip_mask; p.wf = (const float*)wfrag; p.z = (bf16_t*)z; p.colpart = colpart;
  p.Hs = (crop_h + 1) / 2; p.Ws = (crop_w + 1) / 2;
  p.band = stem_band(crop_w, p.Hs);
  p.nbands = (p.Hs + p.band - 1) / p.band;
  const size_t smem = 2 * ((size_t)(2 * p.band + 1) * (crop_w + 2) * 8 + 16);
  TD_CHECK(smem <= FRONT_LDS_CAP, "stem_mfma: a row band of %d px does not fit LDS (use tdeed_stem_fwd)", crop_w);
  const long grid = (long)p.nbands * N;
  TD_CHECK(grid <= 0x7fffffffL, "stem_mfma: grid too large");
  p.g.vec16 = !frames_f32 && (W % 16 == 0) && (crop_w % 16 == 0) && (crop_left % 16 == 0) && (((long)H * W) % 16 == 0) &&
              (((uintptr_t)frames & 15) == 0);
  static TdDevOnce attr_set;
  if (!attr_set.get()) {
    hipError_t e = hipFuncSetAttribute((const void*)stem_mfma_kernel<uint8_t>, hipFuncAttributeMaxDynamicSharedMemorySize, FRONT_LDS_CAP);
    if (e == hipSuccess) e = hipFuncSetAttribute((const void*)stem_mfma_kernel<float>, hipFuncAttributeMaxDynamicSharedMemorySize, FRONT_LDS_CAP);
    if (e != hipSuccess) { tdeed_set_error("stem_mfma: hipFuncSetAttribute: %s", hipGetErrorString(e)); return TDEED_ERR_RUNTIME; }
    attr_set.set();
  }
  if (frames_f32) hipLaunchKernelGGL(stem_mfma_kernel<float>, dim3((unsigned)grid), dim3(256), smem, (hipStream_t)stream, p);
  else hipLaunchKernelGGL(stem_mfma_kernel<uint8_t>, dim3((unsigned)grid), dim3(256), smem, (hipStream_t)stream, p);
  TD_LAUNCH_CHECK("stem_mfma");
  return TDEED_OK;
}

// =============================================================================================
// Stem weight gradient (bf16) without an im2col: dW[co][c][ky][kx] = sum_px dz[px][co] x[2y+ky-1][2x+kx-1][c].
// Pixels are the contraction index.  Both MFMA operands are "one column, 8 consecutive rows" of a row-major LDS image and
// come from transposing reads (ds_read_b64_tr_b16, td_tr_read8):
//   A = dz^T : the dz rows of the band, copied to LDS as they are ([px][32 co], 16-byte copies);
//   B        : for one ky, the row of pixel x is the 16 contiguous bf16 at patch[2y+ky][2x] = (kx 0..3, c 0..3) of the
//              normalised input band ([row][col][RGB0], the forward's image); consecutive pixels are 16 bytes apart, so the
//              "rows" overlap -- every lane supplies its own address, which is all the transposing read needs.  Columns with
//              kx = 3 or c = 3 are by-products and dropped at the end.
// One workgroup = one frame x 16 output rows (the partial layout of the kernel it replaces), in sub-bands of 4 rows; wave w
// takes row w of the sub-band, 32 pixels per MFMA k-step, 2 channel tiles x 3 ky tiles accumulate in registers over the
// whole 16 rows; the waves' partial sums are folded through LDS.  The previous MFMA form built dz^T and the im2col patch^T
// element-wise in LDS (59 two-byte stores and 27 reads per pixel) and ran at 546 us for 800 frames of 224^2 where reading
// dz costs ~120 us.
struct StemWgP {
  const void* frames; StemGeo g; int flip;
  const unsigned char* flip_mask;
  const bf16_t* dz; float* part;
  int Ho, Wo, WoP, groups;
  // bz given: `dz` is the (masked) gradient g at the OUTPUT of the stem's BatchNorm and the BatchNorm backward
  // dz = k1 g + k2 z + k3 is applied while the rows are staged (bz = the raw stem output z, bsums = (sum g, sum g xhat), the
  // statistics and weight of that BatchNorm): the dz map (1.3 GB at cfg3) is neither written nor read back
  const bf16_t* bz; const float* bsums; const float* bmean; const float* brstd; const float* bw; float inv_M;
};

template <typename IN>
__global__ __launch_bounds__(256) void stem_wgrad_tr_kernel(const StemWgP p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int n = blockIdx.x / p.groups, grp = blockIdx.x - n * p.groups;
  const int INW = p.g.cw + 2;
  bf16_t* inp = reinterpret_cast<bf16_t*>(smem);                          // [9][INW][4] + slack
  const int patch_el = ((9 * INW + 2) * 4 + 7) & ~7;
  bf16_t* dzs = inp + patch_el;                                           // [4][WoP][32]
  const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int g16 = lane >> 4, l16 = lane & 15, qq = l16 >> 2, pp = l16 & 3;
  const int flip = p.flip_mask ? (int)p.flip_mask[n] : p.flip;
  const IN* src = reinterpret_cast<const IN*>(p.frames) + (long)n * 3 * p.g.H * p.g.W;
  const bf16_t* dzf = p.dz + (long)n * p.Ho * p.Wo * 32;
  f32x4 acc[2][3];
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int ky = 0; ky < 3; ++ky) acc[t][ky] = (f32x4){0.f, 0.f, 0.f, 0.f};
  const int cpr = p.WoP * 4;                                              // 16-byte chunks per dz image row
  // (cpr is a multiple of 4 and 256 too: a thread stages the same 8 channels in every iteration -> its constants in registers)
  float k1[8], k2[8], k3[8];
  const bf16_t* zf = p.bz ? p.bz + (long)n * p.Ho * p.Wo * 32 : nullptr;
  if (p.bz) {
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int c = (tid & 3) * 8 + e;
      const float rs = p.brstd[c], mu = p.bmean[c], m1 = p.bsums[c] * p.inv_M, m2 = p.bsums[32 + c] * p.inv_M;
      k1[e] = p.bw[c] * rs;
      k2[e] = -k1[e] * rs * m2;
      k3[e] = k1[e] * (mu * rs * m2 - m1);
    }
  }
  for (int sb = 0; sb < 4; ++sb) {
    const int oy0 = grp * 16 + sb * 4;
    if (oy0 >= p.Ho) break;
    if (sb) __syncthreads();                                              // the previous sub-band has been consumed
    stem_fill_patch<IN, false>(inp, nullptr, src, p.g, 2 * oy0 - 1, 9, flip);
    for (int i = tid; i < 4 * cpr; i += 256) {
      const int ry = i / cpr, cx = i - ry * cpr;
      const int oy = oy0 + ry;
      u32x4 v = {0u, 0u, 0u, 0u};
      if (oy < p.Ho && cx < p.Wo * 4) {
        v = *reinterpret_cast<const u32x4*>(dzf + ((long)oy * p.Wo) * 32 + (long)cx * 8);
        if (zf) {
          const u32x4 zv = *reinterpret_cast<const u32x4*>(zf + ((long)oy * p.Wo) * 32 + (long)cx * 8);
          const bf16x8 g8 = *reinterpret_cast<const bf16x8*>(&v), z8 = *reinterpret_cast<const bf16x8*>(&zv);
          bf16x8 o8;
#pragma unroll
          for (int e = 0; e < 8; ++e) o8[e] = (bf16_t)fmaf(k1[e], (float)g8[e], fmaf(k2[e], (float)z8[e], k3[e]));
          v = *reinterpret_cast<const u32x4*>(&o8);
        }
      }
      *reinterpret_cast<u32x4*>(dzs + (long)i * 8) = v;
    }
    __syncthreads();
    // wave wv: output row oy0 + wv; its pixels in k-steps of 32
    const bf16_t* arow = dzs + (long)wv * p.WoP * 32;
    for (int x0 = 0; x0 < p.WoP; x0 += 32) {
      const int xr = x0 + 8 * g16 + qq;                                   // this lane's address row (pixel) in rows 0..3
      bf16x8 af[2], bfr[3];
#pragma unroll
      for (int t = 0; t < 2; ++t)
        af[t] = td_tr_read8(arow + (long)xr * 32 + t * 16 + 4 * pp, arow + (long)(xr + 4) * 32 + t * 16 + 4 * pp);
#pragma unroll
      for (int ky = 0; ky < 3; ++ky) {
        const bf16_t* prow = inp + (long)(2 * wv + ky) * INW * 4;
        bfr[ky] = td_tr_read8(prow + 2 * xr * 4 + 4 * pp, prow + 2 * (xr + 4) * 4 + 4 * pp);
      }
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) acc[t][ky] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[t], bfr[ky], acc[t][ky], 0, 0, 0);
    }
  }
  __syncthreads();
  float* red = reinterpret_cast<float*>(smem);                            // [4 waves][6 tiles][64 lanes][4]
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int ky = 0; ky < 3; ++ky) *reinterpret_cast<f32x4*>(red + ((wv * 6 + t * 3 + ky) * 64 + lane) * 4) = acc[t][ky];
  __syncthreads();
  float* out = p.part + ((long)n * p.groups + grp) * 864;
  for (int i = tid; i < 6 * 64; i += 256) {
    const int tile = i >> 6, l = i & 63;
    const int t = tile / 3, ky = tile - 3 * t;
    const int j = l & 15, kx = j >> 2, c = j & 3;
    if (kx < 3 && c < 3) {
      f32x4 a = *reinterpret_cast<const f32x4*>(red + ((0 * 6 + tile) * 64 + l) * 4);
#pragma unroll
      for (int w = 1; w < 4; ++w) {
        const f32x4 b = *reinterpret_cast<const f32x4*>(red + ((w * 6 + tile) * 64 + l) * 4);
        a[0] += b[0]; a[1] += b[1]; a[2] += b[2]; a[3] += b[3];
      }
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int co = 16 * t + 4 * (l >> 4) + e;
        out[co * 27 + c * 9 + ky * 3 + kx] = a[e];
      }
    }
  }
}

// launcher used by tdeed_stem_wgrad (trunk_bwd.hip); returns 0 when the geometry does not fit (the caller falls back)
int td_stem_wgrad_tr_launch(const void* frames, int frames_f32, int N, int H, int W, int crop_top, int crop_left, int crop_h,
                            int crop_w, int flip, const unsigned char* flip_mask, const void* dz, float* part,
                            hipStream_t st, const void* bz, const float* bsums, const float* bmean, const float* brstd,
                            const float* bw) {
  StemWgP p;
  p.bz = (const bf16_t*)bz; p.bsums = bsums; p.bmean = bmean; p.brstd = brstd; p.bw = bw;
  p.inv_M = 1.0f / ((float)N * (float)((crop_h + 1) / 2) * (float)((crop_w + 1) / 2));
  p.frames = frames; p.g.H = H; p.g.W = W; p.g.top = crop_top; p.g.left = crop_left; p.g.ch = crop_h; p.g.cw = crop_w;
  p.flip = flip; p.flip_mask = flip_mask; p.dz = (const bf16_t*)dz; p.part = part;
  p.Ho = (crop_h + 1) / 2; p.Wo = (crop_w + 1) / 2;
  p.WoP = (p.Wo + 31) / 32 * 32;
  p.groups = (p.Ho + 15) / 16;
  const size_t patch_b = (size_t)((((9 * (crop_w + 2) + 2) * 4 + 7) & ~7)) * 2;
  size_t smem = patch_b + (size_t)4 * p.WoP * 64;
  if (smem < 4 * 6 * 64 * 4 * sizeof(float)) smem = 4 * 6 * 64 * 4 * sizeof(float);
  // the last k-step of a row reads patch columns up to 2 (WoP - 1) + 3: inside the shared allocation (the dz image follows)
  if (smem > FRONT_LDS_CAP || ((size_t)8 * (crop_w + 2) + 2 * (size_t)p.WoP + 4) * 8 > patch_b + (size_t)4 * p.WoP * 64) return 0;
  p.g.vec16 = !frames_f32 && (W % 16 == 0) && (crop_w % 16 == 0) && (crop_left % 16 == 0) && (((long)H * W) % 16 == 0) &&
              (((uintptr_t)frames & 15) == 0);
  static TdDevOnce attr_set;
  if (!attr_set.get()) {
    hipError_t e = hipFuncSetAttribute((const void*)stem_wgrad_tr_kernel<uint8_t>, hipFuncAttributeMaxDynamicSharedMemorySize, FRONT_LDS_CAP);
    if (e == hipSuccess) e = hipFuncSetAttribute((const void*)stem_wgrad_tr_kernel<float>, hipFuncAttributeMaxDynamicSharedMemorySize, FRONT_LDS_CAP);
    if (e != hipSuccess) return 0;
    attr_set.set();
  }
  const long grid = (long)N * p.groups;
  if (grid > 0x7fffffffL) return 0;
  if (frames_f32) hipLaunchKernelGGL(stem_wgrad_tr_kernel<float>, dim3((unsigned)grid), dim3(256), smem, st, p);
  else hipLaunchKernelGGL(stem_wgrad_tr_kernel<uint8_t>, dim3((unsigned)grid), dim3(256), smem, st, p);
  return 1;
}
