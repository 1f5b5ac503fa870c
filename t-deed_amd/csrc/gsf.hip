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
#include <cstdlib>

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

// copy n 16-byte items global -> LDS with all loads of a batch of 8 in flight together (a plain copy loop compiles
// to load / wait / write per item: one full memory round trip each)
__device__ __forceinline__ void copy16_batched(u32x4* __restrict__ dst, const u32x4* __restrict__ src, int n) {
  for (int i0 = threadIdx.x; i0 < n; i0 += 256 * 8) {
    u32x4 r[8];
#pragma unroll
    for (int b = 0; b < 8; ++b) r[b] = src[min(i0 + b * 256, n - 1)];
#pragma unroll
    for (int b = 0; b < 8; ++b)
      if (i0 + b * 256 < n) dst[i0 + b * 256] = r[b];
  }
}

// stage rows [y0-1, y0+rows-1) of frame f as relu(bn(x)) in bf16, 8-channel chunks, into a[rows][w+2][PSQ]:
// zero halo ring, channels >= F and the stride pad are zero.  Branch-free loads (clamped addresses, select
// afterwards) so a batch of 8 x 16 B per lane is in flight at once; sbn = BN scale[F] | shift[F] in LDS.
__device__ __forceinline__ void gsf_stage_frame(unsigned char* a, const bf16_t* __restrict__ x, long f, int h, int w,
                                                int C, int F, int y0, int rows, int nch, int PSQ,
                                                const float* sbn) {
  const int tid = threadIdx.x, WP = w + 2;
  const int cpp = PSQ >> 4;                                     // 16-byte pieces per pixel incl. pad
  const int total = rows * WP * cpp;
  const bf16_t* xf = x + f * h * w * C;
  const IDiv dcpp(cpp), dwp(WP);
  for (int i0 = tid; i0 < total; i0 += 256 * 8) {
    u32x4 v[8];
    int cj[8];
#pragma unroll
    for (int b8 = 0; b8 < 8; ++b8) {
      const int i = min(i0 + b8 * 256, total - 1);
      int j, pix, ry, rx;
      dcpp.divmod(i, pix, j);
      dwp.divmod(pix, ry, rx);
      const int yy = y0 - 1 + ry, xx = rx - 1;
      const bool ok = j < nch && yy >= 0 && yy < h && xx >= 0 && xx < w;
      cj[b8] = ok ? j : -1;
      v[b8] = *reinterpret_cast<const u32x4*>(ok ? xf + ((long)yy * w + xx) * C + j * 8 : xf);
    }
    TD_ISSUE_FENCE();
#pragma unroll
    for (int b8 = 0; b8 < 8; ++b8) {
      const int i = i0 + b8 * 256;
      if (i >= total) continue;
      u32x4 o = {0u, 0u, 0u, 0u};
      if (cj[b8] >= 0) {
        const int c0 = cj[b8] * 8;
        float fv[8];
        Chunk<bf16_t>::load(reinterpret_cast<const bf16_t*>(&v[b8]), fv);
        bf16x8 r;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const int c = min(c0 + e, F - 1);
          r[e] = (c0 + e < F) ? (bf16_t)fmaxf(fmaf(fv[e], sbn[c], sbn[F + c]), 0.f) : (bf16_t)0.f;
        }
        o = *reinterpret_cast<u32x4*>(&r);
      }
      *reinterpret_cast<u32x4*>(a + (long)i * 16) = o;
    }
  }
}

// BN scale / shift [F] each -> LDS sbn[2F]
__device__ __forceinline__ void gsf_stage_bn(float* sbn, const float* __restrict__ bn_scale,
                                             const float* __restrict__ bn_shift, int F) {
  for (int i = threadIdx.x; i < 2 * F; i += 256) sbn[i] = i < F ? bn_scale[i] : bn_shift[i - F];
}

// ---- launch 1a, bf16 throughput mode: the same partial sums as an implicit GEMM on the MFMA pipe.
// D[jg][pixel] = sum_k Wt[jg][k] A[k][pixel], jg = (temporal tap j, gate g) = 6 of the 16 MFMA rows,
// k = (spatial tap, 8-channel chunk): a lane's 8 k-values are one 16-byte read of the BN+ReLU'd frame
// (bf16, zero halo, pixel stride an odd number of 16-byte slots => conflict-free ds_read_b128).
// wqf: [KS][64] fragments (engine.pack_gsf_q_frags), rows jg >= 6 and channels of the other gate group are 0.
// The launch is a latency chain per workgroup (weights + frame -> LDS -> K-loop -> store), so it is built to run it ONCE:
//  * nfr frames per workgroup (launcher: 2 when one frame each would need a second round of workgroups -- 800 frames
//    against 768 resident ones doubled the launch's duration);
//  * weights, BatchNorm table and the frames' pieces are requested together (one memory round trip; the halo ring outside the
//    frame and the pad slots are zero-filled while the loads travel, not fetched from a dummy address);
//  * a wave multiplies two pixel tiles at a time, four k-steps of LDS reads in flight in front of their MFMAs
//    (a step-by-step loop was offset read -> fragment read -> MFMA, serial, per k-step).
//  * WREG (KS <= 28, i.e. F <= 96): a wave copies the weight fragments from LDS into REGISTERS once, ahead of its tiles.  The
//    K-loop was bound by the LDS pipe, which all waves of a CU share: a 16x16x32 MFMA fed with both operands from LDS costs
//    2 x 1 KB = 16 LDS cycles against the 4 cycles per MFMA the CU's four SIMDs sustain; time stamps gave ~1000 cycles per
//    four k-steps.  With the weights in registers only the activation fragment is read.  (Fetching them from global memory
//    per wave instead -- no LDS copy at all -- cost 4 x the weight bytes per workgroup and was slower at the cfg2 sizes.)
template <bool WREG>
__global__ __launch_bounds__(256) void gsf_q_mfma_kernel(const bf16_t* __restrict__ x, int h, int w, int C, int F,
                                                         int band, int nch, int PSQ, int KS,
                                                         const float* __restrict__ bn_scale,
                                                         const float* __restrict__ bn_shift,
                                                         const bf16x8* __restrict__ wqf, float* __restrict__ Q,
                                                         int nbq, int nfr, int nframes) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smq[];
  // frames in contiguous chunks per XCD (all three launches of a site use the same numbering): the maps of frames t-1 / t+1
  // that the next launch reads were written through, and are read through, the same L2
  const long lid = xcd_logical_id(blockIdx.x, gridDim.x);
  int fq_, bq_;
  td_split(lid, nbq, fq_, bq_);
  const int f0 = fq_ * nfr;
  const int nf = min(nfr, nframes - f0);
  const int y0 = bq_ * band;
  const int y1 = min(h, y0 + band);
  const int nrow = y1 - y0, rows = nrow + 2, WP = w + 2;
  const int fbytes = (band + 2) * WP * PSQ;                     // one frame's band
  bf16x8* wl = reinterpret_cast<bf16x8*>(smq);                  // [KS][64]
  unsigned char* a = smq + (size_t)KS * 64 * 16;               // [nfr][rows][WP][PSQ]
  constexpr int KSR = 28;
  const int FP8 = nch * 8;                                      // BatchNorm table padded to whole chunks (zeros: relu(0 * x + 0) = 0)
  float* sbn = reinterpret_cast<float*>(a + (size_t)nfr * fbytes);   // [2][FP8]
  int* soff = reinterpret_cast<int*>(sbn + 2 * FP8);            // [KS*4]
  const int tid = threadIdx.x;
  // ---- requests first: weight fragments, the frames' interior 16-byte pieces, the BatchNorm table
  const int nwp = KS * 64;
  // rows fetched: the band's own rows and the halo rows that exist in the frame (bands of a frame taller than one band)
  const int ylo = max(y0 - 1, 0), yhi = min(y1 + 1, h);
  const int pin = (yhi - ylo) * w * nch, ptot = nf * pin;       // fetched pieces of one frame / of the workgroup
  const IDiv dpin(pin), dnch(nch), dw_(w);
  u32x4 wr[8], v[8];
  int doff[8], dc0[8];
  auto issue_w = [&](int it) {
#pragma unroll
    for (int b8 = 0; b8 < 8; ++b8)
      wr[b8] = reinterpret_cast<const u32x4*>(wqf)[min(it * 2048 + tid + b8 * 256, nwp - 1)];
  };
  auto commit_w = [&](int it) {
#pragma unroll
    for (int b8 = 0; b8 < 8; ++b8) {
      const int i = it * 2048 + tid + b8 * 256;
      if (i < nwp) reinterpret_cast<u32x4*>(wl)[i] = wr[b8];
    }
  };
  const bf16_t* __restrict__ xb = x + (long)f0 * h * w * C;      // uniform base; lane offsets are unsigned 32-bit elements
  auto issue_f = [&](int it) {
#pragma unroll
    for (int b8 = 0; b8 < 8; ++b8) {
      const int i = min(it * 2048 + tid + b8 * 256, ptot - 1);
      int fr, r, pix, j, py, px;
      dpin.divmod(i, fr, r);
      dnch.divmod(r, pix, j);
      dw_.divmod(pix, py, px);
      doff[b8] = fr * fbytes + ((ylo + py - (y0 - 1)) * WP + px + 1) * PSQ + j * 16;
      dc0[b8] = j * 8;
      v[b8] = *reinterpret_cast<const u32x4*>(xb + ((unsigned)((fr * h + ylo + py) * w + px) * (unsigned)C + (unsigned)j * 8u));
    }
  };
  auto commit_f = [&](int it) {
#pragma unroll
    for (int b8 = 0; b8 < 8; ++b8) {
      if (it * 2048 + tid + b8 * 256 >= ptot) continue;
      const int c0 = dc0[b8];
      float fv[8];
      Chunk<bf16_t>::load(reinterpret_cast<const bf16_t*>(&v[b8]), fv);
      const f32x4 s0 = *reinterpret_cast<const f32x4*>(sbn + c0), s1 = *reinterpret_cast<const f32x4*>(sbn + c0 + 4);
      const f32x4 h0 = *reinterpret_cast<const f32x4*>(sbn + FP8 + c0), h1 = *reinterpret_cast<const f32x4*>(sbn + FP8 + c0 + 4);
      bf16x8 r;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        r[e] = (bf16_t)fmaxf(fmaf(fv[e], s0[e], h0[e]), 0.f);
        r[4 + e] = (bf16_t)fmaxf(fmaf(fv[4 + e], s1[e], h1[e]), 0.f);
      }
      TD_LDS_CHECK((a - smq) + doff[b8], 16, (unsigned char*)sbn - smq);
      *reinterpret_cast<u32x4*>(a + doff[b8]) = *reinterpret_cast<u32x4*>(&r);
    }
  };
  issue_w(0);
  issue_f(0);
  float bnv[2];
#pragma unroll
  for (int u = 0; u < 2; ++u) {                                 // entry i of [2][FP8]: scale | shift, zeros behind channel F
    const int i = tid + u * 256;
    const int c = i < FP8 ? i : i - FP8;
    bnv[u] = (i < FP8 ? bn_scale : bn_shift)[min(c, F - 1)];
    if (c >= F || i >= 2 * FP8) bnv[u] = 0.f;
  }
  TD_ISSUE_FENCE();
  // ---- while they travel: zero the bands (halo ring, pad slots, pixels of missing halo rows), k-slot offsets
  for (int i = tid; i < nf * (fbytes >> 4); i += 256) reinterpret_cast<u32x4*>(a)[i] = (u32x4){0u, 0u, 0u, 0u};
  for (int s_ = tid; s_ < KS * 4; s_ += 256) {                  // byte offset of every k-slot (tap, 8-channel chunk)
    int tap, ck;
    dnch.divmod(s_, tap, ck);
    const int dy = (tap * 11) >> 5, dx = tap - dy * 3;          // tap / 3 for tap < 12
    soff[s_] = tap < 9 ? (dy * WP + dx) * PSQ + ck * 16 : 0;
  }
#pragma unroll
  for (int u = 0; u < 2; ++u)
    if (tid + u * 256 < 2 * FP8) sbn[tid + u * 256] = bnv[u];
  commit_w(0);
  for (int it = 1; it * 2048 < nwp; ++it) { issue_w(it); TD_ISSUE_FENCE(); commit_w(it); }
  __syncthreads();                                              // zeros and the BatchNorm table are in place
  commit_f(0);
  for (int it = 1; it * 2048 < ptot; ++it) { issue_f(it); TD_ISSUE_FENCE(); commit_f(it); }
  __syncthreads();
  // ---- K-loop: tile pairs (t, t + 1) of the workgroup's nf * ntl pixel tiles
  const int lane = tid & 63, wv = tid >> 6, pl = lane & 15, q = lane >> 4;
  bf16x8 wreg[WREG ? KSR : 1];
  if constexpr (WREG) {                                         // this wave's copy of the fragments: LDS -> registers, once
#pragma unroll
    for (int ks = 0; ks < KSR; ++ks)
      if (ks < KS) wreg[ks] = wl[ks * 64 + lane];
  }
  const int npix = nrow * w;
  const int ntl = (npix + 15) >> 4;
  const int ntot = nf * ntl;
  const IDiv dntl(ntl);
  for (int t0 = wv * 2; t0 < ntot; t0 += 8) {
    const unsigned char* base[2];
    bool pok[2];
    long qrow[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int t = min(t0 + u, ntot - 1);
      int fr, mt;
      dntl.divmod(t, fr, mt);
      const int pp = mt * 16 + pl;
      pok[u] = (t0 + u < ntot) && pp < npix;
      const int pc = pok[u] ? pp : 0;
      int py, px;
      dw_.divmod(pc, py, px);
      base[u] = a + fr * fbytes + ((long)py * WP + px) * PSQ;
      qrow[u] = (long)(f0 + fr) * h * w + (long)(y0 + py) * w + px;
    }
    f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
    if constexpr (WREG) {
#pragma unroll
      for (int g = 0; g < KSR / 4; ++g) {
        if (g * 4 < KS) {                                        // (uniform)
          int o[4];
          bf16x8 a0[4], a1[4];
#pragma unroll
          for (int j = 0; j < 4; ++j) o[j] = soff[min(g * 4 + j, KS - 1) * 4 + q];
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            a0[j] = *reinterpret_cast<const bf16x8*>(base[0] + o[j]);
            a1[j] = *reinterpret_cast<const bf16x8*>(base[1] + o[j]);
          }
#pragma unroll
          for (int j = 0; j < 4; ++j)
            if (g * 4 + j < KS) {
              acc0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wreg[g * 4 + j], a0[j], acc0, 0, 0, 0);
              acc1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wreg[g * 4 + j], a1[j], acc1, 0, 0, 0);
            }
        }
      }
    } else {
      for (int ks0 = 0; ks0 < KS; ks0 += 4) {
        int o[4];
        bf16x8 wf[4], a0[4], a1[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) o[j] = soff[min(ks0 + j, KS - 1) * 4 + q];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          wf[j] = wl[min(ks0 + j, KS - 1) * 64 + lane];
          a0[j] = *reinterpret_cast<const bf16x8*>(base[0] + o[j]);
          a1[j] = *reinterpret_cast<const bf16x8*>(base[1] + o[j]);
        }
#pragma unroll
        for (int j = 0; j < 4; ++j)
          if (ks0 + j < KS) {
            acc0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[j], a0[j], acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[j], a1[j], acc1, 0, 0, 0);
          }
      }
    }
#pragma unroll
    for (int u = 0; u < 2; ++u)
      if (pok[u]) {
        const f32x4 acc = u ? acc1 : acc0;
        float* dst = Q + qrow[u] * 6;
        if (q == 0) { dst[0] = acc[0]; dst[1] = acc[1]; dst[2] = acc[2]; dst[3] = acc[3]; }
        else if (q == 1) { dst[4] = acc[0]; dst[5] = acc[1]; }
      }
  }
}

// deterministic spatial sums of gate*x and x over one frame: channel pairs across lanes (coalesced), S pixel slices
// per pair, ordered reduce over the slices.  sg = gates [hw][2] in LDS.
// The frame's values do not depend on the gates: a thread's first GS_NPRE pixels are REQUESTED (gsf_sums_issue) in front of the
// gate computation, together with the tap maps the gates are made of -- one memory round trip for the launch instead of two
// (round 6: the launch is a latency chain, 7-9 us per site; slices longer than GS_NPRE pixels take further batches of 8).
constexpr int GS_NPRE = 24;
template <typename T>
struct GsfSums {
  float v0[GS_NPRE], v1[GS_NPRE];
  int s, cp, S;
  bool act;
};
template <typename T>
__device__ __forceinline__ void gsf_sums_issue(GsfSums<T>& st, const T* __restrict__ x, long f, int hw, int C, int F) {
  const int nq = F >> 1;
  st.S = 256 / nq;
  IDiv(nq).divmod((int)threadIdx.x, st.s, st.cp);
  st.act = st.s < st.S;
  const T* __restrict__ xf = x + f * hw * C;                    // uniform base, unsigned 32-bit lane offsets
  const unsigned c2 = 2u * (unsigned)(st.act ? st.cp : 0);
  const int s0 = st.act ? st.s : 0;
#pragma unroll
  for (int b = 0; b < GS_NPRE; ++b) Pair<T>::load(xf + ((unsigned)min(s0 + b * st.S, hw - 1) * (unsigned)C + c2), st.v0[b], st.v1[b]);
}
template <typename T>
__device__ __forceinline__ void gsf_spatial_sums(const GsfSums<T>& st, const T* __restrict__ x, long f, int hw, int C, int F,
                                                 const float* sg, float* part, float* __restrict__ ysum,
                                                 float* __restrict__ xsum) {
  const int Fh = F >> 1;
  const int S = st.S, s = st.s, cp = st.cp;
  if (st.act) {
    float y0 = 0.f, y1 = 0.f, x0 = 0.f, x1 = 0.f;
    const T* __restrict__ xf = x + f * hw * C;
    const unsigned c2 = 2u * (unsigned)cp;
    const int g = (2 * cp) >= Fh;
#pragma unroll
    for (int b = 0; b < GS_NPRE; ++b) {
      const int p = s + b * S;
      if (p < hw) {
        const float gt = sg[2 * p + g];
        x0 += st.v0[b]; x1 += st.v1[b];
        y0 += st.v0[b] * gt; y1 += st.v1[b] * gt;
      }
    }
    for (int p0 = s + GS_NPRE * S; p0 < hw; p0 += S * 8) {
      float v0[8], v1[8];
#pragma unroll
      for (int b = 0; b < 8; ++b) Pair<T>::load(xf + ((unsigned)min(p0 + b * S, hw - 1) * (unsigned)C + c2), v0[b], v1[b]);
#pragma unroll
      for (int b = 0; b < 8; ++b) {
        const int p = p0 + b * S;
        if (p < hw) {
          const float gt = sg[2 * p + g];
          x0 += v0[b]; x1 += v1[b];
          y0 += v0[b] * gt; y1 += v1[b] * gt;
        }
      }
    }
    part[s * F + 2 * cp] = y0;
    part[s * F + 2 * cp + 1] = y1;
    part[(S + s) * F + 2 * cp] = x0;
    part[(S + s) * F + 2 * cp + 1] = x1;
  }
  __syncthreads();
  for (int c = threadIdx.x; c < F; c += 256) {
    float ys = 0.f, xs = 0.f;
    for (int s2 = 0; s2 < S; ++s2) {
      ys += part[s2 * F + c];
      xs += part[(S + s2) * F + c];
    }
    ysum[f * F + c] = ys;
    xsum[f * F + c] = xs;
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
  const long f = xcd_logical_id(blockIdx.x, gridDim.x);
  const int t = (int)(f % T_len);
  float* sg = sm;
  float* part = sm + 2 * hw;
  const float has_prev = t > 0 ? 1.f : 0.f, has_next = t < T_len - 1 ? 1.f : 0.f;
  const long fp = t > 0 ? f - 1 : f, fn = t < T_len - 1 ? f + 1 : f;      // clamped: loads stay branch-free
  const float b0 = b3d[0], b1 = b3d[1];
  const float* __restrict__ Qc = Q + f * hw * 6;                // uniform bases of the three frames' tap maps
  const float* __restrict__ Qp = Q + fp * hw * 6;
  const float* __restrict__ Qn = Q + fn * hw * 6;
  float* __restrict__ gout = gate + f * hw * 2;
  // requests: the first batch of tap-map values, then the frame's values for the sums (they travel while the gates are made)
  float qc[4], qp[4], qn[4];
  auto q_issue = [&](int i0) {
#pragma unroll
    for (int b = 0; b < 4; ++b) {
      const unsigned i = (unsigned)min(i0 + b * 256, 2 * hw - 1);
      const unsigned o = (i >> 1) * 6u + (i & 1u);
      qc[b] = Qc[o + 2u];
      qp[b] = Qp[o];
      qn[b] = Qn[o + 4u];
    }
  };
  q_issue(threadIdx.x);
  GsfSums<T> st;
  gsf_sums_issue<T>(st, x, f, hw, C, F);
  TD_ISSUE_FENCE();
  for (int i0 = threadIdx.x; i0 < 2 * hw; i0 += 256 * 4) {
    if (i0 != (int)threadIdx.x) q_issue(i0);
#pragma unroll
    for (int b = 0; b < 4; ++b) {
      const int i = i0 + b * 256;
      if (i < 2 * hw) {
        const float v = tanhf(((i & 1) ? b1 : b0) + qc[b] + has_prev * qp[b] + has_next * qn[b]);
        sg[i] = v;
        gout[i] = v;
      }
    }
  }
  __syncthreads();
  gsf_spatial_sums<T>(st, x, f, hw, C, F, sg, part, ysum, xsum);
}

// (a merged form of launches 1a + 1b -- every frame staged three times, no Q maps -- was measured twice as slow and is parked:
// experiments/r4_parked/gsf_with_merge.hip)

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
  if (dtype == TDEED_BF16 && wqf && (F % 8 == 0 || F + 4 <= C)) {   // the staging reads whole 16-byte chunks
    const int nch = (F + 7) / 8;
    int ps16 = nch + 1;
    if ((ps16 & 1) == 0) ++ps16;
    const int PSQ = ps16 * 16, KSq = (9 * nch + 3) / 4;
    const bool wreg = KSq <= 28;                                  // weight fragments copied to registers per wave (F <= 96)
    const long wbytes = (long)KSq * 64 * 16;
    // LDS budget per workgroup: 52 KB = three workgroups per CU (measured: DESIGN §9); the register-weight form may take
    // 78 KB = two per CU when a frame does not fit 52 (800MF s3, F = 80: a whole 14 x 14 frame instead of two bands with
    // the weights staged twice)
    long q_cap = (wreg ? 78 : 52) * 1024;
    int bq = (int)((q_cap - wbytes - 64L * nch - 16L * KSq) / ((long)(w + 2) * PSQ)) - 2;
    if (bq < 1) {
      // wide slices (F = 196 of RegNetY-800MF s4: 58 KB of tap-weight fragments alone): up to 150 KB of LDS, one
      // workgroup per CU, instead of falling back to the vector-ALU tap kernel (118 vs ~35 us per site)
      q_cap = 150 * 1024;
      bq = (int)((q_cap - wbytes - 64L * nch - 16L * KSq) / ((long)(w + 2) * PSQ)) - 2;
    }
    if (bq >= 1) {
      static TdDevOnce attr_q;
      if (!attr_q.get()) {
        if (hipFuncSetAttribute((const void*)gsf_q_mfma_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024) !=
                hipSuccess ||
            hipFuncSetAttribute((const void*)gsf_q_mfma_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024) !=
                hipSuccess) {
          tdeed_set_error("gsf_gate: hipFuncSetAttribute failed");
          return TDEED_ERR_RUNTIME;
        }
        attr_q.set();
      }
      if (bq > h) bq = h;
      const int nbq = cdiv(h, bq);
      const size_t fbytes = (size_t)(bq + 2) * (w + 2) * PSQ, fixed = (size_t)wbytes + (size_t)8 * (nch * 8) + (size_t)16 * KSq;
      // frames per workgroup: 2 when one frame each does not fit the chip in ONE round of resident workgroups and two do
      // (cfg2: 800 frames against 3 x 256 resident workgroups).  Resident workgroups per CU: by LDS, and three by registers
      const long lds_cu = 160 * 1024, ncu = 256, reg_cap = 3;
      int nfr = 1;
      if (nbq == 1 && B * T > 1) {
        const long slots1 = std::min<long>(reg_cap, lds_cu / (long)(fixed + fbytes)) * ncu;
        const long per2 = (long)(fixed + 2 * fbytes);
        const long slots2 = per2 <= 150 * 1024 ? std::min<long>(reg_cap, lds_cu / per2) * ncu : 0;
        if ((long)B * T > slots1 && ((long)B * T + 1) / 2 <= slots2) nfr = 2;
      }
      const size_t smq = fixed + (size_t)nfr * fbytes;
      if (wreg)
        hipLaunchKernelGGL(gsf_q_mfma_kernel<true>, dim3(cdiv(B * T, nfr) * nbq), dim3(256), smq, st, (const bf16_t*)x, h, w, C,
                           F, bq, nch, PSQ, KSq, bn_scale, bn_shift, (const bf16x8*)wqf, Q, nbq, nfr, B * T);
      else
        hipLaunchKernelGGL(gsf_q_mfma_kernel<false>, dim3(cdiv(B * T, nfr) * nbq), dim3(256), smq, st, (const bf16_t*)x, h, w, C,
                           F, bq, nch, PSQ, KSq, bn_scale, bn_shift, (const bf16x8*)wqf, Q, nbq, nfr, B * T);
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

// launch 1b alone: the tap maps Q come from elsewhere (the tail of the one-launch bottleneck in front of the site,
// tdeed_bneck_gs_fwd).  bf16.
extern "C" int tdeed_gsf_gate_sums_fwd(const void* x, int B, int T, int h, int w, int C, int F, const float* b3d, const float* Q,
                                       float* gate, float* ysum, float* xsum, void* stream) {
  TD_CHECK(x && b3d && Q && gate && ysum && xsum, "gsf_gate_sums: null pointer");
  TD_CHECK(B > 0 && T > 0 && h > 0 && w > 0 && F > 0 && F % 4 == 0 && F <= C && F <= 256,
           "gsf_gate_sums: bad sizes B=%d T=%d h=%d w=%d C=%d F=%d", B, T, h, w, C, F);
  const int hw = h * w, S = 256 / (F / 2);
  const size_t smem2 = (size_t)(2 * hw + 2 * S * F) * sizeof(float);
  TD_CHECK(smem2 <= 64 * 1024, "gsf_gate_sums: frame too large for the gate/sum pass (%d px)", hw);
  hipLaunchKernelGGL(gsf_gate_sums_kernel<bf16_t>, dim3(B * T), dim3(256), smem2, (hipStream_t)stream, (const bf16_t*)x, Q, T, hw,
                     C, F, b3d, gate, ysum, xsum);
  TD_LAUNCH_CHECK("gsf_gate_sums");
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

// gsf_apply_kernel for bf16 slices whose fold is a multiple of 16 (RegNetY-800MF: F = 80, 192): a thread makes 8 consecutive
// output channels = the interleave of two runs of 4 consecutive SOURCE channels (cols 2j + i <- i * F/4 + j), so the frame and its
// temporal neighbour are read as 8-byte pieces and the result leaves as one 16-byte store (the scalar form: eight 2-byte
// gathers and an 8-byte store per 4 channels, 77 us per site at cfg3 against ~40 here).  Same arithmetic, same rounding.
__global__ __launch_bounds__(256) void gsf_apply_vec8_kernel(const bf16_t* __restrict__ x, const float* __restrict__ gate,
                                                             const float* __restrict__ fw, int T_len, int hw, int C, int F,
                                                             int Fp, bf16_t* __restrict__ out, long total) {
  const int Fh = F >> 1, Fq = F >> 2;
  const int cpr = Fp >> 3;             // 8-channel chunks per pixel
  for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
    const int k = (int)(idx % cpr);
    const long pix = idx / cpr;
    const long f = pix / hw;
    const int t = (int)(f % T_len);
    const long b = f / T_len;
    const int co0 = k * 8;
    bf16_t* dst = out + pix * Fp + co0;
    if (co0 >= F) {                                              // pass-through padding columns [F, Fp)
      *reinterpret_cast<u32x4*>(dst) = *reinterpret_cast<const u32x4*>(x + pix * C + co0);
      continue;
    }
    const int g = co0 >= Fh, j0 = (co0 - g * Fh) >> 1;
    const int c_lo = g * Fh + j0, c_hi = c_lo + Fq;              // source runs of i = 0 and i = 1
    const int ts = g ? t - 1 : t + 1;
    const bool has = ts >= 0 && ts < T_len;
    const long pix2 = has ? pix + (long)(ts - t) * hw : pix;
    const bf16x4 lo = *reinterpret_cast<const bf16x4*>(x + pix * C + c_lo), hi = *reinterpret_cast<const bf16x4*>(x + pix * C + c_hi);
    const bf16x4 lo2 = *reinterpret_cast<const bf16x4*>(x + pix2 * C + c_lo), hi2 = *reinterpret_cast<const bf16x4*>(x + pix2 * C + c_hi);
    const float gt = gate[pix * 2 + g], gt2 = has ? gate[pix2 * 2 + g] : 0.f;
    float wl[4], wh[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      wl[e] = fw ? fw[(b * F + c_lo + e) * T_len + t] : 0.f;
      wh[e] = fw ? fw[(b * F + c_hi + e) * T_len + t] : 0.f;
    }
    bf16x8 o;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int jj = e >> 1, i = e & 1;
      const float xv = i ? (float)hi[jj] : (float)lo[jj];
      const float x2 = i ? (float)hi2[jj] : (float)lo2[jj];
      const float r = xv - gt * xv;
      const float ysh = has ? gt2 * x2 : 0.f;
      const float wv = i ? wh[jj] : wl[jj];
      o[e] = (bf16_t)(fw ? ysh * wv + r * (1.0f - wv) : ysh + r);
    }
    *reinterpret_cast<bf16x8*>(dst) = o;
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

// bf16 throughput form of the same launch, built around loads in flight: the frame's channels [0,Fp), the
// channels of its two temporal neighbours, the three gate maps and the five rows of spatial sums the fusion conv
// touches are all fetched in wide batches into LDS (clamped addresses, no branches around loads); the blend and the
// channel interleave then run out of LDS and leave as 8-byte stores.
// GA_UX: 8-byte pieces per thread and batch.  A chunk of pixels is sized by the launcher to ONE batch (256 * GA_UX pieces,
// 512 pixels): every further batch is one more exposed memory round trip in a launch that is nothing but round trips
// (7x7x96: 1176 pieces were two batches of 4 per thread, 28x28x16: three chunks of 1048 pieces = six round trips).
constexpr int GA_UX = 8;
__global__ __launch_bounds__(256) void gsf_apply_fused_bf16_kernel(const bf16_t* __restrict__ x,
                                                                   const float* __restrict__ gate,
                                                                   const float* __restrict__ ysum,
                                                                   const float* __restrict__ xsum, float inv_hw,
                                                                   const float* __restrict__ cw1,
                                                                   const float* __restrict__ cb1,
                                                                   const float* __restrict__ cw2,
                                                                   const float* __restrict__ cb2, int T_len, int hw,
                                                                   int C, int F, int Fp, int pchunk,
                                                                   bf16_t* __restrict__ out) {
  extern __shared__ __attribute__((aligned(16))) unsigned char sma[];
  const long f = xcd_logical_id(blockIdx.x, gridDim.x);
  const int t = (int)(f % T_len);
  const long b = f / T_len;
  const int Fh = F >> 1, Fq = F >> 2, tid = threadIdx.x;
  float* ssum = reinterpret_cast<float*>(sma);                 // [2][5][F]: ysum, xsum of frames t-2 .. t+2 (0 outside)
  float* fwl = ssum + 10 * F;                                   // [F] fusion weight, indexed by source channel
  float* cwl = fwl + F;                                         // [2][18] conv weights, [2] biases
  float* gc = cwl + 40;                                         // [pchunk][2] gates of this frame
  float* gs = gc + 2 * pchunk;                                  // [pchunk][2] gate 0 of frame t+1 | gate 1 of frame t-1
  bf16_t* xc = reinterpret_cast<bf16_t*>(gs + 2 * pchunk);      // [pchunk][Fp] this frame
  bf16_t* xs = xc + (size_t)pchunk * Fp;                        // [pchunk][Fp] c < Fh: frame t+1, else frame t-1
  const bool has_next = t < T_len - 1, has_prev = t > 0;
  const long fn = has_next ? f + 1 : f, fp = has_prev ? f - 1 : f;
  const int npc = Fp >> 2;                                      // 4-channel pieces per pixel
  // ---- issue first: spatial sums of the 5 frames around t (F <= 256 -> at most 10 per lane), conv weights
  const IDiv dF(F), dnpc(npc);
  float sv[10];
#pragma unroll
  for (int u = 0; u < 10; ++u) {
    const int i = min(tid + u * 256, 10 * F - 1);
    int row, c;                                                 // row = arr * 5 + r
    dF.divmod(i, row, c);
    const int arr = row >= 5, r = row - 5 * arr;
    const int t2 = min(max(t + r - 2, 0), T_len - 1);
    sv[u] = (arr ? xsum : ysum)[(b * T_len + t2) * F + c];
  }
  const float cwv = *(tid < 18 ? cw1 + tid : tid < 36 ? cw2 + (tid - 18) : tid == 36 ? cb1 : cb2);
  f32x2 ga[2];
  float gn_[2], gp_[2];     // only gate 0 of frame t+1 and gate 1 of frame t-1 are needed (a half-used wide load
                            // leaves a dead register that gets reused while the load is in flight: WAW stall)
  u32x2 vc[GA_UX], vn[GA_UX], vp[GA_UX];
  // gates (one float2 per pixel from each of the three frames) and activations (8-byte pieces, three frames)
  auto issue = [&](int p0, int pn, int total, int it) {
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int p = p0 + min(it * 512 + tid + u * 256, pn - 1);
      ga[u] = *reinterpret_cast<const f32x2*>(gate + (f * hw + p) * 2);
      gn_[u] = gate[(fn * hw + p) * 2];
      gp_[u] = gate[(fp * hw + p) * 2 + 1];
    }
#pragma unroll
    for (int u = 0; u < GA_UX; ++u) {
      const int i = min(it * (256 * GA_UX) + tid + u * 256, total - 1);
      int pj, pq;
      dnpc.divmod(i, pq, pj);
      const long off = (long)(p0 + pq) * C + pj * 4;
      vc[u] = *reinterpret_cast<const u32x2*>(x + f * hw * C + off);
      // a piece's channels below Fh are shifted in from frame t+1, the others from frame t-1: only the piece that straddles
      // Fh (F / 2 not a multiple of 4) needs both neighbours
      vn[u] = (u32x2){0u, 0u};
      vp[u] = (u32x2){0u, 0u};
      if (pj * 4 < Fh) vn[u] = *reinterpret_cast<const u32x2*>(x + fn * hw * C + off);
      if (pj * 4 + 3 >= Fh && pj * 4 < F) vp[u] = *reinterpret_cast<const u32x2*>(x + fp * hw * C + off);
    }
  };
  auto commit = [&](int pn, int total, int it) {
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int i = it * 512 + tid + u * 256;
      if (i < pn) {
        gc[2 * i] = ga[u][0];
        gc[2 * i + 1] = ga[u][1];
        gs[2 * i] = has_next ? gn_[u] : 0.f;
        gs[2 * i + 1] = has_prev ? gp_[u] : 0.f;
      }
    }
#pragma unroll
    for (int u = 0; u < GA_UX; ++u) {
      const int i = it * (256 * GA_UX) + tid + u * 256;
      if (i < total) {
        int pj, pl_;
        dnpc.divmod(i, pl_, pj);
        *reinterpret_cast<u32x2*>(xc + (long)pl_ * Fp + pj * 4) = vc[u];
        const bf16x4 n4 = *reinterpret_cast<const bf16x4*>(&vn[u]);
        const bf16x4 p4 = *reinterpret_cast<const bf16x4*>(&vp[u]);
        bf16x4 o4;
#pragma unroll
        for (int e = 0; e < 4; ++e) o4[e] = (pj * 4 + e < Fh) ? n4[e] : p4[e];
        *reinterpret_cast<bf16x4*>(xs + (long)pl_ * Fp + pj * 4) = o4;
      }
    }
  };
  // the first batch of the first chunk travels together with the sums: one memory round trip for everything
  {
    const int pn = min(pchunk, hw), total = pn * npc;
    issue(0, pn, total, 0);
    TD_ISSUE_FENCE();
#pragma unroll
    for (int u = 0; u < 10; ++u) {
      const int i = tid + u * 256;
      if (i < 10 * F) {
        const int row = dF.div(i);
        const int t2 = t + (row >= 5 ? row - 5 : row) - 2;
        ssum[i] = (t2 >= 0 && t2 < T_len) ? sv[u] : 0.f;
      }
    }
    if (tid < 38) cwl[tid] = cwv;
    commit(pn, total, 0);
  }
  for (int p0 = 0; p0 < hw; p0 += pchunk) {
    const int pn = min(pchunk, hw - p0);
    const int total = pn * npc;
    if (p0 > 0) __syncthreads();
    for (int it = p0 == 0 ? 1 : 0; it * (256 * GA_UX) < total || it * 512 < pn; ++it) {
      issue(p0, pn, total, it);
      TD_ISSUE_FENCE();
      commit(pn, total, it);
    }
    __syncthreads();
    if (p0 == 0) {
      // fusion weights of this frame: 3x3 conv over the (channel, time) plane of the spatial means + sigmoid
      for (int c = tid; c < F; c += 256) {
        const int g = c >= Fh;
        const int cl = c - g * Fh;
        const float* cw = cwl + 18 * g;
        float a = cwl[36 + g];
#pragma unroll
        for (int dc = -1; dc <= 1; ++dc) {
          const int c2 = cl + dc;
          if (c2 < 0 || c2 >= Fh) continue;
          const int cc = g * Fh + c2;
#pragma unroll
          for (int dt = -1; dt <= 1; ++dt) {
            const int t2 = t + dt;
            if (t2 < 0 || t2 >= T_len) continue;
            const float rm = (ssum[(5 + dt + 2) * F + cc] - ssum[(dt + 2) * F + cc]) * inv_hw;
            const int r2 = dt + 2 + (g ? -1 : 1);                 // row of the shifted source frame (zeros outside the clip)
            const float ysh = ssum[r2 * F + cc] * inv_hw;
            a = fmaf(cw[(dc + 1) * 3 + (dt + 1)], ysh, a);
            a = fmaf(cw[9 + (dc + 1) * 3 + (dt + 1)], rm, a);
          }
        }
        fwl[c] = sigmoidf_(a);
      }
      __syncthreads();
    }
    for (int idx = tid; idx < total; idx += 256) {
      int qd, pl_;
      dnpc.divmod(idx, pl_, qd);
      const bf16_t* xr = xc + (long)pl_ * Fp;
      const bf16_t* sr = xs + (long)pl_ * Fp;
      bf16x4 o;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int co = qd * 4 + e;
        if (co >= F) { o[e] = xr[co]; continue; }
        const int g = co >= Fh;
        const int col = co - g * Fh;
        const int ci = g * Fh + (col & 1) * Fq + (col >> 1);
        const float xv = (float)xr[ci];
        // (explicit fused multiply-adds: bneck.hip computes the same expression while it loads a block's frames, and
        //  -ffp-contract=fast would otherwise be free to contract the two copies differently)
        const float r = fmaf(-gc[2 * pl_ + g], xv, xv);
        const float ysh = gs[2 * pl_ + g] * (float)sr[ci];
        const float wv = fwl[ci];
        o[e] = (bf16_t)fmaf(ysh, wv, r * (1.0f - wv));
      }
      *reinterpret_cast<bf16x4*>(out + (f * hw + p0 + pl_) * Fp + qd * 4) = o;
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
  else if (dtype == TDEED_BF16) {
    // pixels per LDS chunk: (2 gate pairs fp32 + 2 x Fp bf16) per pixel next to the fixed tables; a chunk is at most one
    // batch of loads (256 * GA_UX pieces, 512 pixels) and at most 36 KB of LDS ( four workgroups per CU,
    // 1024 resident workgroups for the 800 frames of a batch); equal chunks
    const long fixed = (long)(11 * F + 40) * sizeof(float);
    constexpr long a_kb = 36;
    long pchunk = (a_kb * 1024 - fixed) / (16 + 4L * Fp);
    if (pchunk < 1) pchunk = (60 * 1024 - fixed) / (16 + 4L * Fp);
    TD_CHECK(pchunk >= 1 && F <= 256, "gsf_apply_fused: fold %d too wide", F);
    pchunk = std::min<long>(pchunk, std::min<long>(512, (256L * GA_UX) / (Fp >> 2)));
    if (pchunk > hw) pchunk = hw;
    pchunk = (hw + (hw + pchunk - 1) / pchunk - 1) / ((hw + pchunk - 1) / pchunk);      // equal chunks
    const size_t smb = (size_t)fixed + (size_t)pchunk * (16 + 4 * Fp);
    hipLaunchKernelGGL(gsf_apply_fused_bf16_kernel, dim3(B * T), dim3(256), smb, st, (const bf16_t*)x, gate, ysum, xsum,
                       1.0f / (float)hw, cw1, cb1, cw2, cb2, T, hw, C, F, Fp, (int)pchunk, (bf16_t*)out);
  }
  else { tdeed_set_error("gsf_apply_fused: bad dtype %d", dtype); return TDEED_ERR_ARG; }
  TD_LAUNCH_CHECK("gsf_apply_fused");
  return TDEED_OK;
}

// ---- the same blend with the output left in SOURCE channel order (no `c = i*(F/4)+j -> 2j+i` interleave): for callers that
// fold the interleave into the weight columns of the 1x1 conv consuming the slice (engine.py permutes conv1's first F columns).
// Without the gather nothing needs LDS but the F fusion weights: a thread's 8-byte pieces of this frame, the matching pieces of
// the neighbour frame its channels shift in from (t+1 below F/2, t-1 above; the piece straddling F/2 takes both), the three
// gate values of its pixel and the spatial sums are requested together -- ONE memory round trip per workgroup -- and the blend
// runs out of registers (the interleaving launch staged three frames' pieces in LDS and gathered 2-byte elements from them:
// ~20 LDS reads per piece).  Arithmetic = gsf_apply_fused_bf16_kernel's, bit for bit.
// Instruction count matters as much as the round trip here: time stamps showed 3.9 us of a workgroup's 7.9 us between its start
// and the END OF ISSUING its loads -- a divmod per piece and per sum, 64-bit address arithmetic per load.  Hence: every
// address is a workgroup-uniform base (frame t-1 of the slice / the gate map: scalar registers) plus an unsigned 32-bit lane
// offset, a thread's pieces step by 256 through (pixel, piece-of-pixel) with adds instead of divisions, the spatial sums are
// read row by row by thread c < F.
constexpr int GB_UX = 8;          // pieces per thread and batch
__global__ __launch_bounds__(256) void gsf_blend_src_kernel(const bf16_t* __restrict__ x, const float* __restrict__ gate,
                                                            const float* __restrict__ ysum, const float* __restrict__ xsum,
                                                            float inv_hw, const float* __restrict__ cw1,
                                                            const float* __restrict__ cb1, const float* __restrict__ cw2,
                                                            const float* __restrict__ cb2, int T_len, int hw, int C, int F,
                                                            int Fp, bf16_t* __restrict__ out) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smb[];
  const long f = xcd_logical_id(blockIdx.x, gridDim.x);
  const int t = (int)(f % T_len);
  const long b = f / T_len;
  const int Fh = F >> 1, tid = threadIdx.x;
  float* ssum = reinterpret_cast<float*>(smb);                 // [2][5][F]: ysum, xsum of frames t-2 .. t+2 (0 outside)
  float* fwl = ssum + 10 * F;                                   // [Fp] fusion weight by source channel (pad channels: unused)
  float* cwl = fwl + Fp;                                        // [2][18] conv weights, [2] biases
  const bool has_next = t < T_len - 1, has_prev = t > 0;
  const long fp = has_prev ? f - 1 : f;
  const int npc = Fp >> 2, total = hw * npc;
  const bool strad = (Fh & 3) != 0;                             // a piece holds channels of both gate groups
  // uniform bases (frame t-1, or t at a clip's first frame) and the element offsets of frames t / t+1 from them
  const bf16_t* __restrict__ xb = x + fp * hw * C;
  const float* __restrict__ gb = gate + fp * hw * 2;
  bf16_t* __restrict__ ob = out + f * hw * Fp;
  const unsigned oc = has_prev ? (unsigned)hw : 0u;            // this frame, in pixels from the base
  const unsigned on = oc + (has_next ? (unsigned)hw : 0u);     // frame t+1 (clamped to t at a clip's last frame)
  // ---- requests: sums + conv weights, then the first batch of pieces and gates
  float sv[10];
#pragma unroll
  for (int row = 0; row < 10; ++row) {                          // row = arr * 5 + r: thread c < F reads channel c of every row
    const int arr = row >= 5, r = row - 5 * arr;
    const int t2 = min(max(t + r - 2, 0), T_len - 1);
    sv[row] = 0.f;
    if (tid < F) sv[row] = (arr ? xsum : ysum)[(b * T_len + t2) * F + tid];
  }
  const float cwv = *(tid < 18 ? cw1 + tid : tid < 36 ? cw2 + (tid - 18) : tid == 36 ? cb1 : cb2);
  u32x2 vc[GB_UX], vs[GB_UX], vs2[GB_UX];
  f32x2 ga[GB_UX];
  float gnb[GB_UX], gnb2[GB_UX];
  int pa[GB_UX], pja[GB_UX];                                    // (pixel, piece of the pixel) of each of the thread's pieces
  const IDiv dnpc(npc);
  const int q256 = 256 / npc, r256 = 256 - q256 * npc;          // a step of 256 pieces in (pixel, piece) terms
  auto issue = [&](int it) {
    int p, pj;
    dnpc.divmod(it * (256 * GB_UX) + tid, p, pj);
#pragma unroll
    for (int u = 0; u < GB_UX; ++u) {
      pa[u] = p;
      pja[u] = pj;
      const unsigned pp = (unsigned)min(p, hw - 1);            // (pieces beyond the frame: clamped addresses, never blended)
      const unsigned off = pp * (unsigned)C + (unsigned)pj * 4u;
      const bool lo = pj * 4 < Fh;                               // first channel of the piece in gate group 0
      vc[u] = *reinterpret_cast<const u32x2*>(xb + (oc * (unsigned)C + off));
      vs[u] = *reinterpret_cast<const u32x2*>(xb + ((lo ? on : 0u) * (unsigned)C + off));
      ga[u] = *reinterpret_cast<const f32x2*>(gb + ((oc + pp) * 2u));
      gnb[u] = gb[(lo ? on + pp : pp) * 2u + (lo ? 0u : 1u)];
      vs2[u] = (u32x2){0u, 0u};
      gnb2[u] = 0.f;
      if (strad && lo && pj * 4 + 3 >= Fh) {                     // the straddling piece: its upper channels come from t-1
        vs2[u] = *reinterpret_cast<const u32x2*>(xb + off);
        gnb2[u] = gb[pp * 2u + 1u];
      }
      pj += r256;
      p += q256;
      if (pj >= npc) { pj -= npc; ++p; }
    }
  };
  auto blend = [&](int it) {
#pragma unroll
    for (int u = 0; u < GB_UX; ++u) {
      const int p = pa[u], pj = pja[u];
      if (p >= hw) continue;
      const bf16x4 xc4 = *reinterpret_cast<const bf16x4*>(&vc[u]);
      const bf16x4 s4 = *reinterpret_cast<const bf16x4*>(&vs[u]);
      const bf16x4 t4 = *reinterpret_cast<const bf16x4*>(&vs2[u]);
      const bool lo = pj * 4 < Fh;
      const f32x4 w4 = *reinterpret_cast<const f32x4*>(fwl + pj * 4);
      bf16x4 o;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int ci = pj * 4 + e;
        if (ci >= F) { o[e] = xc4[e]; continue; }
        const int g = ci >= Fh;
        const bool second = lo && g;                             // upper channel of the straddling piece
        const float xv = (float)xc4[e];
        const float r = fmaf(-(g ? ga[u][1] : ga[u][0]), xv, xv);
        float gsh = second ? gnb2[u] : gnb[u];
        gsh = (g ? has_prev : has_next) ? gsh : 0.f;
        const float ysh = gsh * (float)(second ? t4[e] : s4[e]);
        const float wv = w4[e];
        o[e] = (bf16_t)fmaf(ysh, wv, r * (1.0f - wv));
      }
      *reinterpret_cast<bf16x4*>(ob + ((unsigned)p * (unsigned)Fp + (unsigned)pj * 4u)) = o;
    }
  };
  issue(0);
  TD_ISSUE_FENCE();
  if (tid < F) {
#pragma unroll
    for (int row = 0; row < 10; ++row) {
      const int t2 = t + (row >= 5 ? row - 5 : row) - 2;
      ssum[row * F + tid] = (t2 >= 0 && t2 < T_len) ? sv[row] : 0.f;
    }
  }
  if (tid < 38) cwl[tid] = cwv;
  __syncthreads();
  // fusion weights of this frame: 3x3 conv over the (channel, time) plane of the spatial means + sigmoid
  for (int c = tid; c < Fp; c += 256) {
    float wgt = 0.f;
    if (c < F) {
      const int g = c >= Fh;
      const int cl = c - g * Fh;
      const float* cw = cwl + 18 * g;
      float a = cwl[36 + g];
#pragma unroll
      for (int dc = -1; dc <= 1; ++dc) {
        const int c2 = cl + dc;
        if (c2 < 0 || c2 >= Fh) continue;
        const int cc = g * Fh + c2;
#pragma unroll
        for (int dt = -1; dt <= 1; ++dt) {
          const int t2 = t + dt;
          if (t2 < 0 || t2 >= T_len) continue;
          const float rm = (ssum[(5 + dt + 2) * F + cc] - ssum[(dt + 2) * F + cc]) * inv_hw;
          const int r2 = dt + 2 + (g ? -1 : 1);                 // row of the shifted source frame (zeros outside the clip)
          const float ysh = ssum[r2 * F + cc] * inv_hw;
          a = fmaf(cw[(dc + 1) * 3 + (dt + 1)], ysh, a);
          a = fmaf(cw[9 + (dc + 1) * 3 + (dt + 1)], rm, a);
        }
      }
      wgt = sigmoidf_(a);
    }
    fwl[c] = wgt;
  }
  __syncthreads();
  blend(0);
  for (int it = 1; it * (256 * GB_UX) < total; ++it) {
    issue(it);
    TD_ISSUE_FENCE();
    blend(it);
  }
}

extern "C" int tdeed_gsf_blend_src_fwd(const void* x, const float* gate, const float* ysum, const float* xsum,
                                       const float* cw1, const float* cb1, const float* cw2, const float* cb2,
                                       int B, int T, int h, int w, int C, int F, int Fp, void* out, int dtype,
                                       void* stream) {
  TD_CHECK(x && gate && ysum && xsum && cw1 && cb1 && cw2 && cb2 && out, "gsf_blend_src: null pointer");
  TD_CHECK(F % 4 == 0 && Fp % 8 == 0 && Fp >= F && Fp <= C && F <= 256, "gsf_blend_src: bad fold F=%d Fp=%d C=%d", F, Fp, C);
  TD_CHECK(B > 0 && T > 0 && h > 0 && w > 0 && (long)B * T <= 0x7fffffffL, "gsf_blend_src: bad sizes");
  TD_CHECK(dtype == TDEED_BF16, "gsf_blend_src: bf16 only (dtype %d)", dtype);
  TD_CHECK(C % 4 == 0, "gsf_blend_src: rows of %d channels are not 8-byte pieces", C);
  const int hw = h * w;
  const size_t smem = (size_t)(10 * F + Fp + 40) * sizeof(float);
  hipLaunchKernelGGL(gsf_blend_src_kernel, dim3(B * T), dim3(256), smem, (hipStream_t)stream, (const bf16_t*)x, gate, ysum,
                     xsum, 1.0f / (float)hw, cw1, cb1, cw2, cb2, T, hw, C, F, Fp, (bf16_t*)out);
  TD_LAUNCH_CHECK("gsf_blend_src");
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
  else if (dtype == TDEED_BF16 && F % 16 == 0 && C % 8 == 0) {
    const long tot8 = (long)B * T * hw * (Fp / 8);
    const int g8 = (int)((tot8 + 255) / 256 < 8192 ? (tot8 + 255) / 256 : 8192);
    hipLaunchKernelGGL(gsf_apply_vec8_kernel, dim3(g8), dim3(256), 0, st, (const bf16_t*)x, gate, fw, T, hw, C, F, Fp,
                       (bf16_t*)out, tot8);
  }
  else if (dtype == TDEED_BF16)
    hipLaunchKernelGGL(gsf_apply_kernel<bf16_t>, dim3(grid), dim3(256), 0, st, (const bf16_t*)x, gate, fw, T, hw, C,
                       F, Fp, (bf16_t*)out, total);
  else { tdeed_set_error("gsf_apply: bad dtype %d", dtype); return TDEED_ERR_ARG; }
  TD_LAUNCH_CHECK("gsf_apply");
  return TDEED_OK;
}
