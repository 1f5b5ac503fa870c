// A whole stride-1 RegNetY bottleneck with identity shortcut on a SMALL map in one launch (bf16 throughput mode):
//   conv1 1x1 (+ gate-shift splice) + BN + ReLU -> conv2 grouped 3x3 + BN + ReLU -> SE squeeze / excite
//   -> conv3 1x1 on y2 * gate + BN + residual + ReLU          (timm Bottleneck.forward)
// for the blocks whose frames fit LDS twice over: s4.b2..b7 (7 x 7 x 368, TWO frames per workgroup) and s3.b2..b4
// (14 x 14 x 152, one frame) of RegNetY-200MF.  The launch-per-layer chain moves x, y1, y2 through HBM / L2 seven times
// in four latency-bound launches (80 us per s4 block for 29 MB); here a workgroup's frames stay in LDS -- region A holds
// x (gate-shift columns spliced in), later y2; region B holds y1, later the SE scratch, later the output rows -- and the
// weights stream from L2 as MFMA A-operand fragments, one 16-channel tile per wave at a time, the next tile's fragments
// requested while the current one is multiplied.  Two frames per workgroup halve the weight bytes per frame (the round-1
// one-frame form, experiments/, re-streamed ~0.7 MB per frame and lost); 8 waves keep two waves per SIMD.
// Rounding points are those of the chain (y1, y2, y2 * gate, out in bf16; everything else fp32, same k order); only the
// squeeze sums of a workgroup's SECOND frame associate differently (its pixels sit in other lanes of the tiles), i.e. the
// result equals gemm -> gconv3x3 -> se_gate_mfma -> gemm up to the last bit of those fp32 sums.
#include "common.h"
#include "se_excite.h"

struct BneckP {
  const bf16_t* x; const bf16_t* G; int Fp;       // block input [N][hw][C]; gate-shift splice [N * hw][Fp] (or null)
  const bf16x8* w1f; const float* s1; const float* h1;        // [NT][KS][64]
  const bf16x8* w2f; const float* s2; const float* h2;        // [NT4][5][64]  (engine.pack_gconv_frags)
  SeP se;                                                     // weights / biases / R (pooled, n_parts, C filled in-kernel)
  const bf16x8* w3f; const float* s3; const float* h3;        // [NT][KS][64]
  bf16_t* out; bf16_t* out2; int n2;              // out [N][hw][C]; optional compact copy of channels [0, n2)
  int N, h, w, C;
  int w2_tap_major;                               // k-slot order of w2f: 0 = half * 9 + tap (gconv3x3_mfma_kernel's), 1 = 2 * tap + half
  long long* dbg;                                 // diagnostic: per-workgroup phase time stamps (or null)
  // gate-shift-fuse blend inside the frame load (gs.gate != null; G unused): what tdeed_gsf_blend_src_fwd would have written
  // into G is made here from the gate maps and spatial sums of tdeed_gsf_gate_fwd -- one launch and one round trip of the
  // slice through memory less per block
  struct Gs {
    const bf16_t* x; int ldx;                     // the slice's source: block input or its compact copy, row stride in elements
    const float* gate; const float* ysum; const float* xsum;   // [N][hw][2], [N][F], [N][F]
    const float* cw1; const float* cb1; const float* cw2; const float* cb2;
    int T_len, F;
  } gs;
  // tap maps of the NEXT block's gate-shift site (qt.Q != null): what tdeed_gsf_gate_fwd's first launch (gsf_q_mfma_kernel) would
  // compute from this block's output, made from the output rows while they are still in LDS -- that launch is gone
  struct Qt {
    const bf16x8* wqf;                            // [4][ceil(nch / 4)][64] fragments (engine.pack_gsf_p_frags), nch = ceil(F / 8)
    const float* bn;                              // [2][8 * nch]: folded BatchNorm3d scale | shift of the site, zeros behind channel F
    float* Q;                                     // [N][hw][6]
    int F;
  } qt;
};

#define BN_STAMP(i) do { if (p.dbg && threadIdx.x == 0) p.dbg[(long)blockIdx.x * 16 + (i)] = clock64(); } while (0)

constexpr int BNK_NW = 8, BNK_THR = BNK_NW * 64;

// activation row stride in bytes: C * 2 rounded up to 96 mod 128.  The 16 pixel rows x 4 k-chunks of an MFMA operand read
// (ds_read_b128, lane groups {0-3,12-15,20-27}, ...) then fall on 16 distinct 16-byte bank slots per group: 4.0 LDS cycles
// per read against 7.7 with a plain 16-byte skew (C = 368: stride 736, no pad at all; C = 152: 352)
__host__ __device__ inline int bneck_rs(int C) { return C * 2 + ((96 - (C * 2) % 128) + 128) % 128; }

__host__ __device__ static inline size_t bneck_smem(int fpw, int hw, int C) {
  const size_t RS = (size_t)bneck_rs(C);
  return 2 * (size_t)fpw * hw * RS + 2 * RS + 64 + (size_t)2 * fpw * C * 4 + (size_t)6 * ((C + 15) / 16 * 16) * 4;
}

// sum over each row of 16 lanes (the 16 pixels of an MFMA tile), every lane gets it: DPP moves, the pairing of a
// shuffle-xor butterfly (bit-identical to it), none of its four LDS crossbar round trips
__device__ __forceinline__ float bnk_row16_sum(float v) {
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));   // lane ^ 1
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true));   // lane ^ 2
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xF, 0xF, true));  // other quad
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x140, 0xF, 0xF, true));  // other half
  return v;
}

// FPW frames per workgroup, NPTM >= ceil(FPW * hw / 16) pixel tiles
// TAPM: conv2's k-slot order (BneckP::w2_tap_major) as a compile-time constant -- as a run-time select the same addresses
// cost the conv2 phase its whole gain (186 vs 175 stamp units: measured)
// BLEND: the gate-shift-fuse blend inside the frame load (p.gs) -- a template parameter so that either form of the load phase is
// straight-line code (as a run-time branch the requests of the two forms met at joins, and waits there are conservative)
template <int KS, int FPW, int NPTM, bool TAPM, bool BLEND>
__global__ __launch_bounds__(BNK_THR, 1) void bneck_kernel(const BneckP p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int hw = p.h * p.w, C = p.C;
  // frames in contiguous chunks per XCD, the numbering of the gate-shift launches in front of this one: the spliced columns
  // (and most of x) arrive through the L2 they were written through
  const int f0 = (int)xcd_logical_id(blockIdx.x, gridDim.x) * FPW;
  const int nfr = min(FPW, p.N - f0);                    // frames of this workgroup
  const int npix = nfr * hw;
  const int RS = bneck_rs(C);                            // activation row stride (bytes)
  const int NT = (C + 15) >> 4;
  unsigned char* At = smem;                              // [FPW * hw][RS]   x, later y2
  unsigned char* Bt = At + FPW * hw * RS;                // [FPW * hw][RS]   y1, later SE scratch, later the output rows
  unsigned char* Zr = Bt + FPW * hw * RS;                // one row of zeros (taps outside the map) + slack for the k pad
  unsigned char* Tr = Zr + RS + 64;                      // one row nobody reads: where the lanes of a ragged last tile store
  float* pooled = reinterpret_cast<float*>(Tr + RS);     // [FPW][C] squeeze sums
  float* gtab = pooled + FPW * C;                        // [FPW][C] gates
  float* bnv = gtab + FPW * C;                           // [6][NT * 16]: folded BatchNorm scale / shift of conv1, conv2, conv3
  const int CP = ((C + 15) >> 4) << 4;
  const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int pl = lane & 15, q = lane >> 4;
  const bf16_t* xg = p.x + (long)f0 * hw * C;
  [[maybe_unused]] const long lds_bytes = (long)bneck_smem(FPW, hw, C);     // what the launch asked for (debug flavour checks)
  TD_DEV_ASSERT(nfr >= 1 && (npix + 15) / 16 <= NPTM && (C + 31) / 32 == KS);
  TD_LDS_CHECK((unsigned char*)(bnv + 6 * CP) - smem, 0, lds_bytes);

  // The first channel tile's weights of each contraction are requested one phase early (conv1's before the frames are
  // loaded, conv2's before conv1 runs, conv3's before the SE phase): a phase otherwise opens with a bare L2 round trip.
  const int NT_ = (C + 15) >> 4;
  // PAIR (the 368-wide form, >= 16 channel tiles): a wave multiplies TWO of its channel tiles (wv, wv + 8) against each pixel
  // fragment it reads, in two half-K blocks of 2 x KS/2 weight fragments -- the activation reads of conv1 / conv3, which bound
  // those phases (a fragment read per MFMA: 7.9 k LDS cycles per contraction against 4.0 k of MFMA issue), fall by a third
  // (the wave's third tile runs alone).  Same k order per output: bit-identical results.  Register cost: 7 more accumulators.
  constexpr bool PAIR = KS >= 8 && KS % 2 == 0;
  constexpr int KH = KS / 2;
  // TRIPLE (KS = 12): ALL THREE of a wave's channel tiles (wv, wv + 8, wv + 16) against each activation fragment, in three
  // third-K blocks of 3 x KS/3 weight fragments (the same 2 x KS fragment registers as PAIR, 7 more accumulators): one LDS
  // fragment read per three MFMAs.  conv1 / conv3 at the pair form were bound by exactly those reads (8 waves x 168 reads x
  // 8 LDS cycles = 10.7 k cycles of a 14.2 k-cycle phase against 8 k cycles of MFMA issue).  A wave without a third tile
  // (wv = 7 at 23 tiles) multiplies a clamped one and drops it.  Same k order per output: bit-identical results.
  constexpr bool TRIPLE = PAIR && KS % 3 == 0;
  constexpr int KT = KS / 3;
  bf16x8 wc1[KS], wf2[5];
  auto load_block3 = [&](bf16x8 (&dst)[KS], const bf16x8* __restrict__ wsrc, int ta, int tb, int tc, int ks0) {
#pragma unroll
    for (int k = 0; k < KT; ++k) {
      dst[k] = wsrc[((long)ta * KS + ks0 + k) * 64 + lane];
      dst[KT + k] = wsrc[((long)tb * KS + ks0 + k) * 64 + lane];
      dst[2 * KT + k] = wsrc[((long)tc * KS + ks0 + k) * 64 + lane];
    }
  };
  const int T3A = wv, T3B = wv + BNK_NW, T3C = min(wv + 2 * BNK_NW, NT_ - 1);
  const bool t3c_ok = wv + 2 * BNK_NW < NT_;
  auto load_block = [&](bf16x8 (&dst)[KS], const bf16x8* __restrict__ wsrc, int ta, int tb, int ks0) {
    // dst[0 .. KH) = tile ta, k-steps ks0 .. ks0 + KH; dst[KH .. KS) = tile tb, the same k-steps
#pragma unroll
    for (int k = 0; k < KH; ++k) {
      dst[k] = wsrc[((long)ta * KS + ks0 + k) * 64 + lane];
      dst[KH + k] = wsrc[((long)tb * KS + ks0 + k) * 64 + lane];
    }
  };
  bf16x8 wn1[TRIPLE ? KS : 1];                           // TRIPLE: conv1's second block, requested here as well (a block is
                                                         // ~1.3 k cycles of MFMAs: one block ahead is less than an L2 round trip)
  if constexpr (TRIPLE) {
    load_block3(wc1, p.w1f, T3A, T3B, T3C, 0);
    load_block3(wn1, p.w1f, T3A, T3B, T3C, KT);
  } else if constexpr (PAIR) {
    load_block(wc1, p.w1f, wv, wv + BNK_NW, 0);
  } else if (wv < NT_) {
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) wc1[ks] = p.w1f[((long)wv * KS + ks) * 64 + lane];
  }
  BN_STAMP(0);
  // ---- P0: x (gate-shift columns spliced in) -> region A; pads, zero row
  {
    // the folded BatchNorm tables are requested FIRST and stored after the frames: behind the frames' LDS stores (where they
    // used to be read) they were a second, exposed memory round trip per workgroup
    constexpr int NBV = (6 * 16 * ((KS * 32 + 15) / 16) + BNK_THR - 1) / BNK_THR;
    float bv[NBV];
#pragma unroll
    for (int j = 0; j < NBV; ++j) {
      const int i = tid + j * BNK_THR;
      const int v = min(i / CP, 5), c = i - (i / CP) * CP;
      const float* src = v == 0 ? p.s1 : v == 1 ? p.h1 : v == 2 ? p.s2 : v == 3 ? p.h2 : v == 4 ? p.s3 : p.h3;
      bv[j] = src[min(c, C - 1)];
    }
    const int cpr = C >> 3;
    const bf16_t* gg = p.G ? p.G + (long)f0 * hw * p.Fp : nullptr;
    // ---- gate-shift-fuse blend (gsf.hip: gsf_blend_src_kernel's arithmetic, bit for bit): requests first.  Scratch in region B,
    // which nothing else touches before conv1: rows 1 + 12 fs + (0..9) = the spatial sums of frames t-2 .. t+2 (y, then x) of
    // the workgroup's frame fs, row + 10 its fusion weights, row 1 + 12 FPW the conv weights (row 0's head is the k pad's slack)
    constexpr bool blend = BLEND;
    const int gF = p.gs.F, gFh = gF >> 1, gT = p.gs.T_len;
    const int npc8 = p.Fp >> 3;                                  // 16-byte pieces of the slice per pixel
    const int ck0 = blend ? npc8 : 0;                            // the plain pass below skips the slice's pieces: the blend makes them
    const int cprx = cpr - ck0, total = npix * cprx;
    const IDiv dcpr(cprx);
    auto srow = [&](int r) { return reinterpret_cast<float*>(Bt + r * RS); };
    float sv[10], cwv = 0.f;
    u32x4 bxc[3], bxn[3], bxp[3];
    f32x2 bga[3];
    float bgn[3], bgp[3];
    const int sfs = (FPW > 1 && tid >= gF) ? 1 : 0, sc_ = tid - sfs * gF;          // sums: thread = (frame slot, channel)
    const IDiv dgt(blend ? gT : 1);
    if (blend) {
      {
        const int f = f0 + min(sfs, nfr - 1);
        int b, t;
        dgt.divmod(f, b, t);
        const int cc = min(sc_, gF - 1);
#pragma unroll
        for (int row = 0; row < 10; ++row) {
          const int r = row >= 5 ? row - 5 : row;
          const int t2 = min(max(t + r - 2, 0), gT - 1);
          sv[row] = (row >= 5 ? p.gs.xsum : p.gs.ysum)[((long)b * gT + t2) * gF + cc];
        }
        const int ci = min(max(tid - 256, 0), 37);
        cwv = *(ci < 18 ? p.gs.cw1 + ci : ci < 36 ? p.gs.cw2 + (ci - 18) : ci == 36 ? p.gs.cb1 : p.gs.cb2);
      }
      const IDiv dnpc(npc8);
#pragma unroll
      for (int j = 0; j < 3; ++j) {
        const int i = min(tid + j * BNK_THR, npix * npc8 - 1);
        int px, ck;
        dnpc.divmod(i, px, ck);
        const int fs = (FPW > 1 && px >= hw) ? 1 : 0, pix = px - fs * hw;
        const int f = f0 + fs;
        const int t = f - dgt.div(f) * gT;
        const long fn = t < gT - 1 ? f + 1 : f, fpv = t > 0 ? f - 1 : f;
        const bf16_t* xs0 = p.gs.x + (long)pix * p.gs.ldx + ck * 8;
        const long fstride = (long)hw * p.gs.ldx;
        // a piece's neighbour: frame t+1 while its channels are in gate group 0, t-1 in group 1; the one piece that straddles
        // F/2 needs both (the others repeat the first address: no branch around a load, and no second line either)
        const bool lo = ck * 8 < gFh, strad = lo && ck * 8 + 7 >= gFh;
        bxc[j] = *reinterpret_cast<const u32x4*>(xs0 + f * fstride);
        bxn[j] = *reinterpret_cast<const u32x4*>(xs0 + (lo ? fn : fpv) * fstride);
        bxp[j] = *reinterpret_cast<const u32x4*>(xs0 + ((lo && !strad) ? fn : fpv) * fstride);
        bga[j] = *reinterpret_cast<const f32x2*>(p.gs.gate + ((long)f * hw + pix) * 2);
        bgn[j] = p.gs.gate[(fn * hw + pix) * 2];
        bgp[j] = p.gs.gate[(fpv * hw + pix) * 2 + 1];
      }
    }
    constexpr int NLD = 9;                                       // independent 16-byte loads in flight per thread: a workgroup's
    u32x4 v[NLD];                                                // 72 KB (7 x 7 x 368, two frames) in ONE round trip
    auto x_issue = [&](int i0) {
#pragma unroll
      for (int b = 0; b < NLD; ++b) {
        const int i = min(i0 + b * BNK_THR, total - 1);
        int px, ck;
        dcpr.divmod(i, px, ck);
        const int k = (ck + ck0) * 8;
        const bf16_t* src = (gg && k < p.Fp) ? gg + (long)px * p.Fp + k : xg + (long)px * C + k;
        v[b] = *reinterpret_cast<const u32x4*>(src);
      }
    };
    auto x_store = [&](int i0) {
#pragma unroll
      for (int b = 0; b < NLD; ++b) {
        const int i = i0 + b * BNK_THR;
        if (i < total) {
          int px, ck;
          dcpr.divmod(i, px, ck);
          ck += ck0;
          TD_LDS_CHECK(px * RS + ck * 16, 16, FPW * hw * RS);
          *reinterpret_cast<u32x4*>(At + px * RS + ck * 16) = v[b];
        }
      }
    };
    x_issue(tid);
    TD_ISSUE_FENCE();
    // (the blend runs on requests that were made before the frames': it is done while those travel)
    if (blend) {
      // sums (zeros outside the clip) and conv weights -> scratch; fusion weights of the workgroup's frames: the 3x3 conv over
      // the (channel, time) plane of the spatial means + sigmoid; then the blend of this thread's pieces into region A
      if (sfs < nfr && sc_ < gF) {
        const int f = f0 + sfs, t = f - dgt.div(f) * gT;
#pragma unroll
        for (int row = 0; row < 10; ++row) {
          const int t2 = t + (row >= 5 ? row - 5 : row) - 2;
          srow(1 + 12 * sfs + row)[sc_] = (t2 >= 0 && t2 < gT) ? sv[row] : 0.f;
        }
      }
      float* cwl = srow(1 + 12 * FPW);
      if (tid >= 256 && tid < 256 + 38) cwl[tid - 256] = cwv;
      __syncthreads();
      {
        const int fs = (FPW > 1 && tid >= p.Fp) ? 1 : 0, c = tid - fs * p.Fp;
        if (fs < nfr && c < p.Fp) {
          const int t = (f0 + fs) - dgt.div(f0 + fs) * gT;
          const int rb = 1 + 12 * fs;
          const float inv_hw = 1.0f / (float)hw;
          float wgt = 0.f;
          if (c < gF) {
            const int g = c >= gFh;
            const int cl = c - g * gFh;
            const float* cw = cwl + 18 * g;
            float a = cwl[36 + g];
#pragma unroll
            for (int dc = -1; dc <= 1; ++dc) {
              const int c2 = cl + dc;
              if (c2 < 0 || c2 >= gFh) continue;
              const int cc = g * gFh + c2;
#pragma unroll
              for (int dt = -1; dt <= 1; ++dt) {
                const int t2 = t + dt;
                if (t2 < 0 || t2 >= gT) continue;
                const float rm = (srow(rb + 5 + dt + 2)[cc] - srow(rb + dt + 2)[cc]) * inv_hw;
                const int r2 = dt + 2 + (g ? -1 : 1);
                const float ysh = srow(rb + r2)[cc] * inv_hw;
                a = fmaf(cw[(dc + 1) * 3 + (dt + 1)], ysh, a);
                a = fmaf(cw[9 + (dc + 1) * 3 + (dt + 1)], rm, a);
              }
            }
            wgt = sigmoidf_(a);
          }
          TD_LDS_CHECK((Bt - smem) + (rb + 11) * RS + c * 4, 4, (Zr - smem));
          srow(rb + 10)[c] = wgt;
          srow(rb + 11)[c] = 1.0f - wgt;
        }
      }
      __syncthreads();
      const IDiv dnpc(npc8);
#pragma unroll
      for (int j = 0; j < 3; ++j) {
        const int i = tid + j * BNK_THR;
        if (i < npix * npc8) {
          int px, ck;
          dnpc.divmod(i, px, ck);
          const int fs = (FPW > 1 && px >= hw) ? 1 : 0;
          const int t = (f0 + fs) - dgt.div(f0 + fs) * gT;
          const bool has_next = t < gT - 1, has_prev = t > 0;
          // a PAIR of channels per step (F / 2 is even: a pair never straddles the gate groups or the fold's end), packed fp32
          // math -- the blend is bound by instruction issue: ~12 operations per element as scalar code, 3 items per thread
          const float* fwl = srow(1 + 12 * fs + 10) + ck * 8;
          const float* oml = srow(1 + 12 * fs + 11) + ck * 8;
          const f32x4 w0 = *reinterpret_cast<const f32x4*>(fwl), w1 = *reinterpret_cast<const f32x4*>(fwl + 4);
          const f32x4 m0 = *reinterpret_cast<const f32x4*>(oml), m1 = *reinterpret_cast<const f32x4*>(oml + 4);
          const float gn_ = has_next ? bgn[j] : 0.f, gp_ = has_prev ? bgp[j] : 0.f;
          const bool lo_piece = ck * 8 < gFh;
          u32x4 o;
#pragma unroll
          for (int pr = 0; pr < 4; ++pr) {
            const int ci = ck * 8 + 2 * pr;
            const bool g = ci >= gFh;
            const unsigned xcw = bxc[j][pr], nbw = (g && lo_piece) ? bxp[j][pr] : bxn[j][pr];      // g in a "lo" piece: the straddler's upper part
            const f32x2 xv = {__builtin_bit_cast(float, xcw << 16), __builtin_bit_cast(float, xcw & 0xffff0000u)};
            const f32x2 nb = {__builtin_bit_cast(float, nbw << 16), __builtin_bit_cast(float, nbw & 0xffff0000u)};
            const float gate = g ? bga[j][1] : bga[j][0], gsh = g ? gp_ : gn_;
            const f32x2 gate2 = {gate, gate}, gsh2 = {gsh, gsh};
            const f32x2 wv = pr < 2 ? (f32x2){w0[2 * pr], w0[2 * pr + 1]} : (f32x2){w1[2 * pr - 4], w1[2 * pr - 3]};
            const f32x2 om = pr < 2 ? (f32x2){m0[2 * pr], m0[2 * pr + 1]} : (f32x2){m1[2 * pr - 4], m1[2 * pr - 3]};
            const f32x2 r = __builtin_elementwise_fma(-gate2, xv, xv);
            const f32x2 ysh = gsh2 * nb;
            const f32x2 t = r * om;
            const f32x2 ov = __builtin_elementwise_fma(ysh, wv, t);
            const bf16x2 ob = {(bf16_t)ov[0], (bf16_t)ov[1]};
            o[pr] = ci < gF ? __builtin_bit_cast(unsigned, ob) : xcw;
          }
          TD_LDS_CHECK(px * RS + ck * 16, 16, FPW * hw * RS);
          *reinterpret_cast<u32x4*>(At + px * RS + ck * 16) = o;
        }
      }
    }
    x_store(tid);
    for (int i0 = tid + BNK_THR * NLD; i0 < total; i0 += BNK_THR * NLD) {
      x_issue(i0);
      TD_ISSUE_FENCE();
      x_store(i0);
    }
    // the pad bytes behind every row (they meet the zero weights of the k pad, but 0 * stale NaN = NaN), the zero row
    // (region B's pad bytes that the grouped conv reads are written by conv1 itself: the zeros of its overhang tile)
    const int padp = (RS - C * 2) >> 4;
    for (int i = tid; i < 2 * FPW * hw * padp; i += BNK_THR) {
      const int r = i / padp, j = i - r * padp;
      *reinterpret_cast<u32x4*>(smem + r * RS + C * 2 + j * 16) = (u32x4){0u, 0u, 0u, 0u};
    }
    for (int i = tid; i < (RS + 64) >> 4; i += BNK_THR) *reinterpret_cast<u32x4*>(Zr + i * 16) = (u32x4){0u, 0u, 0u, 0u};
    // the k pad of the last loaded row reads on into the next row (KS * 32 > C + pad / 2): region B's head, or the first
    // unused row of region A when the workgroup has fewer frames -- finite before anything has been written there
    if (tid < 4) *reinterpret_cast<u32x4*>(At + npix * RS + tid * 16) = (u32x4){0u, 0u, 0u, 0u};
#pragma unroll
    for (int j = 0; j < NBV; ++j) {
      const int i = tid + j * BNK_THR;
      if (i < 6 * CP) bnv[i] = (i - (i / CP) * CP) < C ? bv[j] : 0.f;      // channels >= C: exact zeros out of every epilogue
    }
  }
  __syncthreads();
  BN_STAMP(1);

  // this lane's pixel row in each pixel tile (rows beyond npix clamp to row 0 and are never stored)
  // ... and where it stores, as an offset from a region's base: its own row, or the trash row (no branch around a store:
  // the tiles of a wave then schedule as one block)
  int prow[NPTM], srowA[NPTM];
#pragma unroll
  for (int pt = 0; pt < NPTM; ++pt) {
    const int px = pt * 16 + pl;
    prow[pt] = (px < npix ? px : 0) * RS;
    srowA[pt] = px < npix ? px * RS : (int)(Tr - At);
  }
  const int dAB = (int)(Bt - At);
  // one 16-channel tile of a 1x1 conv: acc[pt] = sum_ks W[ks] . act[pixel tile pt][ks].  The pixel tiles go in two halves:
  // the LDS reads of one half are issued before the MFMAs of the other (sched_barrier pins that order), so a wave's reads
  // travel under its own MFMAs instead of in front of them
  constexpr int HA = (NPTM + 1) / 2;
  auto contract = [&](const bf16x8 (&wc)[KS], const unsigned char* act, f32x4 (&acc)[NPTM]) {
#pragma unroll
    for (int pt = 0; pt < NPTM; ++pt) acc[pt] = (f32x4){0.f, 0.f, 0.f, 0.f};
    bf16x8 a[NPTM];
    const int kq = 16 * q;
#pragma unroll
    for (int pt = 0; pt < HA; ++pt) a[pt] = *reinterpret_cast<const bf16x8*>(act + prow[pt] + kq);
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      const int kb = 64 * ks + kq;
#pragma unroll
      for (int pt = HA; pt < NPTM; ++pt) a[pt] = *reinterpret_cast<const bf16x8*>(act + prow[pt] + kb);
#pragma unroll
      for (int pt = 0; pt < HA; ++pt) acc[pt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wc[ks], a[pt], acc[pt], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      if (ks + 1 < KS) {
#pragma unroll
        for (int pt = 0; pt < HA; ++pt) a[pt] = *reinterpret_cast<const bf16x8*>(act + prow[pt] + kb + 64);
      }
#pragma unroll
      for (int pt = HA; pt < NPTM; ++pt) acc[pt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wc[ks], a[pt], acc[pt], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
  };

  // half a contraction of TWO channel tiles: k-steps ks0 .. ks0 + KH of both, one activation fragment read per pair of MFMAs
  auto contract_half = [&](const bf16x8 (&wb)[KS], int ks0, const unsigned char* act, f32x4 (&acc0)[NPTM], f32x4 (&acc1)[NPTM]) {
    bf16x8 a[NPTM];
    const int kq = 16 * q;
#pragma unroll
    for (int pt = 0; pt < HA; ++pt) a[pt] = *reinterpret_cast<const bf16x8*>(act + prow[pt] + 64 * ks0 + kq);
#pragma unroll
    for (int k = 0; k < KH; ++k) {
      const int kb = 64 * (ks0 + k) + kq;
#pragma unroll
      for (int pt = HA; pt < NPTM; ++pt) a[pt] = *reinterpret_cast<const bf16x8*>(act + prow[pt] + kb);
#pragma unroll
      for (int pt = 0; pt < HA; ++pt) {
        acc0[pt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wb[k], a[pt], acc0[pt], 0, 0, 0);
        acc1[pt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wb[KH + k], a[pt], acc1[pt], 0, 0, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
      if (k + 1 < KH) {
#pragma unroll
        for (int pt = 0; pt < HA; ++pt) a[pt] = *reinterpret_cast<const bf16x8*>(act + prow[pt] + kb + 64);
      }
#pragma unroll
      for (int pt = HA; pt < NPTM; ++pt) {
        acc0[pt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wb[k], a[pt], acc0[pt], 0, 0, 0);
        acc1[pt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wb[KH + k], a[pt], acc1[pt], 0, 0, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  };
  // a third of a contraction of THREE channel tiles: k-steps ks0 .. ks0 + KT, one activation fragment read per three MFMAs
  auto contract_third = [&](const bf16x8 (&wb)[KS], int ks0, const unsigned char* act, f32x4 (&acc0)[NPTM], f32x4 (&acc1)[NPTM],
                            f32x4 (&acc2)[NPTM]) {
    bf16x8 a[NPTM];
    const int kq = 16 * q;
#pragma unroll
    for (int pt = 0; pt < HA; ++pt) a[pt] = *reinterpret_cast<const bf16x8*>(act + prow[pt] + 64 * ks0 + kq);
#pragma unroll
    for (int k = 0; k < KT; ++k) {
      const int kb = 64 * (ks0 + k) + kq;
#pragma unroll
      for (int pt = HA; pt < NPTM; ++pt) a[pt] = *reinterpret_cast<const bf16x8*>(act + prow[pt] + kb);
#pragma unroll
      for (int pt = 0; pt < HA; ++pt) {
        acc0[pt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wb[k], a[pt], acc0[pt], 0, 0, 0);
        acc1[pt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wb[KT + k], a[pt], acc1[pt], 0, 0, 0);
        acc2[pt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wb[2 * KT + k], a[pt], acc2[pt], 0, 0, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
      if (k + 1 < KT) {
#pragma unroll
        for (int pt = 0; pt < HA; ++pt) a[pt] = *reinterpret_cast<const bf16x8*>(act + prow[pt] + kb + 64);
      }
#pragma unroll
      for (int pt = HA; pt < NPTM; ++pt) {
        acc0[pt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wb[k], a[pt], acc0[pt], 0, 0, 0);
        acc1[pt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wb[KT + k], a[pt], acc1[pt], 0, 0, 0);
        acc2[pt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wb[2 * KT + k], a[pt], acc2[pt], 0, 0, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  };
  // epilogue of conv1 for one channel tile: BN + ReLU -> y1 rows in region B (channels >= C get exact zeros: row pad)
  auto epi1 = [&](int T, const f32x4 (&acc)[NPTM]) {
    const int ch0 = T * 16 + 4 * q;
    const f32x4 sc = *reinterpret_cast<const f32x4*>(bnv + ch0), sh = *reinterpret_cast<const f32x4*>(bnv + CP + ch0);
#pragma unroll
    for (int pt = 0; pt < NPTM; ++pt) {
      bf16x4 o;
#pragma unroll
      for (int r = 0; r < 4; ++r) o[r] = (bf16_t)fmaxf(acc[pt][r] * sc[r] + sh[r], 0.f);
      unsigned char* dst = At + srowA[pt];
      TD_LDS_CHECK(((srowA[pt] < dAB ? dst + dAB : dst) + ch0 * 2) - smem, 8, (Tr + RS) - smem);
      *reinterpret_cast<bf16x4*>((srowA[pt] < dAB ? dst + dAB : dst) + ch0 * 2) = o;
    }
  };

  // ---- P1: conv1: y1 = relu(bn(W1 x'))  (wave = channel tiles wv, wv + 8, ...)
  if constexpr (TRIPLE) {
    bf16x8 (&wn)[KS] = wn1;
    f32x4 acc0[NPTM], acc1[NPTM], acc2[NPTM];
#pragma unroll
    for (int pt = 0; pt < NPTM; ++pt) acc0[pt] = acc1[pt] = acc2[pt] = (f32x4){0.f, 0.f, 0.f, 0.f};
    contract_third(wc1, 0, At, acc0, acc1, acc2);
    load_block3(wc1, p.w1f, T3A, T3B, T3C, 2 * KT);          // the last block travels under the second
    contract_third(wn, KT, At, acc0, acc1, acc2);
    if (wv < NT) {                                           // conv2's first unit: into the registers the second block left
#pragma unroll
      for (int ks = 0; ks < 5; ++ks) wf2[ks] = p.w2f[((long)wv * 5 + ks) * 64 + lane];
    }
    contract_third(wc1, 2 * KT, At, acc0, acc1, acc2);
    epi1(T3A, acc0);
    epi1(T3B, acc1);
    if (t3c_ok) epi1(wv + 2 * BNK_NW, acc2);
  } else if constexpr (PAIR) {
    bf16x8 wn[KS];
    if (wv < NT) {
#pragma unroll
      for (int ks = 0; ks < 5; ++ks) wf2[ks] = p.w2f[((long)wv * 5 + ks) * 64 + lane];       // conv2's first unit
    }
    const int TA = wv, TB = wv + BNK_NW, TC = wv + 2 * BNK_NW;
    load_block(wn, p.w1f, TA, TB, KH);                       // second half-K block of the pair
    f32x4 acc0[NPTM], acc1[NPTM];
#pragma unroll
    for (int pt = 0; pt < NPTM; ++pt) acc0[pt] = acc1[pt] = (f32x4){0.f, 0.f, 0.f, 0.f};
    contract_half(wc1, 0, At, acc0, acc1);
    if (TC < NT) {                                           // the third tile's weights travel under the second half
      // (made unconditional with a clamped tile the kernel spills 158 VGPRs: measured 114 us; the branch stays)
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) wc1[ks] = p.w1f[((long)TC * KS + ks) * 64 + lane];
    }
    contract_half(wn, KH, At, acc0, acc1);
    epi1(TA, acc0);
    epi1(TB, acc1);
    if (TC < NT) {
      contract(wc1, At, acc0);
      epi1(TC, acc0);
    }
  } else {
    bf16x8 (&wc)[KS] = wc1;
    bf16x8 wn[KS];
    if (wv < NT) {
#pragma unroll
      for (int ks = 0; ks < 5; ++ks) wf2[ks] = p.w2f[((long)wv * 5 + ks) * 64 + lane];       // conv2's first unit
    }
    for (int T = wv; T < NT; T += BNK_NW) {
      if (T + BNK_NW < NT) {
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) wn[ks] = p.w1f[((long)(T + BNK_NW) * KS + ks) * 64 + lane];
      }
      const int ch0 = T * 16 + 4 * q;
      const f32x4 sc = *reinterpret_cast<const f32x4*>(bnv + ch0), sh = *reinterpret_cast<const f32x4*>(bnv + CP + ch0);
      f32x4 acc[NPTM];
      contract(wc, At, acc);
      // channels >= C of the last tile get exact zeros (scale = shift = 0) and land in the row pad
#pragma unroll
      for (int pt = 0; pt < NPTM; ++pt) {
        bf16x4 o;
#pragma unroll
        for (int r = 0; r < 4; ++r) o[r] = (bf16_t)fmaxf(acc[pt][r] * sc[r] + sh[r], 0.f);
        unsigned char* dst = At + srowA[pt];
        TD_LDS_CHECK(((srowA[pt] < dAB ? dst + dAB : dst) + ch0 * 2) - smem, 8, (Tr + RS) - smem);
        *reinterpret_cast<bf16x4*>((srowA[pt] < dAB ? dst + dAB : dst) + ch0 * 2) = o;
      }
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) wc[ks] = wn[ks];
    }
  }
  __syncthreads();
  BN_STAMP(2);

  bf16x8 sew1[KS];                                       // SE fc1 fragments of this wave's first hidden tile (requested inside P2)
  // ---- P2: conv2 grouped 3x3 from y1 (taps outside the map read the zero row) -> y2 in region A; squeeze sums per frame
  {
    const int NU = NT;
    // per (pixel tile, k-step): LDS offset of the tap pixel's 8 input channels in region B (k-slot s = 4 ks + q = half * 9 +
    // tap), or of the zero row -- the same for every 16-channel unit up to the unit's own 32-byte column offset
    int toff[NPTM][5];
    {
      const int zoff = (int)(Zr - Bt);
      int dyv[5], dxv[5], doff[5];
      bool sok[5];
#pragma unroll
      for (int ks = 0; ks < 5; ++ks) {                   // k-slot -> (tap, half): tap-major fragments (engine.pack_gconv_frags(tap_major))
        // the two k-slots of a ds_read_b128 lane group (q = 0 / 1: slots 4 ks, 4 ks + 1) are the two 16-byte halves of ONE tap
        // pixel: at this kernel's even row stride (14 or 6 slots mod 16) the group's 16 lanes fall on 16 distinct bank slots;
        // the half-major order of gconv3x3_mfma_kernel (slots of adjacent taps, same half) gave two-way conflicts on every read
        // (for group width 8 an output's sum is bit-identical under either order in every test; for 16-wide groups -- both
        //  halves of a k-slot pair carry weights of the same output -- it is not, and the caller keeps the half-major order)
        const int sidx = 4 * ks + q;
        const int half = TAPM ? (sidx & 1) : (sidx >= 9 ? 1 : 0);
        const int tap = TAPM ? (sidx >> 1) : sidx - 9 * half;
        const int ty = (tap >= 3 ? 1 : 0) + (tap >= 6 ? 1 : 0);
        dyv[ks] = ty - 1;
        dxv[ks] = tap - 3 * ty - 1;
        sok[ks] = sidx < 18;
        doff[ks] = (dyv[ks] * p.w + dxv[ks]) * RS + half * 16;
      }
      const IDiv dhw(hw), dw(p.w);
#pragma unroll
      for (int pt = 0; pt < NPTM; ++pt) {
        const int px = min(pt * 16 + pl, npix - 1);
        int f, r, oy, ox;
        dhw.divmod(px, f, r);
        dw.divmod(r, oy, ox);
#pragma unroll
        for (int ks = 0; ks < 5; ++ks) {
          const bool ok = sok[ks] && (unsigned)(oy + dyv[ks]) < (unsigned)p.h && (unsigned)(ox + dxv[ks]) < (unsigned)p.w;
          toff[pt][ks] = ok ? px * RS + doff[ks] : zoff;
          TD_LDS_CHECK((Bt - smem) + toff[pt][ks] + (NT - 1) * 32, 16, lds_bytes);
        }
      }
    }
    // the weights of ALL of this wave's units are requested up front (the first came with conv1): a unit is ~1.3 k cycles of
    // work, far less than an L2 round trip, so a one-ahead prefetch would expose one round trip per unit
    constexpr int MAXU = KS <= 8 ? 2 : 3;          // units per wave: C <= 256 has at most 16 units over the 8 waves
    bf16x8 wfx[MAXU - 1][5];
#pragma unroll
    for (int j = 1; j < MAXU; ++j) {
      const int U = min(wv + j * BNK_NW, NU - 1);
#pragma unroll
      for (int ks = 0; ks < 5; ++ks) wfx[j - 1][ks] = p.w2f[((long)U * 5 + ks) * 64 + lane];
    }
    BN_STAMP(8);
    // one unit: this wave's ju-th 16-channel group with the weights wfu (all requested above / with conv1)
    auto unit = [&](int ju, const bf16x8 (&wfu)[5]) {
      const int U = wv + ju * BNK_NW;
      if (U >= NU) return;
      const int ch0 = U * 16 + 4 * q;
      const f32x4 sc = *reinterpret_cast<const f32x4*>(bnv + 2 * CP + ch0), sh = *reinterpret_cast<const f32x4*>(bnv + 3 * CP + ch0);
      float psum[FPW][4];
#pragma unroll
      for (int f = 0; f < FPW; ++f)
#pragma unroll
        for (int r = 0; r < 4; ++r) psum[f][r] = 0.f;
      // k-step outer, pixel tile inner: NPTM independent accumulators (a tile-at-a-time loop is five dependent MFMAs)
      f32x4 acc[NPTM];
#pragma unroll
      for (int pt = 0; pt < NPTM; ++pt) acc[pt] = (f32x4){0.f, 0.f, 0.f, 0.f};
      {
        bf16x8 yf[NPTM];
        const unsigned char* bu = Bt + U * 32;
#pragma unroll
        for (int pt = 0; pt < HA; ++pt) yf[pt] = *reinterpret_cast<const bf16x8*>(bu + toff[pt][0]);
#pragma unroll
        for (int ks = 0; ks < 5; ++ks) {
#pragma unroll
          for (int pt = HA; pt < NPTM; ++pt) yf[pt] = *reinterpret_cast<const bf16x8*>(bu + toff[pt][ks]);
#pragma unroll
          for (int pt = 0; pt < HA; ++pt) acc[pt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wfu[ks], yf[pt], acc[pt], 0, 0, 0);
          __builtin_amdgcn_sched_barrier(0);
          if (ks + 1 < 5) {
#pragma unroll
            for (int pt = 0; pt < HA; ++pt) yf[pt] = *reinterpret_cast<const bf16x8*>(bu + toff[pt][ks + 1]);
          }
#pragma unroll
          for (int pt = HA; pt < NPTM; ++pt) acc[pt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wfu[ks], yf[pt], acc[pt], 0, 0, 0);
          __builtin_amdgcn_sched_barrier(0);
        }
      }
#pragma unroll
      for (int pt = 0; pt < NPTM; ++pt) {
        const int px = pt * 16 + pl;
        const bool pok = px < npix;
        bf16x4 o;
#pragma unroll
        for (int r = 0; r < 4; ++r) o[r] = (bf16_t)fmaxf(acc[pt][r] * sc[r] + sh[r], 0.f);
        *reinterpret_cast<bf16x4*>(At + srowA[pt] + ch0 * 2) = o;
        const bool second = FPW > 1 && px >= hw;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float v = pok ? (float)o[r] : 0.f;
          if (FPW > 1) psum[FPW - 1][r] += second ? v : 0.f;
          psum[0][r] += second ? 0.f : v;
        }
      }
#pragma unroll
      for (int f = 0; f < FPW; ++f)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float v = bnk_row16_sum(psum[f][r]);
          if (pl == 0 && ch0 + r < C) pooled[f * C + ch0 + r] = v;
        }
      BN_STAMP(9 + ju);
    };
    // The SE excitation's first fc1 tile (KS fragments) is requested in front of the wave's LAST unit slot: by then the other
    // units' weights are dead registers, and the L2 round trip that used to open the SE phase (7.4 k of its 18.7 k cycles at
    // 7 x 7 x 368, tools/bench_bneck.py) travels under a unit's MFMAs instead.
    auto se_prefetch = [&]() {
      const int KS1 = (C + 31) >> 5, RT = (p.se.R + 15) >> 4;
      const int tc = min(wv, RT - 1);
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) sew1[ks] = p.se.w1f[((long)tc * KS1 + min(ks, KS1 - 1)) * 64 + lane];
    };
    unit(0, wf2);
    if constexpr (MAXU == 2) {
      se_prefetch();
      unit(1, wfx[0]);
    } else {
      unit(1, wfx[0]);
      se_prefetch();
      unit(2, wfx[MAXU - 2]);
    }
  }
  __syncthreads();
  BN_STAMP(3);

  bf16x8 wc3[KS];
  if constexpr (TRIPLE) {
    load_block3(wc3, p.w3f, T3A, T3B, T3C, 0);
  } else if constexpr (PAIR) {
    load_block(wc3, p.w3f, wv, wv + BNK_NW, 0);
  } else if (wv < NT) {
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) wc3[ks] = p.w3f[((long)wv * KS + ks) * 64 + lane];
  }
  // ---- P3: SE excitation of the workgroup's frames on the MFMA pipe (scratch: the dead y1 region), y2 *= gate in place
  {
    SeP se = p.se;
    se.pooled = pooled; se.n_parts = 1; se.inv_cnt = 1.0f / (float)hw; se.C = C; se.gate_out = nullptr;
    se_excite_lds<KS, 3, true, BNK_NW, true>(se, 0, nfr, nfr - 1, gtab, C, Bt, p.dbg ? p.dbg + (long)blockIdx.x * 16 + 12 : nullptr,
                                             &sew1);
    const int cpr = C >> 3;
    const IDiv dcpr(cpr);
    for (int i = tid; i < npix * cpr; i += BNK_THR) {
      int px, ck;
      dcpr.divmod(i, px, ck);
      const float* g = gtab + ((FPW > 1 && px >= hw) ? C : 0) + ck * 8;
      bf16x8* ptr8 = reinterpret_cast<bf16x8*>(At + px * RS + ck * 16);
      bf16x8 v = *ptr8;
      const f32x4 g0 = *reinterpret_cast<const f32x4*>(g), g1 = *reinterpret_cast<const f32x4*>(g + 4);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        v[e] = (bf16_t)((float)v[e] * g0[e]);
        v[4 + e] = (bf16_t)((float)v[4 + e] * g1[e]);
      }
      *ptr8 = v;
    }
  }
  __syncthreads();
  BN_STAMP(4);

  // the optional tail P6 (tap maps of the next gate-shift site): its requests -- weight fragments, BatchNorm entries -- go out
  // behind conv3's last contraction block / behind the epilogues, into the registers those leave (behind the output stores of P5
  // they would wait for the stores' acknowledgements: the tail runs in front of P5)
  const bool qtail = p.qt.Q != nullptr;
  const int qF = p.qt.F, nch = (qF + 7) >> 3, KSc = (nch + 3) >> 2, PSQ = (nch | 1) * 16;
  constexpr int QRSP = 224;                                    // bytes per pixel of the per-tap sums: 54 floats + 2
  unsigned char* qa = At;                                      // [npix][PSQ]  relu(bn(out[:, :8 nch])) in bf16
  unsigned char* wl = qa + FPW * hw * PSQ;                     // [4][KSc][64] 16-byte fragments
  unsigned char* Pl = wl + 4 * KSc * 1024;                     // [npix][QRSP] per-tap sums
  u32x4 wr[2];
  f32x4 bs[3][2], bh[3][2];
  int ipx[3], ick[3];
  const IDiv dnch(qtail ? nch : 1), dw_(p.w), dhw_(hw);
  auto q_request = [&]() {                                     // the fragments
    if (!qtail) return;
#pragma unroll
    for (int b = 0; b < 2; ++b) wr[b] = reinterpret_cast<const u32x4*>(p.qt.wqf)[min(tid + b * BNK_THR, 4 * KSc * 64 - 1)];
  };
  auto q_request_bn = [&]() {                                  // behind the epilogues (48 registers)
    if (!qtail) return;
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      const int i = min(tid + j * BNK_THR, npix * nch - 1);
      dnch.divmod(i, ipx[j], ick[j]);
      const float* bq = p.qt.bn + ick[j] * 8;
      bs[j][0] = *reinterpret_cast<const f32x4*>(bq);
      bs[j][1] = *reinterpret_cast<const f32x4*>(bq + 4);
      bh[j][0] = *reinterpret_cast<const f32x4*>(bq + nch * 8);
      bh[j][1] = *reinterpret_cast<const f32x4*>(bq + nch * 8 + 4);
    }
  };
  // ---- P4: conv3 on the gated y2, + residual x, ReLU -> output rows in region B
  // residual of one channel tile: 4 channels of this lane's pixel per pixel tile, L2-hot
  auto load_res = [&](int T, bf16x4 (&rres)[NPTM]) {
    const int ch0 = T * 16 + 4 * q;
    const bool cok = ch0 < C;
#pragma unroll
    for (int pt = 0; pt < NPTM; ++pt) {
      const int px = pt * 16 + pl;
      rres[pt] = *reinterpret_cast<const bf16x4*>(xg + (long)((px < npix && cok) ? px : 0) * C + (cok ? ch0 : 0));
    }
  };
  auto epi3 = [&](int T, const f32x4 (&acc)[NPTM], const bf16x4 (&rres)[NPTM]) {
    const int ch0 = T * 16 + 4 * q;
    const f32x4 sc = *reinterpret_cast<const f32x4*>(bnv + 4 * CP + ch0), sh = *reinterpret_cast<const f32x4*>(bnv + 5 * CP + ch0);
#pragma unroll
    for (int pt = 0; pt < NPTM; ++pt) {
      bf16x4 o;
#pragma unroll
      for (int r = 0; r < 4; ++r) o[r] = (bf16_t)fmaxf(acc[pt][r] * sc[r] + sh[r] + (float)rres[pt][r], 0.f);
      unsigned char* dst = At + srowA[pt];
      *reinterpret_cast<bf16x4*>((srowA[pt] < dAB ? dst + dAB : dst) + ch0 * 2) = o;
    }
  };
  if constexpr (TRIPLE) {
    bf16x8 wn[KS];
    load_block3(wn, p.w3f, T3A, T3B, T3C, KT);
    f32x4 acc0[NPTM], acc1[NPTM], acc2[NPTM];
#pragma unroll
    for (int pt = 0; pt < NPTM; ++pt) acc0[pt] = acc1[pt] = acc2[pt] = (f32x4){0.f, 0.f, 0.f, 0.f};
    contract_third(wc3, 0, At, acc0, acc1, acc2);
    load_block3(wc3, p.w3f, T3A, T3B, T3C, 2 * KT);
    contract_third(wn, KT, At, acc0, acc1, acc2);
    // the residual rows of the three tiles: requested into the registers the second block left, they travel under the third
    bf16x4 rA[NPTM], rB[NPTM], rC[NPTM];
    load_res(T3A, rA);
    load_res(T3B, rB);
    load_res(T3C, rC);
    contract_third(wc3, 2 * KT, At, acc0, acc1, acc2);
    q_request();
    epi3(T3A, acc0, rA);
    epi3(T3B, acc1, rB);
    if (t3c_ok) epi3(wv + 2 * BNK_NW, acc2, rC);
  } else if constexpr (PAIR) {
    bf16x8 wn[KS];
    const int TA = wv, TB = wv + BNK_NW, TC = wv + 2 * BNK_NW;
    load_block(wn, p.w3f, TA, TB, KH);
    bf16x4 rA[NPTM], rB[NPTM];
    load_res(TA, rA);
    f32x4 acc0[NPTM], acc1[NPTM];
#pragma unroll
    for (int pt = 0; pt < NPTM; ++pt) acc0[pt] = acc1[pt] = (f32x4){0.f, 0.f, 0.f, 0.f};
    contract_half(wc3, 0, At, acc0, acc1);
    load_res(TB, rB);
    if (TC < NT) {
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) wc3[ks] = p.w3f[((long)TC * KS + ks) * 64 + lane];
    }
    contract_half(wn, KH, At, acc0, acc1);
    if (!(TC < NT)) q_request();
    epi3(TA, acc0, rA);
    if (TC < NT) load_res(TC, rA);
    epi3(TB, acc1, rB);
    if (TC < NT) {
      contract(wc3, At, acc0);
      q_request();
      epi3(TC, acc0, rA);
    }
  } else {
    bf16x8 (&wc)[KS] = wc3;
    bf16x8 wn[KS];
    for (int T = wv; T < NT; T += BNK_NW) {
      if (T + BNK_NW < NT) {
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) wn[ks] = p.w3f[((long)(T + BNK_NW) * KS + ks) * 64 + lane];
      }
      const int ch0 = T * 16 + 4 * q;
      const bool cok = ch0 < C;
      const f32x4 sc = *reinterpret_cast<const f32x4*>(bnv + 4 * CP + ch0), sh = *reinterpret_cast<const f32x4*>(bnv + 5 * CP + ch0);
      bf16x4 rres[NPTM];                                 // residual: 4 channels of this lane's pixel, L2-hot
#pragma unroll
      for (int pt = 0; pt < NPTM; ++pt) {
        const int px = pt * 16 + pl;
        rres[pt] = *reinterpret_cast<const bf16x4*>(xg + (long)((px < npix && cok) ? px : 0) * C + (cok ? ch0 : 0));
      }
      f32x4 acc[NPTM];
      contract(wc, At, acc);
#pragma unroll
      for (int pt = 0; pt < NPTM; ++pt) {
        bf16x4 o;
#pragma unroll
        for (int r = 0; r < 4; ++r) o[r] = (bf16_t)fmaxf(acc[pt][r] * sc[r] + sh[r] + (float)rres[pt][r], 0.f);
        unsigned char* dst = At + srowA[pt];
        *reinterpret_cast<bf16x4*>((srowA[pt] < dAB ? dst + dAB : dst) + ch0 * 2) = o;
      }
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) wc[ks] = wn[ks];
    }
    q_request();
  }
  q_request_bn();
  __syncthreads();
  BN_STAMP(5);
  // ---- P6 (optional): Q[f][pixel][jg] = conv2d_3x3(relu(bn(out[f][:, :F])), w3d[g][:, j]) for the three temporal taps j and both
  // gate groups g of the NEXT site (impl/gsf.py:49-52 with the conv3d as three 2-D convs per frame; gsf.hip launch 1a), as
  //   P[pixel][tap][jg] = sum_c W[jg][tap][c] * a[pixel][c]      one 1x1 contraction on the MFMA pipe, 54 of 64 output rows, then
  //   Q[pixel][jg]      = sum_tap P[pixel + tap][tap][jg]        nine fp32 adds in tap order, taps outside the frame skipped
  // -- a third of the implicit GEMM's MFMAs and none of its nine-fold activation reads (that form, measured here first, was bound
  // by the LDS pipe: 10.7 k cycles per workgroup against the launch it replaced).  Region A is dead and takes all three arrays.
  if (qtail) {
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      if (tid + j * BNK_THR < npix * nch) {
        const bf16x8 v = *reinterpret_cast<const bf16x8*>(Bt + ipx[j] * RS + ick[j] * 16);
        bf16x8 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          o[e] = (bf16_t)fmaxf(fmaf((float)v[e], bs[j][0][e], bh[j][0][e]), 0.f);
          o[4 + e] = (bf16_t)fmaxf(fmaf((float)v[4 + e], bs[j][1][e], bh[j][1][e]), 0.f);
        }
        TD_LDS_CHECK(ipx[j] * PSQ + ick[j] * 16, 16, wl - smem);
        *reinterpret_cast<bf16x8*>(qa + ipx[j] * PSQ + ick[j] * 16) = o;
      }
    }
#pragma unroll
    for (int b = 0; b < 2; ++b)
      if (tid + b * BNK_THR < 4 * KSc * 64) {
        TD_LDS_CHECK((wl - smem) + (tid + b * BNK_THR) * 16, 16, Pl - smem);
        *reinterpret_cast<u32x4*>(wl + (tid + b * BNK_THR) * 16) = wr[b];
      }
    __syncthreads();
    BN_STAMP(8);
    constexpr int KSCM = (KS + 3) / 4;                           // nch <= KS
    for (int pt = wv; pt * 16 < npix; pt += BNK_NW) {
      const int row = min(pt * 16 + pl, npix - 1);
      f32x4 acc[4];
#pragma unroll
      for (int rt = 0; rt < 4; ++rt) acc[rt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < KSCM; ++ks) {
        if (ks < KSc) {                                          // (uniform)
          const int chunk = 4 * ks + q;                          // chunks behind the slice meet zero weights: any finite operand
          const bf16x8 av = *reinterpret_cast<const bf16x8*>(qa + row * PSQ + (chunk < nch ? chunk : 0) * 16);
#pragma unroll
          for (int rt = 0; rt < 4; ++rt)
            acc[rt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(*reinterpret_cast<const bf16x8*>(wl + ((rt * KSc + ks) * 64 + lane) * 16),
                                                              av, acc[rt], 0, 0, 0);
        }
      }
      if (pt * 16 + pl < npix) {
#pragma unroll
        for (int rt = 0; rt < 4; ++rt)
          if (rt * 16 + 4 * q < 56) {
            TD_LDS_CHECK((Pl - smem) + row * QRSP + (rt * 16 + 4 * q) * 4, 16, Bt - smem);
            *reinterpret_cast<f32x4*>(Pl + row * QRSP + (rt * 16 + 4 * q) * 4) = acc[rt];
          }
      }
    }
    __syncthreads();
    BN_STAMP(9);
    {
      const IDiv d6(6);
      for (int i = tid; i < npix * 6; i += BNK_THR) {
        int px, jg, fr, pix, py, pxx;
        d6.divmod(i, px, jg);
        dhw_.divmod(px, fr, pix);
        dw_.divmod(pix, py, pxx);
        float sum = 0.f;
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
          const int dy = tap / 3 - 1, dx = tap % 3 - 1;
          const bool ok = (unsigned)(py + dy) < (unsigned)p.h && (unsigned)(pxx + dx) < (unsigned)p.w;
          const float v = *reinterpret_cast<const float*>(Pl + (ok ? px + dy * p.w + dx : px) * QRSP + (tap * 6 + jg) * 4);
          sum += ok ? v : 0.f;
        }
        p.qt.Q[((long)f0 * hw + px) * 6 + jg] = sum;
      }
    }
    BN_STAMP(7);
  }
  // ---- P5: the output rows leave as whole 16-byte pieces (the frames of a workgroup are contiguous in memory)
  {
    const int cpr = C >> 3;
    const IDiv dcpr(cpr);
    bf16_t* og = p.out + (long)f0 * hw * C;
    for (int i = tid; i < npix * cpr; i += BNK_THR) {
      int px, ck;
      dcpr.divmod(i, px, ck);
      const u32x4 v = *reinterpret_cast<const u32x4*>(Bt + px * RS + ck * 16);
      *reinterpret_cast<u32x4*>(og + (long)px * C + ck * 8) = v;
      if (p.out2 && ck * 8 < p.n2) *reinterpret_cast<u32x4*>(p.out2 + ((long)f0 * hw + px) * p.n2 + ck * 8) = v;
    }
  }
  BN_STAMP(6);
}

static int bneck_fpw(int hw) { return hw <= 64 ? 2 : 1; }

static long long* g_bneck_dbg = nullptr;
extern "C" int tdeed_bneck_set_debug(void* buf) { g_bneck_dbg = (long long*)buf; return TDEED_OK; }

// 1 when (h, w, C, R) is served: 7 x 7 x 368 (two frames per workgroup) and 14 x 14 x 152 (one) are what it was built for
extern "C" int tdeed_bneck_fits(int h, int w, int C, int R) {
  const int hw = h * w, KS = (C + 31) / 32, fpw = bneck_fpw(hw);
  if (h < 3 || w < 3 || C % 8 != 0 || R < 1 || R > 96 || C > 384) return 0;
  if (!(KS == 5 || KS == 12)) return 0;
  if ((C + 15) / 16 > 3 * BNK_NW) return 0;              // conv2 keeps the weights of at most 3 units per wave
  if ((fpw * hw + 15) / 16 > (KS == 12 ? 7 : (fpw == 2 ? 8 : 13))) return 0;
  if ((size_t)se_excite_scratch_bytes(C, R) > (size_t)fpw * hw * bneck_rs(C)) return 0;       // SE scratch lives in the y1 region
  return bneck_smem(fpw, hw, C) <= 160 * 1024 ? 1 : 0;
}

static int bneck_launch(BneckP& p, hipStream_t st) {
  const int h = p.h, w = p.w, C = p.C, N = p.N;
  p.dbg = g_bneck_dbg;
  const int hw = h * w, fpw = bneck_fpw(hw), KS = (C + 31) / 32;
  const size_t smem = bneck_smem(fpw, hw, C);
  static TdDevOnce attr_set;
  if (!attr_set.get()) {
    hipError_t e = hipSuccess;
#define BNK_ATTR(...) if (e == hipSuccess) e = hipFuncSetAttribute((const void*)bneck_kernel<__VA_ARGS__>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)
#define BNK_ATTR4(...) BNK_ATTR(__VA_ARGS__, true, true); BNK_ATTR(__VA_ARGS__, true, false); BNK_ATTR(__VA_ARGS__, false, true); \
                       BNK_ATTR(__VA_ARGS__, false, false)
    BNK_ATTR4(12, 2, 7); BNK_ATTR4(5, 1, 13); BNK_ATTR4(5, 2, 8); BNK_ATTR4(12, 1, 7);
#undef BNK_ATTR4
#undef BNK_ATTR
    if (e != hipSuccess) { tdeed_set_error("bneck: hipFuncSetAttribute: %s", hipGetErrorString(e)); return TDEED_ERR_RUNTIME; }
    attr_set.set();
  }
  const int grid = (N + fpw - 1) / fpw;
  const bool tm = p.w2_tap_major != 0;
  const bool bl = p.gs.gate != nullptr;
#define BNK_GO1(...) hipLaunchKernelGGL((bneck_kernel<__VA_ARGS__>), dim3(grid), dim3(BNK_THR), smem, st, p)
#define BNK_GO(...) do { if (tm && bl) BNK_GO1(__VA_ARGS__, true, true); else if (tm) BNK_GO1(__VA_ARGS__, true, false); \
                         else if (bl) BNK_GO1(__VA_ARGS__, false, true); else BNK_GO1(__VA_ARGS__, false, false); } while (0)
  if (KS == 12 && fpw == 2) BNK_GO(12, 2, 7);
  else if (KS == 5 && fpw == 1) BNK_GO(5, 1, 13);
  else if (KS == 5 && fpw == 2) BNK_GO(5, 2, 8);
  else if (KS == 12 && fpw == 1) BNK_GO(12, 1, 7);   // e.g. 13 x 7 maps
  else { tdeed_set_error("bneck: KS=%d with %d frames per workgroup", KS, fpw); return TDEED_ERR_ARG; }
#undef BNK_GO
#undef BNK_GO1
  TD_LAUNCH_CHECK("bneck");
  return TDEED_OK;
}

static int bneck_fill(BneckP& p, const void* x, int N, int h, int w, int C, const void* w1f, const float* s1, const float* h1,
                      const void* w2f, const float* s2, const float* h2, const void* se_w1f, const float* se_b1,
                      const void* se_w2f, const float* se_b2, int R, const void* w3f, const float* s3, const float* h3, void* out,
                      void* out2, int n2, int w2_tap_major) {
  TD_CHECK(x && w1f && s1 && h1 && w2f && s2 && h2 && se_w1f && se_b1 && se_w2f && se_b2 && w3f && s3 && h3 && out,
           "bneck: null pointer");
  TD_CHECK(N > 0 && tdeed_bneck_fits(h, w, C, R), "bneck: geometry h=%d w=%d C=%d R=%d unsupported", h, w, C, R);
  TD_CHECK(!out2 || (n2 % 8 == 0 && n2 > 0 && n2 <= C), "bneck: bad second output width %d", n2);
  p = BneckP{};
  p.x = (const bf16_t*)x;
  p.w1f = (const bf16x8*)w1f; p.s1 = s1; p.h1 = h1;
  p.w2f = (const bf16x8*)w2f; p.s2 = s2; p.h2 = h2;
  p.se = SeP{};
  p.se.R = R; p.se.w1f = (const bf16x8*)se_w1f; p.se.b1 = se_b1; p.se.w2f = (const bf16x8*)se_w2f; p.se.b2 = se_b2;
  p.w3f = (const bf16x8*)w3f; p.s3 = s3; p.h3 = h3;
  p.out = (bf16_t*)out; p.out2 = (bf16_t*)out2; p.n2 = out2 ? n2 : 0;
  p.N = N; p.h = h; p.w = w; p.C = C;
  p.w2_tap_major = w2_tap_major ? 1 : 0;
  return TDEED_OK;
}

// 1 when the tap maps of a following gate-shift site of fold F can be made in the launch's tail (tdeed_bneck_gs_fwd's Q)
extern "C" int tdeed_bneck_qtail_fits(int h, int w, int C, int F) {
  if (F <= 0 || F % 4 != 0 || 2 * F > C) return 0;
  const int hw = h * w, fpw = bneck_fpw(hw), nch = (F + 7) / 8, KSc = (nch + 3) / 4, PSQ = (nch | 1) * 16;
  if (nch > (C + 31) / 32 || 4 * KSc * 64 > 2 * BNK_THR || fpw * hw * nch > 3 * BNK_THR) return 0;          // request slots per thread
  return (size_t)fpw * hw * (PSQ + 224) + (size_t)4 * KSc * 1024 <= (size_t)fpw * hw * bneck_rs(C) ? 1 : 0;
}

extern "C" int tdeed_bneck_fwd(const void* x, const void* G, int Fp, int N, int h, int w, int C, const void* w1f,
                               const float* s1, const float* h1, const void* w2f, const float* s2, const float* h2,
                               const void* se_w1f, const float* se_b1, const void* se_w2f, const float* se_b2, int R,
                               const void* w3f, const float* s3, const float* h3, void* out, void* out2, int n2,
                               int w2_tap_major, void* stream) {
  BneckP p;
  const int rc = bneck_fill(p, x, N, h, w, C, w1f, s1, h1, w2f, s2, h2, se_w1f, se_b1, se_w2f, se_b2, R, w3f, s3, h3, out, out2, n2,
                            w2_tap_major);
  if (rc != TDEED_OK) return rc;
  TD_CHECK(!G || (Fp % 8 == 0 && Fp > 0 && Fp <= C), "bneck: bad splice width %d", Fp);
  p.G = (const bf16_t*)G; p.Fp = G ? Fp : 0;
  return bneck_launch(p, (hipStream_t)stream);
}

// The same block behind a gate-shift-fuse site, with the site's LAST launch (tdeed_gsf_blend_src_fwd: fusion weights + blend,
// source channel order) done inside the frame load: gx [N][h*w][ldx] is what that launch would read (the block input, or the
// compact copy of its first Fp channels), gate / ysum / xsum are tdeed_gsf_gate_fwd's outputs for the N = B * T frames, T frames
// per clip.  out == tdeed_bneck_fwd(x, G = tdeed_gsf_blend_src_fwd(...)), bit for bit.
extern "C" int tdeed_bneck_gs_fwd(const void* x, const void* gx, int ldx, const float* gate, const float* ysum, const float* xsum,
                                  const float* cw1, const float* cb1, const float* cw2, const float* cb2, int T, int F, int Fp,
                                  int N, int h, int w, int C, const void* w1f, const float* s1, const float* h1, const void* w2f,
                                  const float* s2, const float* h2, const void* se_w1f, const float* se_b1, const void* se_w2f,
                                  const float* se_b2, int R, const void* w3f, const float* s3, const float* h3, void* out,
                                  void* out2, int n2, int w2_tap_major, const void* q_wqf, const float* q_bn, int q_F, float* Q,
                                  void* stream) {
  BneckP p;
  const int rc = bneck_fill(p, x, N, h, w, C, w1f, s1, h1, w2f, s2, h2, se_w1f, se_b1, se_w2f, se_b2, R, w3f, s3, h3, out, out2, n2,
                            w2_tap_major);
  if (rc != TDEED_OK) return rc;
  if (Q) {
    TD_CHECK(q_wqf && q_bn, "bneck_gs: tap maps asked for without fragments / BatchNorm table");
    TD_CHECK(tdeed_bneck_qtail_fits(h, w, C, q_F), "bneck_gs: the tap-map tail does not fit (h=%d w=%d C=%d F=%d)", h, w, C, q_F);
    p.qt.wqf = (const bf16x8*)q_wqf; p.qt.bn = q_bn; p.qt.Q = Q; p.qt.F = q_F;
  }
  TD_CHECK(gx && gate && ysum && xsum && cw1 && cb1 && cw2 && cb2, "bneck_gs: null pointer");
  TD_CHECK(F % 4 == 0 && F > 0 && Fp % 8 == 0 && Fp >= F && Fp < F + 8 && 2 * Fp <= C && ldx >= Fp && ldx % 8 == 0,
           "bneck_gs: bad fold F=%d Fp=%d ldx=%d C=%d", F, Fp, ldx, C);
  TD_CHECK(T > 0 && N % T == 0 && N < (1 << 22), "bneck_gs: %d frames are not whole clips of %d", N, T);
  const int hw = h * w, fpw = bneck_fpw(hw);
  TD_CHECK(hw >= 14 && fpw * hw * (Fp / 8) <= 3 * BNK_THR && fpw * Fp <= BNK_THR,
           "bneck_gs: a %d x %d map with a %d-channel slice does not fit the blend's scratch / piece slots", h, w, Fp);
  p.Fp = Fp;
  p.gs.x = (const bf16_t*)gx; p.gs.ldx = ldx; p.gs.gate = gate; p.gs.ysum = ysum; p.gs.xsum = xsum;
  p.gs.cw1 = cw1; p.gs.cb1 = cb1; p.gs.cw2 = cw2; p.gs.cb2 = cb2; p.gs.T_len = T; p.gs.F = F;
  return bneck_launch(p, (hipStream_t)stream);
}
