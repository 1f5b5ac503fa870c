// PARKED (round 3): the gate-shift(-fuse) module of a site as ONE launch -- a workgroup owns a chunk of TC consecutive
// frames of a clip and passes frames t0-3 .. t0+TC+2 through an LDS ring (BN + ReLU, conv3d taps on the MFMA pipe, gates,
// spatial sums, fusion conv, blend).  Correct: bit-level agreement class with the three-launch form and within the bf16
// tolerance of the reference's goldens for F = 16 / 40 / 92 and _GSM, chunks of 1, 2, 3 frames, map or compact-slice input
// (the test lived in tests/test_gpu_r3.py::test_one_launch_gate_shift_matches_the_reference_and_the_three_launch_form,
// 12 cases green on MI355X at git 2ad4b1e+).  MEASURED SLOWER on MI355X (cfg2, bf16, two sub-batches of 4 clips, graph
// replay): gate-shift family 1.61 ms per step vs 0.71 ms for the three launches, 2657 vs 3494 clips/s.  Why: (TC + 6) / TC
// = 4 passes per output frame at TC = 2, each pass a chain of six barriers (ring write, band build, MFMA, gate
// accumulation, tanh, spatial sums) in a workgroup that the 110-150 KB of LDS pins at ONE per CU = one wave per SIMD, so
// every LDS -> MFMA dependency is exposed; the three-launch form spreads the same frames over 400 one-pass workgroups,
// two to three per CU.  The same lesson as TDEED_GSF_MERGE (round 2): at these sizes serial work per workgroup costs more
// than the launch boundaries it removes.  To build: append to t-deed_amd/csrc/gsf.hip (uses its helpers) and declare
// tdeed_gsf_fused_chunk / tdeed_gsf_fused_fwd in include/tdeed_hip.h.
// =========================================================================== the whole module in ONE launch (bf16, small maps)
// Three launches per site (tap maps Q -> gates + spatial sums -> blend) are three dependent latency chains of ~6-13 us
// over a few hundred one-frame workgroups each.  For the small maps (14x14 x 40 channels, 7x7 x 92: ten of the eleven
// sites of RegNetY-200MF) a workgroup can own a CHUNK of TC consecutive frames of one clip and do everything itself:
//
//   frames t0-3 .. t0+TC+2 pass through the workgroup one after the other (raw channels -> an LDS ring slot, BN + ReLU
//   -> the zero-haloed MFMA band, implicit-GEMM tap products Q on the MFMA pipe exactly like gsf_q_mfma_kernel), each
//   pass adding its three temporal taps into the gate pre-activations of frames f+1, f, f-1 (LDS, fp32);
//   a frame's gate is final one pass later: tanh, then its spatial sums of gate*x and x out of the ring;
//   behind the last pass: the fusion conv over the (channel, time) plane for the TC output frames, and the blend +
//   shift + interleave of those frames out of the ring.
//
// The conv3d is evaluated (TC + 6) / TC times (2.5x at TC = 4; it is ~0.2 GFLOP per site) and the slice is read
// (TC + 6) / TC times from L2 instead of 3 + 3 + 3 partial reads; nothing but the input slice and the output crosses the
// launch boundary (no Q, gate, ysum, xsum tensors).  Ring slots: frames t0-1 .. t0+TC stay for the blend; the four edge
// frames share two slots (t0-3 / t0+TC+1 and t0-2 / t0+TC+2: their last uses never overlap).
struct GsfFusedP {
  const bf16_t* x;          // rows [N*hw][ldx], channels [0, Fp) used
  long ldx;
  int T_len, h, w, F, Fp, TC, nch, PSQ, KS;
  const float *bn_scale, *bn_shift;
  const bf16x8* wqf;
  const float *b3d, *cw1, *cb1, *cw2, *cb2;   // cw1 == nullptr: plain gate-shift (_GSM): out = y_shift + r
  bf16_t* out;              // [N*hw][Fp]
};

__global__ __launch_bounds__(256) void gsf_fused_kernel(const GsfFusedP P) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smf[];
  const int T_len = P.T_len, h = P.h, w = P.w, F = P.F, Fp = P.Fp, TC = P.TC, nch = P.nch, PSQ = P.PSQ, KS = P.KS;
  const int hw = h * w, WP = w + 2, Fh = F >> 1, Fq = F >> 2;
  const int NG = TC + 4;                                          // gate frames t0-2 .. t0+TC+1
  const int ncc = (T_len + TC - 1) / TC;
  const int b = blockIdx.x / ncc, t0 = (blockIdx.x - b * ncc) * TC;
  const int tc = min(TC, T_len - t0);
  bf16x8* wl = reinterpret_cast<bf16x8*>(smf);                                   // [KS][64]
  unsigned char* band = smf + (size_t)KS * 64 * 16;                             // [h+2][WP][PSQ]
  bf16_t* ring = reinterpret_cast<bf16_t*>(band + (size_t)(h + 2) * WP * PSQ);  // [TC+4][hw][Fp]
  float* pre = reinterpret_cast<float*>(ring + (size_t)(TC + 4) * hw * Fp);     // [NG][hw][2]
  float* ys = pre + (size_t)NG * hw * 2;                                        // [NG][F]  sum_p gate*x
  float* xsu = ys + NG * F;                                                     // [NG][F]  sum_p x
  float* part = xsu + NG * F;                                                   // [2][S][F], S*F = 512
  float* fwl = part + 1024;                                                     // [TC][F]
  float* sbn = fwl + TC * F;                                                    // [2F]
  float* cwl = sbn + 2 * F;                                                     // [40]
  int* soff = reinterpret_cast<int*>(cwl + 40);                                 // [KS*4]
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, pl = lane & 15, q = lane >> 4;
  const bool fuse = P.cw1 != nullptr;

  auto slot_of = [&](int tf) {                                    // ring slot of frame tf in [t0-3, t0+TC+2]
    const int d = tf - (t0 - 1);
    if (d >= 0 && d <= TC + 1) return d;
    return (tf == t0 - 3 || tf == t0 + TC + 1) ? TC + 2 : TC + 3;
  };
  const int npc = (hw * Fp) >> 3;                                 // 16-byte pieces of one frame's slice
  constexpr int MAXPC = 4;                                        // pieces per lane: hw*Fp <= 8192 elements
  const int tf_lo = max(0, t0 - 3), tf_hi = min(T_len - 1, t0 + TC + 2);
  const int g_lo = max(0, t0 - 2), g_hi = min(T_len - 1, t0 + TC + 1);
  const long cbase = (long)b * T_len;
  u32x4 pf[MAXPC];
  auto issue_frame = [&](int tf) {
    const bf16_t* xf = P.x + (cbase + tf) * hw * P.ldx;
    const int cpp = Fp >> 3;
    const IDiv dc(cpp);
#pragma unroll
    for (int u = 0; u < MAXPC; ++u) {
      const int i = min(tid + u * 256, npc - 1);
      int p, j;
      dc.divmod(i, p, j);
      pf[u] = *reinterpret_cast<const u32x4*>(xf + (long)p * P.ldx + j * 8);
    }
  };
  // ---- prologue: first frame's slice, weights, tables
  issue_frame(tf_lo);
  for (int s_ = tid; s_ < KS * 4; s_ += 256) {
    const int tap = s_ / nch, ck = s_ - tap * nch;
    const int dy = tap / 3, dx = tap - dy * 3;
    soff[s_] = tap < 9 ? (dy * WP + dx) * PSQ + ck * 16 : 0;
  }
  gsf_stage_bn(sbn, P.bn_scale, P.bn_shift, F);
  copy16_batched(reinterpret_cast<u32x4*>(wl), reinterpret_cast<const u32x4*>(P.wqf), KS * 64);
  if (fuse && tid < 38) cwl[tid] = *(tid < 18 ? P.cw1 + tid : tid < 36 ? P.cw2 + (tid - 18) : tid == 36 ? P.cb1 : P.cb2);
  {
    const float b0 = P.b3d[0], b1 = P.b3d[1];
    for (int i = tid; i < NG * hw * 2; i += 256) pre[i] = (i & 1) ? b1 : b0;
    for (int i = tid; i < 2 * NG * F; i += 256) ys[i] = 0.f;     // ys and xsu are adjacent: frames outside the clip stay 0
  }
  // the zero halo of the band is written once; the interior is rewritten by every pass
  for (int i = tid; i < (h + 2) * WP * (PSQ >> 4); i += 256) reinterpret_cast<u32x4*>(band)[i] = (u32x4){0u, 0u, 0u, 0u};

  const int ntl = (hw + 15) >> 4;
  const IDiv dw_(w), dnch(nch);
  // gate of frame tg is final: tanh, then its spatial sums out of the ring
  auto finalize = [&](int tg) {
    const int gs = tg - (t0 - 2);
    float* pg = pre + (size_t)gs * hw * 2;
    for (int i = tid; i < 2 * hw; i += 256) pg[i] = tanhf(pg[i]);
    __syncthreads();
    const bf16_t* xr = ring + (size_t)slot_of(tg) * hw * Fp;
    const int nq = F >> 1, S = 256 / nq;
    const int cp = tid % nq, s = tid / nq;
    if (s < S) {
      const int g = (2 * cp) >= Fh;
      float y0 = 0.f, y1 = 0.f, x0 = 0.f, x1 = 0.f;
      for (int p = s; p < hw; p += S) {
        float v0, v1;
        Pair<bf16_t>::load(xr + (long)p * Fp + 2 * cp, v0, v1);
        const float gt = pg[2 * p + g];
        x0 += v0; x1 += v1;
        y0 = fmaf(v0, gt, y0); y1 = fmaf(v1, gt, y1);
      }
      part[s * F + 2 * cp] = y0;
      part[s * F + 2 * cp + 1] = y1;
      part[(S + s) * F + 2 * cp] = x0;
      part[(S + s) * F + 2 * cp + 1] = x1;
    }
    __syncthreads();
    for (int c = tid; c < F; c += 256) {
      float a = 0.f, bq = 0.f;
      for (int s2 = 0; s2 < S; ++s2) {
        a += part[s2 * F + c];
        bq += part[(S + s2) * F + c];
      }
      ys[gs * F + c] = a;
      xsu[gs * F + c] = bq;
    }
    __syncthreads();
  };

  for (int tf = tf_lo; tf <= tf_hi; ++tf) {
    // (1) this frame's slice: registers -> ring slot; (2) request the next frame
    bf16_t* rs = ring + (size_t)slot_of(tf) * hw * Fp;
#pragma unroll
    for (int u = 0; u < MAXPC; ++u) {
      const int i = tid + u * 256;
      if (i < npc) reinterpret_cast<u32x4*>(rs)[i] = pf[u];
    }
    if (tf < tf_hi) issue_frame(tf + 1);
    __syncthreads();
    // (3) band interior = relu(bn(slice)), 8-channel chunks; channels >= F are zero
    for (int i = tid; i < hw * nch; i += 256) {
      int p, j;
      dnch.divmod(i, p, j);
      int py, px;
      dw_.divmod(p, py, px);
      const bf16x8 r8 = *reinterpret_cast<const bf16x8*>(rs + (long)p * Fp + j * 8);
      bf16x8 o8;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int c = min(j * 8 + e, F - 1);
        o8[e] = (j * 8 + e < F) ? (bf16_t)fmaxf(fmaf((float)r8[e], sbn[c], sbn[F + c]), 0.f) : (bf16_t)0.f;
      }
      *reinterpret_cast<bf16x8*>(band + ((long)(py + 1) * WP + px + 1) * PSQ + j * 16) = o8;
    }
    __syncthreads();
    // (4) tap products of this frame, added to the gate pre-activations of frames tf+1 (tap 0), tf (1), tf-1 (2)
    for (int mt = wv; mt < ntl; mt += 4) {
      const int p = mt * 16 + pl;
      const bool pok = p < hw;
      const int pc = pok ? p : 0;
      int py, px;
      dw_.divmod(pc, py, px);
      const unsigned char* base = band + ((long)py * WP + px) * PSQ;
      f32x4 acc = {0.f, 0.f, 0.f, 0.f};
      for (int ks = 0; ks < KS; ++ks) {
        const bf16x8 af = *reinterpret_cast<const bf16x8*>(base + soff[ks * 4 + q]);
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wl[ks * 64 + lane], af, acc, 0, 0, 0);
      }
      if (pok) {
        if (q == 0) {
          if (tf + 1 >= g_lo && tf + 1 <= g_hi) {
            float* d = pre + ((size_t)(tf + 1 - (t0 - 2)) * hw + p) * 2;
            d[0] += acc[0]; d[1] += acc[1];
          }
          if (tf >= g_lo && tf <= g_hi) {
            float* d = pre + ((size_t)(tf - (t0 - 2)) * hw + p) * 2;
            d[0] += acc[2]; d[1] += acc[3];
          }
        } else if (q == 1) {
          if (tf - 1 >= g_lo && tf - 1 <= g_hi) {
            float* d = pre + ((size_t)(tf - 1 - (t0 - 2)) * hw + p) * 2;
            d[0] += acc[0]; d[1] += acc[1];
          }
        }
      }
    }
    __syncthreads();
    // (5) frame tf-1 has all three taps now (and so has tf itself at the end of the clip)
    if (tf - 1 >= g_lo && tf - 1 <= g_hi) finalize(tf - 1);
    if (tf == T_len - 1 && tf >= g_lo && tf <= g_hi) finalize(tf);
  }

  // ---- fusion weights of the output frames: 3x3 conv over the (channel, time) plane of the spatial means + sigmoid
  const float inv_hw = 1.0f / (float)hw;
  if (fuse) {
    for (int i = tid; i < tc * F; i += 256) {
      const int ti = i / F, c = i - ti * F;
      const int t = t0 + ti;
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
          const int gs2 = t2 - (t0 - 2);
          const float rm = (xsu[gs2 * F + cc] - ys[gs2 * F + cc]) * inv_hw;
          const int ts = g ? t2 - 1 : t2 + 1;
          const float ysh = (ts >= 0 && ts < T_len) ? ys[(ts - (t0 - 2)) * F + cc] * inv_hw : 0.f;
          a = fmaf(cw[(dc + 1) * 3 + (dt + 1)], ysh, a);
          a = fmaf(cw[9 + (dc + 1) * 3 + (dt + 1)], rm, a);
        }
      }
      fwl[i] = sigmoidf_(a);
    }
    __syncthreads();
  }
  // ---- blend + shift + interleave of the output frames, out of the ring
  const int nqd = Fp >> 2;
  const IDiv dqd(nqd);
  for (int ti = 0; ti < tc; ++ti) {
    const int t = t0 + ti;
    const bool has_next = t < T_len - 1, has_prev = t > 0;
    const bf16_t* xc = ring + (size_t)slot_of(t) * hw * Fp;
    const bf16_t* xn = ring + (size_t)slot_of(has_next ? t + 1 : t) * hw * Fp;
    const bf16_t* xp = ring + (size_t)slot_of(has_prev ? t - 1 : t) * hw * Fp;
    const float* gc = pre + (size_t)(t - (t0 - 2)) * hw * 2;
    const float* gn = pre + (size_t)((has_next ? t + 1 : t) - (t0 - 2)) * hw * 2;
    const float* gp = pre + (size_t)((has_prev ? t - 1 : t) - (t0 - 2)) * hw * 2;
    const float* fw = fwl + ti * F;
    bf16_t* dst = P.out + (cbase + t) * hw * Fp;
    for (int idx = tid; idx < hw * nqd; idx += 256) {
      int p, qd;
      dqd.divmod(idx, p, qd);
      bf16x4 o;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int co = qd * 4 + e;
        if (co >= F) { o[e] = xc[(long)p * Fp + co]; continue; }
        const int g = co >= Fh;
        const int col = co - g * Fh;
        const int ci = g * Fh + (col & 1) * Fq + (col >> 1);
        const float xv = (float)xc[(long)p * Fp + ci];
        const float r = xv - gc[2 * p + g] * xv;
        float ysh;
        if (g) ysh = has_prev ? gp[2 * p + 1] * (float)xp[(long)p * Fp + ci] : 0.f;
        else ysh = has_next ? gn[2 * p] * (float)xn[(long)p * Fp + ci] : 0.f;
        if (fuse) {
          const float wv_ = fw[ci];
          o[e] = (bf16_t)(ysh * wv_ + r * (1.0f - wv_));
        } else {
          o[e] = (bf16_t)(ysh + r);
        }
      }
      *reinterpret_cast<bf16x4*>(dst + (long)p * Fp + qd * 4) = o;
    }
  }
}

static size_t gsf_fused_smem(int h, int w, int F, int Fp, int TC, int PSQ, int KS) {
  const int hw = h * w, NG = TC + 4;
  return (size_t)KS * 64 * 16 + (size_t)(h + 2) * (w + 2) * PSQ + (size_t)(TC + 4) * hw * Fp * 2 +
         ((size_t)NG * hw * 2 + 2 * NG * F + 1024 + TC * F + 2 * F + 40 + 4 * KS) * sizeof(float);
}

static void gsf_fused_geom(int F, int* nch, int* PSQ, int* KS) {
  *nch = (F + 7) / 8;
  int ps16 = *nch + 1;
  if ((ps16 & 1) == 0) ++ps16;
  *PSQ = ps16 * 16;
  *KS = (9 * *nch + 3) / 4;
}

// frames per workgroup of the one-launch form for a site, 0 when it does not serve it: bf16, the slice (hw * Fp elements)
// within the prefetch registers, the smallest chunk that keeps the grid at or below ~1.25 workgroups per CU and fits 150 KB
extern "C" int tdeed_gsf_fused_chunk(int B, int T, int h, int w, int F, int Fp) {
  if (F % 4 != 0 || Fp % 8 != 0 || Fp < F || F > 256 || (long)h * w * Fp > 8192) return 0;
  if (256 / (F / 2) < 1 || 2 * (256 / (F / 2)) * F > 1024) return 0;
  int nch, PSQ, KS;
  gsf_fused_geom(F, &nch, &PSQ, &KS);
  int best = 0;
  for (int TC = 1; TC <= 8; ++TC) {
    if (gsf_fused_smem(h, w, F, Fp, TC, PSQ, KS) > 150 * 1024) break;
    best = TC;
    if ((long)B * ((T + TC - 1) / TC) <= 320) break;
  }
  static const int force_tc = getenv("TDEED_GSF_FUSED_TC") ? atoi(getenv("TDEED_GSF_FUSED_TC")) : 0;
  if (force_tc > 0 && force_tc <= 8 && gsf_fused_smem(h, w, F, Fp, force_tc, PSQ, KS) <= 150 * 1024) return force_tc;
  return best;
}

// x: rows [B*T*h*w][ldx] (a channels-last map, or the compact [.][Fp] slice a producer wrote); out [B*T*h*w][Fp].
// cw1 == NULL: the plain gate-shift module (_GSM).  wqf: engine.pack_gsf_q_frags.
extern "C" int tdeed_gsf_fused_fwd(const void* x, long ldx, int B, int T, int h, int w, int F, int Fp, const float* bn_scale,
                                   const float* bn_shift, const void* wqf, const float* b3d, const float* cw1,
                                   const float* cb1, const float* cw2, const float* cb2, void* out, void* stream) {
  TD_CHECK(x && bn_scale && bn_shift && wqf && b3d && out, "gsf_fused: null pointer");
  TD_CHECK(!cw1 || (cb1 && cw2 && cb2), "gsf_fused: the fusion conv needs both weights and both biases");
  TD_CHECK(B > 0 && T > 0 && ldx >= Fp && ldx % 8 == 0, "gsf_fused: bad sizes");
  const int TC = tdeed_gsf_fused_chunk(B, T, h, w, F, Fp);
  TD_CHECK(TC > 0, "gsf_fused: site h=%d w=%d F=%d Fp=%d not served (tdeed_gsf_fused_chunk)", h, w, F, Fp);
  GsfFusedP P;
  P.x = (const bf16_t*)x; P.ldx = ldx; P.T_len = T; P.h = h; P.w = w; P.F = F; P.Fp = Fp; P.TC = TC;
  gsf_fused_geom(F, &P.nch, &P.PSQ, &P.KS);
  P.bn_scale = bn_scale; P.bn_shift = bn_shift; P.wqf = (const bf16x8*)wqf; P.b3d = b3d;
  P.cw1 = cw1; P.cb1 = cb1; P.cw2 = cw2; P.cb2 = cb2; P.out = (bf16_t*)out;
  const size_t smem = gsf_fused_smem(h, w, F, Fp, TC, P.PSQ, P.KS);
  static bool attr = false;
  if (!attr) {
    if (hipFuncSetAttribute((const void*)gsf_fused_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) {
      tdeed_set_error("gsf_fused: hipFuncSetAttribute failed");
      return TDEED_ERR_RUNTIME;
    }
    attr = true;
  }
  hipLaunchKernelGGL(gsf_fused_kernel, dim3(B * ((T + TC - 1) / TC)), dim3(256), smem, (hipStream_t)stream, P);
  TD_LAUNCH_CHECK("gsf_fused");
  return TDEED_OK;
}
