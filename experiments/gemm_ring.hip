// PARKED EXPERIMENT (round 2) -- not compiled into libtdeed_hip.so.
// LDS-DMA ring forms of the tiled 1x1 contraction (gemm.hip): the same 128 x BN tile, and a 256 x 128 tile with 8 waves, fed by
// global_load_lds_dwordx4 into a ring of 32-deep K slabs (counted vmcnt, one raw s_barrier per slab).  Both are correct
// (tests/test_gpu_ops.py -k gemm with TDEED_GEMM_RING=1|2 TDEED_GEMM_RING_MIN=1) and neither is faster on MI355X
// (tools/bench_gemm_ring.py, bf16, +bias+ReLU, us per launch: register-staged tile / 128-row ring NS=3 / 256x128 ring NS=3):
//   M=78400  K=784 N=784 : 163 / 166 / 182        M=39200 K=784 N=784 : 91 / 90 / 107
//   M=156800 K=320 N=320 :  70 /  69 /  --        M=39200 K=320 N=784 : 42 / 39 /  56
//   M=19600  K=368 N=368 : 17.7 / 17.5 / --       (a 6-stage ring at one workgroup per CU: 234 us on the first shape)
// i.e. the K loop of the simple one-barrier-per-slab structure is not bound by loads in flight, and a larger tile at one
// workgroup per CU loses more occupancy than it saves in operand re-reads -- as the programming guide's ladder says
// (128x128 best for simple loops; 256x256 only pays with the 8-phase hand-scheduled pipeline).  The pieces below plug into
// gemm.hip: the kernel in front of launch_gemm(), the dispatch block in front of TD_GEMM, gemm_tile_epilogue<T, BN, WR>
// being gemm_kernel's epilogue with the pass count / thread count as template parameters.

// ---- the same 128 x BN tile fed by an LDS-DMA ring (bf16, no per-frame operand scale) ------------------------------------
// gemm_kernel keeps ONE K slab per workgroup in flight (its register stage), so at 2-3 workgroups per CU the K loop runs at
// what 24 KB x occupancy of outstanding loads buys (~10 TB/s of L2 reads chip-wide, 1.5 us per 64-deep slab at M = 19600).
// Here the operands go global -> LDS without passing registers (global_load_lds_dwordx4: 64 lanes x 16 B = 1 KiB of
// contiguous LDS per wave-instruction), into a ring of NS stages of one 32-deep K slab each (128 + BN rows of 64 B), NS - 1
// slabs in flight per workgroup at all times; one raw s_barrier per slab, counted s_waitcnt vmcnt (the DMAs of the later
// slabs stay in flight across the barrier).  LDS rows are 64 B = four 16-byte slots; slot = k-chunk ^ ((row >> 1) & 3)
// makes the ds_read_b128 fragment reads conflict-free; the DMA writes lanes linearly, so the swizzle sits in the SOURCE
// address of each lane (lane l of a 16-row piece: row l >> 2, slot l & 3 -> k-chunk (l & 3) ^ ((l >> 3) & 3)).
// Rows past M / N and chunks past K are loaded from clamped (valid) addresses: rows past the edge are never stored, and the
// fragments of the K tail are zeroed in registers on both operands.
// s_waitcnt immediate (gfx9 encoding) that waits for vmcnt <= n only
constexpr int vm_wait(int n) { return 0x0F70 | (n & 15) | ((n >> 4) << 14); }

template <int BM, int BN, int NS>
__global__ __launch_bounds__(BM * 2, BM == 128 ? 2 : 1) void gemm_ring_kernel(const GemmP p) {
  typedef bf16_t T;
  constexpr int NT = BN / 32;
  constexpr int WR = BM / 64, NW = WR * 2;     // waves WR x 2, each 64 x BN/2
  constexpr int A_BYTES = BM * 64, B_BYTES = BN * 64, STAGE = A_BYTES + B_BYTES;
  constexpr int BP = (BN / 16) / NW;           // B pieces (16 rows x 64 B) per wave and slab
  static_assert(BP >= 1 && BP * NW * 16 == BN, "B pieces must divide over the waves");
  constexpr int DPW = 2 + BP;                  // DMA wave-instructions per wave and slab
  static_assert(NS * STAGE >= 64 * (BN + 4) * 4 && NS * STAGE >= (BM * 2 / (BN / 8)) * 2 * BN * 4, "the epilogue tile must fit the ring");
  extern __shared__ __attribute__((aligned(1024))) unsigned char lds[];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wid >> 1, wc = wid & 1;
  const int nb = (p.N + BN - 1) / BN;
  const long lid = xcd_logical_id(blockIdx.x, gridDim.x);
  const int tile_n = (int)(lid % nb);
  const long tile_m = lid / nb;
  const long m0 = tile_m * BM;
  const int n0 = tile_n * BN;

  // ---- DMA sources of this lane: rows (lane >> 2) of its wave's pieces, k-chunk = slot ^ swizzle(row)
  const int prow = lane >> 2;
  const int kch = ((lane & 3) ^ ((lane >> 3) & 3)) * 8;          // element offset of the lane's chunk inside a slab
  const T* arow[2];
  const T* a0row[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    long m = m0 + (wid * 2 + i) * 16 + prow;
    m = m < p.M ? m : (long)p.M - 1;
    long src = m;
    if (p.g_stride > 1) {
      const long per = (long)p.g_ho * p.g_wo;
      const long f = m / per;
      const int rem = (int)(m - f * per);
      const int yo = rem / p.g_wo, xo = rem - yo * p.g_wo;
      src = (f * p.g_hi + (long)yo * p.g_stride) * p.g_wi + (long)xo * p.g_stride;
    }
    arow[i] = reinterpret_cast<const T*>(p.A) + src * p.lda;
    a0row[i] = p.A0 ? reinterpret_cast<const T*>(p.A0) + src * p.lda0 : arow[i];
  }
  const T* brow[BP];
#pragma unroll
  for (int i = 0; i < BP; ++i) {
    int n = n0 + (wid * BP + i) * 16 + prow;
    n = n < p.N ? n : p.N - 1;
    brow[i] = reinterpret_cast<const T*>(p.W) + (long)n * p.ldw;
  }
  typedef __attribute__((address_space(3))) void* lds_ptr_t;
  typedef const __attribute__((address_space(1))) void* glb_ptr_t;
  auto issue = [&](int kt) {
    unsigned char* st = lds + (kt % NS) * STAGE;
    int k = kt * 32 + kch;
    k = k < p.K ? k : p.K - 8;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const T* src = (k < p.k0 ? a0row[i] : arow[i]) + k;
      __builtin_amdgcn_global_load_lds((glb_ptr_t)src, (lds_ptr_t)(st + (wid * 2 + i) * 1024), 16, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < BP; ++i)
      __builtin_amdgcn_global_load_lds((glb_ptr_t)(brow[i] + k), (lds_ptr_t)(st + A_BYTES + (wid * BP + i) * 1024), 16, 0, 0);
  };

  f32x4 acc[4][NT];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const int nkt = (p.K + 31) / 32;
  const int fr = lane & 15, fq = lane >> 4;
  const int kv = (p.K - (nkt - 1) * 32) >> 3;            // valid 16-byte chunks of the last slab (1..4)
  const int fslot = (fq ^ ((fr >> 1) & 3)) << 4;
  const int a_off = (wr * 64 + fr) * 64 + fslot;
  const int b_off = A_BYTES + (wc * (BN / 2) + fr) * 64 + fslot;

#pragma unroll
  for (int s_ = 0; s_ < NS - 1; ++s_)
    if (s_ < nkt) issue(s_);
  for (int kt = 0; kt < nkt; ++kt) {
    // this wave's DMAs of slab kt have landed once at most (slabs issued behind it) x DPW are still outstanding
    const int behind = min(NS - 2, nkt - 1 - kt);
    static_assert(NS >= 2 && NS <= 6, "ring depth");
    switch (behind) {
      case 4: __builtin_amdgcn_s_waitcnt(vm_wait(4 * DPW)); break;
      case 3: __builtin_amdgcn_s_waitcnt(vm_wait(3 * DPW)); break;
      case 2: __builtin_amdgcn_s_waitcnt(vm_wait(2 * DPW)); break;
      case 1: __builtin_amdgcn_s_waitcnt(vm_wait(DPW)); break;
      default: __builtin_amdgcn_s_waitcnt(vm_wait(0)); break;
    }
    __builtin_amdgcn_s_barrier();      // every wave's part of slab kt is in LDS; every wave is done reading slab kt - 1
    if (kt + NS - 1 < nkt) issue(kt + NS - 1);                   // into the stage slab kt - 1 occupied
    const unsigned char* st = lds + (kt % NS) * STAGE;
    bf16x8 af[4], bfr[NT];
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) af[mt] = *reinterpret_cast<const bf16x8*>(st + a_off + mt * 16 * 64);
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) bfr[nt] = *reinterpret_cast<const bf16x8*>(st + b_off + nt * 16 * 64);
    if (kt == nkt - 1 && fq >= kv) {
      const bf16x8 z = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
      for (int mt = 0; mt < 4; ++mt) af[mt] = z;
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) bfr[nt] = z;
    }
#pragma unroll
    for (int mt = 0; mt < 4; ++mt)
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[mt], bfr[nt], acc[mt][nt], 0, 0, 0);
  }
  __syncthreads();                     // the ring becomes the epilogue's fp32 tile
  gemm_tile_epilogue<T, BN, WR>(p, acc, lds, tile_m, m0, n0);
}


// ---- dispatch (inside launch_gemm<T>) ----
#if 0
  // bf16 without a per-frame operand scale: the LDS-DMA ring forms (TDEED_GEMM_RING: 0 register staging, 1 the same
  // 128-row tile fed by the ring, 2 256 x 128 tiles where the grid still fills the chip)
  if constexpr (sizeof(T) == 2) {
    static int ring = -1, ring_ns = 3;
    static long ring_min = 512;
    if (ring < 0) {
      const char* e = getenv("TDEED_GEMM_RING");
      ring = e ? atoi(e) : 0;
      const char* d = getenv("TDEED_GEMM_RING_NS");
      if (d) ring_ns = atoi(d);
      const char* m = getenv("TDEED_GEMM_RING_MIN");
      if (m) ring_min = atol(m);
    }
#define TD_RING(BMv, BNv, NSv, gridv)                                                                                  \
  do {                                                                                                                 \
    constexpr int bytes = NSv * (BMv * 64 + BNv * 64);                                                                 \
    static bool attr = false;                                                                                          \
    if (!attr) {                                                                                                       \
      (void)hipFuncSetAttribute((const void*)gemm_ring_kernel<BMv, BNv, NSv>, hipFuncAttributeMaxDynamicSharedMemorySize, \
                                bytes);                                                                                \
      attr = true;                                                                                                     \
    }                                                                                                                  \
    hipLaunchKernelGGL((gemm_ring_kernel<BMv, BNv, NSv>), dim3((unsigned)(gridv)), dim3(BMv * 2), bytes, st, p);       \
  } while (0)
    if (ring == 2 && !p.a_scale && bn == 128 && ((p.M + 255) / 256) * nb >= ring_min) {
      const long g2 = ((p.M + 255) / 256) * nb;
      if (ring_ns == 4) TD_RING(256, 128, 4, g2);
      else TD_RING(256, 128, 3, g2);
      TD_LAUNCH_CHECK("gemm_ring256");
      return TDEED_OK;
    }
    if (ring == 1 && !p.a_scale && bn >= 64) {
      if (bn == 128) {
        if (ring_ns == 4) TD_RING(128, 128, 4, grid);
        else TD_RING(128, 128, 3, grid);
      } else {
        if (ring_ns == 4) TD_RING(128, 64, 4, grid);
        else TD_RING(128, 64, 3, grid);
      }
      TD_LAUNCH_CHECK("gemm_ring");
      return TDEED_OK;
    }
#undef TD_RING
  }
#endif
