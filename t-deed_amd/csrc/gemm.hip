// Dense 1x1 contraction on the MFMA pipe (gfx950).
//
//   C[m][n] = act((sum_k A'[m][k] W[n][k]) * scale[n] + shift[n] + R[m][n])
//
// Both operands are K-contiguous ([M][K] activations channels-last, [N][K] weights), so one
// 16-byte chunk per lane is exactly an MFMA operand fragment:
//   bf16: v_mfma_f32_16x16x32_bf16, lane l holds A[row l&15][k = 8(l>>4)+j], j<8
//   f32 : 4 x v_mfma_f32_16x16x4_f32 on the 4 floats of the chunk (same k assignment for A and B,
//         so the k permutation is harmless); exact f32 FMA chains -> parity mode.
// Tile: 128 (M) x BN (N) per 256-thread block, K in slabs of 128 bytes per row (64 bf16 / 32 f32),
// waves 2x2, each 64 x BN/2.  LDS rows are 128 B, 16-B chunks XOR-swizzled with (row & 7) so the
// ds_read_b128 fragment reads are conflict-free; double buffered, register-staged global loads
// (the staging pass is where the SE gate / gate-shift splice / stride-2 row gather are applied).
// Epilogue goes through LDS so that C (and the residual) move as whole 16-B chunks per lane.
#include "common.h"

struct GemmP {
  const void* A; long lda;
  const void* A0; long lda0; int k0;
  const float* a_scale; int a_scale_rows;
  int M, K, N;
  const void* W; long ldw;
  const float* scale; const float* shift;
  const void* R; long ldr;
  int act;
  void* C; long ldc;
  int g_stride, g_hi, g_wi, g_ho, g_wo;
};

template <typename T> struct Frag;
template <> struct Frag<bf16_t> { typedef bf16x8 type; };
template <> struct Frag<float> { typedef f32x4 type; };

template <typename T>
__device__ __forceinline__ f32x4 mma(const typename Frag<T>::type& a, const typename Frag<T>::type& b, f32x4 c);
template <>
__device__ __forceinline__ f32x4 mma<bf16_t>(const bf16x8& a, const bf16x8& b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
}
template <>
__device__ __forceinline__ f32x4 mma<float>(const f32x4& a, const f32x4& b, f32x4 c) {
#pragma unroll
  for (int j = 0; j < 4; ++j) c = __builtin_amdgcn_mfma_f32_16x16x4f32(a[j], b[j], c, 0, 0, 0);
  return c;
}

__device__ __forceinline__ int swz(int row, int chunk) { return row * 128 + ((chunk ^ (row & 7)) << 4); }

template <typename T, int BN>
__global__ __launch_bounds__(256) void gemm_kernel(const GemmP p) {
  constexpr int EPC = Chunk<T>::N;        // elements per 16-B chunk
  constexpr int KT = 8 * EPC;             // elements of K per slab
  constexpr int NT = BN / 32;             // 16-wide column tiles per wave
  constexpr int BROWS = BN / 32;          // B rows staged per thread
  constexpr int A_BYTES = 128 * 128, B_BYTES = BN * 128;
  constexpr int STAGE = A_BYTES + B_BYTES;
  constexpr int CS_LD = BN + 4;
  constexpr int LDS_BYTES = (2 * STAGE > 64 * CS_LD * 4) ? 2 * STAGE : 64 * CS_LD * 4;
  __shared__ __attribute__((aligned(16))) unsigned char lds[LDS_BYTES];
  typedef typename Frag<T>::type frag_t;

  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int wr = wid >> 1, wc = wid & 1;
  const int nb = (p.N + BN - 1) / BN;
  const int tile_n = blockIdx.x % nb;
  const long tile_m = blockIdx.x / nb;
  const long m0 = tile_m * 128;
  const int n0 = tile_n * BN;

  // ---- per-thread staging assignment: chunk = tid&7, rows (tid>>3) + 32*i
  const int ch = tid & 7;
  const int r0 = tid >> 3;
  const T* arow[4];
  const T* a0row[4];
  const float* srow[4];
  bool aok[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    long m = m0 + r0 + 32 * i;
    aok[i] = m < p.M;
    long mm = aok[i] ? m : 0;
    long src = mm;
    if (p.g_stride > 1) {
      long per = (long)p.g_ho * p.g_wo;
      long f = mm / per;
      int rem = (int)(mm - f * per);
      int yo = rem / p.g_wo, xo = rem - yo * p.g_wo;
      src = (f * p.g_hi + (long)yo * p.g_stride) * p.g_wi + (long)xo * p.g_stride;
    }
    arow[i] = reinterpret_cast<const T*>(p.A) + src * p.lda;
    a0row[i] = p.A0 ? reinterpret_cast<const T*>(p.A0) + src * p.lda0 : nullptr;
    srow[i] = p.a_scale ? p.a_scale + (mm / p.a_scale_rows) * (long)p.K : nullptr;
  }
  const T* brow[BROWS];
  bool bok[BROWS];
#pragma unroll
  for (int i = 0; i < BROWS; ++i) {
    int n = n0 + r0 + 32 * i;
    bok[i] = n < p.N;
    brow[i] = reinterpret_cast<const T*>(p.W) + (long)(bok[i] ? n : 0) * p.ldw;
  }

  u32x4 areg[4], breg[BROWS];
  const u32x4 zero4 = {0u, 0u, 0u, 0u};

  auto gload = [&](int kt) {
    const int k = kt * KT + ch * EPC;
    const bool kok = k < p.K;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      if (kok && aok[i]) {
        const T* src = (a0row[i] && k < p.k0) ? a0row[i] + k : arow[i] + k;
        areg[i] = *reinterpret_cast<const u32x4*>(src);
      } else {
        areg[i] = zero4;
      }
    }
#pragma unroll
    for (int i = 0; i < BROWS; ++i)
      breg[i] = (kok && bok[i]) ? *reinterpret_cast<const u32x4*>(brow[i] + k) : zero4;
    if (p.a_scale && kok) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        if (aok[i]) {
          float v[EPC];
          Chunk<T>::load(reinterpret_cast<const T*>(&areg[i]), v);
#pragma unroll
          for (int e = 0; e < EPC; ++e) v[e] *= srow[i][k + e];
          Chunk<T>::store(reinterpret_cast<T*>(&areg[i]), v);
        }
      }
    }
  };
  auto lstore = [&](int buf) {
    unsigned char* base = lds + buf * STAGE;
#pragma unroll
    for (int i = 0; i < 4; ++i) *reinterpret_cast<u32x4*>(base + swz(r0 + 32 * i, ch)) = areg[i];
#pragma unroll
    for (int i = 0; i < BROWS; ++i)
      *reinterpret_cast<u32x4*>(base + A_BYTES + swz(r0 + 32 * i, ch)) = breg[i];
  };

  f32x4 acc[4][NT];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const int nkt = (p.K + KT - 1) / KT;
  gload(0);
  lstore(0);
  __syncthreads();
  const int fr = lane & 15, fq = lane >> 4;
  for (int kt = 0; kt < nkt; ++kt) {
    const int cur = kt & 1;
    if (kt + 1 < nkt) gload(kt + 1);
    const unsigned char* abase = lds + cur * STAGE;
    const unsigned char* bbase = abase + A_BYTES;
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      frag_t af[4], bfr[NT];
#pragma unroll
      for (int mt = 0; mt < 4; ++mt)
        af[mt] = *reinterpret_cast<const frag_t*>(abase + swz(wr * 64 + mt * 16 + fr, 4 * s + fq));
#pragma unroll
      for (int nt = 0; nt < NT; ++nt)
        bfr[nt] = *reinterpret_cast<const frag_t*>(bbase + swz(wc * (BN / 2) + nt * 16 + fr, 4 * s + fq));
#pragma unroll
      for (int mt = 0; mt < 4; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = mma<T>(af[mt], bfr[nt], acc[mt][nt]);
    }
    if (kt + 1 < nkt) lstore(cur ^ 1);
    __syncthreads();
  }

  // ---- epilogue: two passes of 64 rows through an fp32 LDS tile
  float* cs = reinterpret_cast<float*>(lds);
  float sc[NT], sh[NT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) {
    int n = n0 + wc * (BN / 2) + nt * 16 + fr;
    bool ok = n < p.N;
    sc[nt] = (p.scale && ok) ? p.scale[n] : 1.0f;
    sh[nt] = (p.shift && ok) ? p.shift[n] : 0.0f;
  }
  constexpr int CPR = BN / EPC;   // chunks per tile row
  for (int half = 0; half < 2; ++half) {
    if (wr == half) {
#pragma unroll
      for (int mt = 0; mt < 4; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            int row = mt * 16 + fq * 4 + r;
            int col = wc * (BN / 2) + nt * 16 + fr;
            cs[row * CS_LD + col] = acc[mt][nt][r] * sc[nt] + sh[nt];
          }
    }
    __syncthreads();
    for (int idx = tid; idx < 64 * CPR; idx += 256) {
      int row = idx / CPR, cj = idx - row * CPR;
      long m = m0 + half * 64 + row;
      int n = n0 + cj * EPC;
      if (m < p.M && n < p.N) {
        float v[EPC];
#pragma unroll
        for (int e = 0; e < EPC; ++e) v[e] = cs[row * CS_LD + cj * EPC + e];
        if (p.R) {
          float rv[EPC];
          Chunk<T>::load(reinterpret_cast<const T*>(p.R) + m * p.ldr + n, rv);
#pragma unroll
          for (int e = 0; e < EPC; ++e) v[e] += rv[e];
        }
        if (p.act == TDEED_ACT_RELU) {
#pragma unroll
          for (int e = 0; e < EPC; ++e) v[e] = fmaxf(v[e], 0.f);
        } else if (p.act == TDEED_ACT_GELU) {
#pragma unroll
          for (int e = 0; e < EPC; ++e) v[e] = gelu_erf(v[e]);
        }
        Chunk<T>::store(reinterpret_cast<T*>(p.C) + m * p.ldc + n, v);
      }
    }
    __syncthreads();
  }
}

template <typename T>
static int launch_gemm(const GemmP& p, hipStream_t st) {
  const long mb = (p.M + 127) / 128;
  // column tile: least padded MFMA work, narrower tiles charged for their extra A re-reads /
  // LDS traffic; then shrink while the grid would leave most of the 256 CUs idle.
  int bn = 128;
  {
    const int cand[3] = {128, 64, 32};
    const float pen[3] = {1.0f, 1.1f, 1.3f};
    float best = 1e30f;
    for (int i = 0; i < 3; ++i) {
      float c = (float)(((p.N + cand[i] - 1) / cand[i]) * cand[i]) * pen[i];
      if (c < best) { best = c; bn = cand[i]; }
    }
    while (bn > 32 && mb * ((p.N + bn - 1) / bn) < 384) bn >>= 1;
  }
  const long nb = (p.N + bn - 1) / bn;
  const long grid = mb * nb;
  if (grid > 0x7fffffffL) { tdeed_set_error("gemm: grid too large"); return TDEED_ERR_ARG; }
  if (bn == 32) hipLaunchKernelGGL((gemm_kernel<T, 32>), dim3((unsigned)grid), dim3(256), 0, st, p);
  else if (bn == 64) hipLaunchKernelGGL((gemm_kernel<T, 64>), dim3((unsigned)grid), dim3(256), 0, st, p);
  else hipLaunchKernelGGL((gemm_kernel<T, 128>), dim3((unsigned)grid), dim3(256), 0, st, p);
  TD_LAUNCH_CHECK("gemm");
  return TDEED_OK;
}

extern "C" int tdeed_gemm_fwd(const void* A, long lda, const void* A0, long lda0, int k0,
                              const float* a_scale, int a_scale_rows, int M, int K, int N,
                              const void* W, long ldw, const float* scale, const float* shift,
                              const void* R, long ldr, int act, void* C, long ldc,
                              int gather_stride, int gather_hi, int gather_wi, int gather_ho,
                              int gather_wo, int dtype, void* stream) {
  TD_CHECK(A && W && C, "gemm: null pointer");
  TD_CHECK(M > 0 && K > 0 && N > 0, "gemm: bad sizes M=%d K=%d N=%d", M, K, N);
  TD_CHECK(dtype == TDEED_F32 || dtype == TDEED_BF16, "gemm: bad dtype %d", dtype);
  const int q = 8;
  TD_CHECK(K % q == 0 && N % q == 0 && lda % q == 0 && ldw % q == 0 && ldc % q == 0,
           "gemm: K=%d N=%d lda=%ld ldw=%ld ldc=%ld must be multiples of 8", K, N, lda, ldw, ldc);
  TD_CHECK(!A0 || (k0 % q == 0 && lda0 % q == 0 && k0 <= K), "gemm: bad splice k0=%d lda0=%ld", k0, lda0);
  TD_CHECK(!R || ldr % q == 0, "gemm: ldr=%ld must be a multiple of 8", ldr);
  TD_CHECK(!a_scale || a_scale_rows > 0, "gemm: a_scale_rows must be > 0");
  TD_CHECK(act >= 0 && act <= 2, "gemm: bad act %d", act);
  if (gather_stride > 1)
    TD_CHECK(gather_hi > 0 && gather_wi > 0 && gather_ho > 0 && gather_wo > 0 && M % (gather_ho * gather_wo) == 0,
             "gemm: bad gather geometry");
  GemmP p;
  p.A = A; p.lda = lda; p.A0 = A0; p.lda0 = lda0; p.k0 = A0 ? k0 : 0;
  p.a_scale = a_scale; p.a_scale_rows = a_scale_rows > 0 ? a_scale_rows : 1;
  p.M = M; p.K = K; p.N = N; p.W = W; p.ldw = ldw; p.scale = scale; p.shift = shift;
  p.R = R; p.ldr = ldr; p.act = act; p.C = C; p.ldc = ldc;
  p.g_stride = gather_stride; p.g_hi = gather_hi; p.g_wi = gather_wi; p.g_ho = gather_ho; p.g_wo = gather_wo;
  hipStream_t st = (hipStream_t)stream;
  return dtype == TDEED_F32 ? launch_gemm<float>(p, st) : launch_gemm<bf16_t>(p, st);
}
