// Shared device/host helpers for the T-DEED gfx950 kernels.
// gfx950 only: wave64, MFMA 16x16x32 bf16 / 16x16x4 f32, 160 KiB LDS per CU.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/tdeed_hip.h"

typedef __bf16 bf16_t;
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

#define WAVE 64

// --------------------------------------------------------------------------- host side error plumbing
void tdeed_set_error(const char* fmt, ...);
#define TD_CHECK(cond, ...)                                   \
  do {                                                        \
    if (!(cond)) {                                            \
      tdeed_set_error(__VA_ARGS__);                           \
      return TDEED_ERR_ARG;                                   \
    }                                                         \
  } while (0)
#define TD_LAUNCH_CHECK(name)                                                     \
  do {                                                                            \
    hipError_t e__ = hipGetLastError();                                           \
    if (e__ != hipSuccess) {                                                      \
      tdeed_set_error("%s: launch failed: %s", name, hipGetErrorString(e__));     \
      return TDEED_ERR_LAUNCH;                                                    \
    }                                                                             \
  } while (0)

// Once-per-DEVICE guard of the launchers' hipFuncSetAttribute(MaxDynamicSharedMemorySize) calls: the attribute belongs
// to the (function, device) pair, so a process that drives a second GPU sets it there again.
struct TdDevOnce {
  bool done[32] = {};
  static int dev() { int d = 0; (void)hipGetDevice(&d); return d & 31; }
  bool get() const { return done[dev()]; }
  void set() { done[dev()] = true; }
};

// --------------------------------------------------------------------------- debug flavour (python t-deed_amd/build.py --debug)
// TD_DEV_ASSERT / TD_LDS_CHECK compile to nothing in the release library.  The debug library (-O1 -g -DTDEED_DEBUG=1,
// csrc/libtdeed_hip_dbg.so, loaded with TDEED_LIB_FLAVOUR=debug) traps the wave at the first violated condition: LDS
// offsets against the bytes the launch asked for, tile / frame indices against their tables (the reference
// has no race / bounds tooling of its own; GPU AddressSanitizer is not available on this pool).
#if defined(TDEED_DEBUG) && TDEED_DEBUG
#define TD_DEV_ASSERT(cond)                                                                              \
  do {                                                                                                   \
    if (!(cond)) {                                                                                       \
      printf("tdeed device assert %s:%d: %s (block %d thread %d)\n", __FILE__, __LINE__, #cond, (int)blockIdx.x, \
             (int)threadIdx.x);                                                                          \
      __builtin_trap();                                                                                  \
    }                                                                                                    \
  } while (0)
#else
#define TD_DEV_ASSERT(cond) do { } while (0)
#endif
// byte offset `off` (+ `len` bytes) inside a dynamic LDS allocation of `limit` bytes
#define TD_LDS_CHECK(off, len, limit) TD_DEV_ASSERT((long)(off) >= 0 && (long)(off) + (long)(len) <= (long)(limit))

// --------------------------------------------------------------------------- element access (T = float | bf16_t)
template <typename T> struct Elem;
template <> struct Elem<float> {
  static constexpr int PER16 = 4;   // elements in a 16-byte chunk
  static __device__ __forceinline__ float ld(const float* p) { return *p; }
  static __device__ __forceinline__ void st(float* p, float v) { *p = v; }
};
template <> struct Elem<bf16_t> {
  static constexpr int PER16 = 8;
  static __device__ __forceinline__ float ld(const bf16_t* p) { return (float)*p; }
  static __device__ __forceinline__ void st(bf16_t* p, float v) { *p = (bf16_t)v; }
};

// value as it will be read back after a store in T
template <typename T> __device__ __forceinline__ float round_to(float v);
template <> __device__ __forceinline__ float round_to<float>(float v) { return v; }
template <> __device__ __forceinline__ float round_to<bf16_t>(float v) { return (float)(bf16_t)v; }

// 16-byte chunk <-> fp32 registers
template <typename T> struct Chunk;
template <> struct Chunk<float> {
  static constexpr int N = 4;
  static __device__ __forceinline__ void load(const float* p, float (&v)[4]) {
    f32x4 t = *reinterpret_cast<const f32x4*>(p);
    v[0] = t[0]; v[1] = t[1]; v[2] = t[2]; v[3] = t[3];
  }
  static __device__ __forceinline__ void store(float* p, const float (&v)[4]) {
    f32x4 t = {v[0], v[1], v[2], v[3]};
    *reinterpret_cast<f32x4*>(p) = t;
  }
};
template <> struct Chunk<bf16_t> {
  static constexpr int N = 8;
  static __device__ __forceinline__ void load(const bf16_t* p, float (&v)[8]) {
    bf16x8 t = *reinterpret_cast<const bf16x8*>(p);
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = (float)t[i];
  }
  static __device__ __forceinline__ void store(bf16_t* p, const float (&v)[8]) {
    bf16x8 t;
#pragma unroll
    for (int i = 0; i < 8; ++i) t[i] = (bf16_t)v[i];
    *reinterpret_cast<bf16x8*>(p) = t;
  }
};

// --------------------------------------------------------------------------- wave / block reductions
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}
// sum over a block of NW waves; every thread gets the result. scratch: >= NW floats of LDS.
template <int NW>
__device__ __forceinline__ float block_sum(float v, float* scratch) {
  v = wave_sum(v);
  const int w = threadIdx.x >> 6;
  __syncthreads();
  if ((threadIdx.x & 63) == 0) scratch[w] = v;
  __syncthreads();
  float r = 0.f;
#pragma unroll
  for (int i = 0; i < NW; ++i) r += scratch[i];
  return r;
}

__device__ __forceinline__ float gelu_erf(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f)); }
__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + expf(-x)); }
// the same on the hardware's exp2 / reciprocal (1 ulp each; saturates to exactly 0 / 1): ~4 instructions against ~40 for libm's
// expf + IEEE division.  For the bf16 throughput path's SE gates, where both forms of the excitation (se_gate_mfma_kernel and
// the one inside the one-launch bottleneck) use it, so they stay bit-identical to each other; the fp32 parity path keeps
// sigmoidf_.  (Round 6: 12 gates per lane were ~1.9 k of the bottleneck workgroup's 76 k cycles.)
__device__ __forceinline__ float sigmoid_fast_(float x) {
  return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.44269504088896341f * x));
}

// XCD-aware workgroup renumbering (bijective): hardware deals workgroups round-robin over the 8 XCDs; this maps
// blockIdx -> logical id so that consecutive logical ids sit on ONE XCD (one L2) and run close in time.  Used
// wherever neighbouring work items re-read the same bytes (GEMM N tiles, conv halo rows).  Speed only.
__device__ __forceinline__ long xcd_logical_id(long bid, long nwg) {
  const long qx = nwg / 8, rx = nwg % 8, xcd = bid % 8;
  return (xcd < rx ? xcd * (qx + 1) : rx * (qx + 1) + (xcd - rx) * qx) + bid / 8;
}

// q = lid / d, r = lid % d for a workgroup index (< 2^31) in ONE unsigned 32-bit division.  `long % int` and `long / int` are
// ~120 vector instructions EACH on this target (no hardware divide: a float-reciprocal sequence in 64-bit arithmetic, executed by
// every wave even for a uniform value); round 6: four of them opened every c1_gconv / grouped-conv workgroup, a third of the
// 2.6 us its time stamps showed in front of the first load.
__device__ __forceinline__ void td_split(long lid, int d, int& q, int& r) {
  const unsigned u = (unsigned)lid, dd = (unsigned)d, qq = u / dd;
  q = (int)qq;
  r = (int)(u - qq * dd);
}

// sum over each row of 16 lanes (lanes sharing lane >> 4), every lane gets it: four DPP moves (lane ^ 1, lane ^ 2, the other quad
// of the half row, the other half row) -- the pairing of the xor butterfly `v += __shfl_xor(v, 1 / 2 / 4 / 8)`, i.e. the same
// bits, without its four ds_bpermute round trips through the LDS crossbar (hipcc emits one per step: ~40 in c1_gconv's tail)
__device__ __forceinline__ float td_row16_sum(float v) {
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true));
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xF, 0xF, true));
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x140, 0xF, 0xF, true));
  return v;
}

static inline int cdiv(long a, long b) { return (int)((a + b - 1) / b); }

// Division by a run-time constant without the ~25-instruction integer-divide sequence (index arithmetic of the
// staging loops is otherwise a large share of the instructions of the small latency-bound kernels): float
// reciprocal estimate + one-step fix-up, exact for 0 <= i < 2^22.
struct IDiv {
  int d;
  float r;
  // v_rcp_f32 (1 ulp) instead of the ~15-instruction IEEE division: i * r is then within 2^22 / d * 1.8e-7 < 1 of the true
  // quotient for i < 2^22, and div() below corrects an estimate that is off by one in either direction
  __device__ __forceinline__ explicit IDiv(int d_) : d(d_), r(__builtin_amdgcn_rcpf((float)d_)) {}
  __device__ __forceinline__ int div(int i) const {
    int q = (int)((float)i * r);
    const int rem = i - q * d;
    q += (rem >= d) ? 1 : 0;
    q -= (rem < 0) ? 1 : 0;
    return q;
  }
  __device__ __forceinline__ void divmod(int i, int& q, int& m) const {
    q = div(i);
    m = i - q * d;
  }
};

// MFMA operand (one column, 8 consecutive rows) out of a row-major [row][column] bf16 LDS image with two transposing
// reads (gfx950 ds_read_b64_tr_b16): per 16-lane group, lane 4q+p supplies the address of row q, columns 4p..4p+3 of a
// 4-row x 16-column block and lane i receives column i of the 4 rows.  lo / hi: this lane's address in rows 0..3 / 4..7.
// EXEC must be all ones (call from wave-uniform code only).
typedef __bf16 td_bf16x4 __attribute__((__vector_size__(4 * sizeof(__bf16))));
__device__ __forceinline__ bf16x8 td_tr_read8(const bf16_t* lo, const bf16_t* hi) {
  typedef __attribute__((address_space(3))) td_bf16x4 lds_v4;
  const td_bf16x4 a = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_v4*)(lo));
  const td_bf16x4 b = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_v4*)(hi));
  return (bf16x8){a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
}

// compiler-level fence: keeps the loads issued above it ahead of the LDS writes below it (the scheduler otherwise
// interleaves them and every write waits on its own load)
#define TD_ISSUE_FENCE() asm volatile("" ::: "memory")
