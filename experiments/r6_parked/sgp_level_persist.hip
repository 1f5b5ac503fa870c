// EXPERIMENT (round 6): one pyramid level of the SGP encoder-decoder -- front (LayerNorm + depthwise
// branches), fc1 (GroupNorm + GELU), fc2 (+ residual, row sums) -- as ONE persistent launch with two device-scope grid
// barriers, against its three launches.  The phase bodies are the PRODUCT kernels themselves: this file includes
// csrc/sgp_fused.hip and csrc/sgp_gemm.hip with `__global__` turned into an inlined device function and `blockIdx` /
// `gridDim` into per-workgroup variables, so every phase runs exactly the code (and produces exactly the bits) of the launch
// it replaces; a workgroup loops over the virtual workgroups of a phase.  Built on its own (tools/bench_sgp_persist.py), never
// part of libtdeed_hip.so.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include "common.h"

void tdeed_set_error(const char* fmt, ...) { (void)fmt; }        // (lives in misc.hip of the product library)

struct TdVDim { unsigned x, y, z; };
__shared__ TdVDim td_vblock, td_vgrid;

#pragma push_macro("__global__")
#pragma push_macro("__launch_bounds__")
#undef __global__
#undef __launch_bounds__
#define __global__ __device__ __attribute__((always_inline))
#define __launch_bounds__(...)
#define blockIdx td_vblock
#define gridDim td_vgrid
#pragma push_macro("hipLaunchKernelGGL")
#undef hipLaunchKernelGGL
#define hipLaunchKernelGGL(...) ((void)0)
#define hipFuncSetAttribute(...) hipSuccess       // (the product launchers in the included files are never called)
#define tdeed_sgp_front_set_debug exp_unused_set_debug
#include "sgp_fused.hip"
#include "sgp_gemm.hip"
#undef blockIdx
#undef gridDim
#undef hipFuncSetAttribute
#pragma pop_macro("hipLaunchKernelGGL")
#pragma pop_macro("__launch_bounds__")
#pragma pop_macro("__global__")

struct LevelP {
  // front
  const float* x; int B, T, C, ks, up; const float* ln_w; const float* ln_b; float eps; const float* dw; const float* db;
  float* y; float* chsum; const float* rowstat; int rs_parts;
  // fc1 / fc2
  SgpGemmP g1, g2;
  unsigned* counter;        // grid barrier counter (zeroed by the host before every launch)
  long long* dbg;
};

// all workgroups of the launch (all resident: the host sizes the grid by the occupancy query) meet; stores before it are
// visible to loads behind it on every CU (release / acquire at agent scope, MI355X_MICROARCH.md "Valid forms")
__device__ __forceinline__ void grid_barrier(unsigned* counter, unsigned target) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    while (__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) __builtin_amdgcn_s_sleep(2);
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  __syncthreads();
}

__device__ __forceinline__ void set_vblock(unsigned bx, unsigned by, unsigned gx, unsigned gy) {
  __syncthreads();
  if (threadIdx.x == 0) { td_vblock = TdVDim{bx, by, 0}; td_vgrid = TdVDim{gx, gy, 1}; }
  __syncthreads();
}

__global__ __launch_bounds__(256, 2) void sgp_level_persist_kernel(const LevelP p) {
  const unsigned nwg = ::gridDim.x, me = ::blockIdx.x;
  if (p.dbg && threadIdx.x == 0) p.dbg[me * 8 + 0] = wall_clock64();
  // ---- phase 1: front, virtual grid (B, C / 16)
  const unsigned gy = (p.C + 15) / 16, n1 = p.B * gy;
  for (unsigned v = me; v < n1; v += nwg) {
    set_vblock(v % p.B, v / p.B, p.B, gy);
    sgp_front_kernel<float>(p.x, p.T, p.C, p.ks, p.up, p.ln_w, p.ln_b, p.eps, p.dw, p.db, p.y, p.chsum, p.rowstat, p.rs_parts,
                            (bf16_t*)nullptr, (long long*)nullptr);
  }
  if (p.dbg && threadIdx.x == 0) p.dbg[me * 8 + 1] = wall_clock64();
  grid_barrier(p.counter, nwg);
  if (p.dbg && threadIdx.x == 0) p.dbg[me * 8 + 2] = wall_clock64();
  // ---- phase 2: fc1 = sgp_gemm MODE 0, form (2, 1), fp32 rows
  const unsigned n2 = p.g1.B * p.g1.NJ * p.g1.nct;
  for (unsigned v = me; v < n2; v += nwg) {
    set_vblock(v, 0, n2, 1);
    sgp_gemm_kernel<2, 1, 0, float, bf16_t>(p.g1);
  }
  if (p.dbg && threadIdx.x == 0) p.dbg[me * 8 + 3] = wall_clock64();
  grid_barrier(p.counter, 2 * nwg);
  if (p.dbg && threadIdx.x == 0) p.dbg[me * 8 + 4] = wall_clock64();
  // ---- phase 3: fc2 = sgp_gemm MODE 1, form (2, 1), fp32 out
  const unsigned n3 = p.g2.B * p.g2.NJ * p.g2.nct;
  for (unsigned v = me; v < n3; v += nwg) {
    set_vblock(v, 0, n3, 1);
    sgp_gemm_kernel<2, 1, 1, bf16_t, float>(p.g2);
  }
  if (p.dbg && threadIdx.x == 0) p.dbg[me * 8 + 5] = wall_clock64();
}

extern "C" int exp_level_smem(int T, int ks, int up, int K1sp) {
  size_t a = front_smem(T, ks, up, 1, 1, 1), b = sg_smem_bytes(32, 0, K1sp), c = sg_smem_bytes(32, 1, 48);
  size_t m = a > b ? a : b;
  return (int)(m > c ? m : c);
}

// x [B][T][C] fp32 -> y = front(x) (fp32), H = GELU(GN(y) W1^T + b1) bf16 [B*T][4C], out = y + H W2^T + b2 fp32, rowstat_part
extern "C" int exp_level_persist(const float* x, int B, int T, int C, int ks, int up, const float* ln_w, const float* ln_b,
                                 float eps, const float* dw, const float* db, float* y, float* chsum, const float* rowstat,
                                 int rs_parts, const float* gn_w, const float* gn_b, const void* W1p, const float* b1, void* H,
                                 const void* W2p, const float* b2, float* out, float* rowstat_part, unsigned* counter,
                                 int grid, long long* dbg, void* stream) {
  LevelP p = {};
  p.x = x; p.B = B; p.T = T; p.C = C; p.ks = ks; p.up = up; p.ln_w = ln_w; p.ln_b = ln_b; p.eps = eps; p.dw = dw; p.db = db;
  p.y = y; p.chsum = chsum; p.rowstat = rowstat; p.rs_parts = rs_parts; p.counter = counter; p.dbg = dbg;
  const int N1 = 4 * C;
  SgpGemmP& a = p.g1;
  a.A = y; a.lda = C; a.W = (const bf16x8*)W1p; a.KSP = tdeed_sgp_gemm_ksteps(C); a.bias = b1; a.out = H; a.ldo = N1;
  a.B = B; a.T = T; a.N = N1; a.K = C; a.NJ = tdeed_sgp_gemm_row_tiles(T, 2); a.nct = tdeed_sgp_gemm_col_tiles(N1, 1);
  a.ct_major = 1; a.chsum = chsum; a.chs_parts = 1; a.gn_w = gn_w; a.gn_b = gn_b; a.G = 16; a.eps = 1e-5f;
  SgpGemmP& c = p.g2;
  c.A = H; c.lda = N1; c.W = (const bf16x8*)W2p; c.KSP = tdeed_sgp_gemm_ksteps(N1); c.bias = b2; c.out = out; c.ldo = C;
  c.B = B; c.T = T; c.N = C; c.K = N1; c.NJ = tdeed_sgp_gemm_row_tiles(T, 2); c.nct = tdeed_sgp_gemm_col_tiles(C, 1);
  c.ct_major = 0; c.resid = y; c.ldr = C; c.rowstat_part = rowstat_part;
  size_t sm = front_smem(T, ks, up, 1, 1, 1);
  if (sg_smem_bytes(32, 0, a.KSP) > sm) sm = sg_smem_bytes(32, 0, a.KSP);
  if (sg_smem_bytes(32, 1, c.KSP) > sm) sm = sg_smem_bytes(32, 1, c.KSP);
  hipStream_t st = (hipStream_t)stream;
  if (hipMemsetAsync(counter, 0, sizeof(unsigned), st) != hipSuccess) return 1;
  hipLaunchKernelGGL(sgp_level_persist_kernel, dim3(grid), dim3(256), sm, st, p);
  return hipGetLastError() == hipSuccess ? 0 : 2;
}

extern "C" int exp_level_max_grid(int smem_bytes) {
  int nb = 0;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, sgp_level_persist_kernel, 256, smem_bytes) != hipSuccess) return 0;
  hipDeviceProp_t pr;
  if (hipGetDeviceProperties(&pr, 0) != hipSuccess) return 0;
  return nb * pr.multiProcessorCount;
}
