// Heads, loss, displacement post-processing, small utilities, HIP-graph capture, error plumbing.
#include "common.h"
#include <stdarg.h>
#include <stdio.h>
#include <string.h>

// =========================================================================== error plumbing
static thread_local char g_err[512] = "";
void tdeed_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}
extern "C" const char* tdeed_last_error(void) { return g_err; }
extern "C" int tdeed_abi_version(void) { return 1; }

extern "C" int tdeed_device_info(int dev, char* name64, int* n_cu, int* is_gfx950) {
  hipDeviceProp_t prop;
  hipError_t e = hipGetDeviceProperties(&prop, dev);
  if (e != hipSuccess) {
    tdeed_set_error("device_info: %s", hipGetErrorString(e));
    return TDEED_ERR_RUNTIME;
  }
  if (name64) { strncpy(name64, prop.name, 63); name64[63] = 0; }
  if (n_cu) *n_cu = prop.multiProcessorCount;
  if (is_gfx950) *is_gfx950 = strncmp(prop.gcnArchName, "gfx950", 6) == 0;
  return TDEED_OK;
}

// =========================================================================== heads
// one wave per row: x row cached in registers, n_out dot products, wave64 reductions.
template <typename T>
__global__ __launch_bounds__(256) void heads_kernel(const T* __restrict__ x, int rows, int C,
                                                    const float* __restrict__ w, const float* __restrict__ b,
                                                    int n_out, float* __restrict__ out) {
  constexpr int EPC = Chunk<T>::N;
  constexpr int MAXCH = 4;
  const int lane = threadIdx.x & 63;
  const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const int nch = C / EPC;
  float v[MAXCH][EPC];
#pragma unroll
  for (int i = 0; i < MAXCH; ++i) {
    const int ck = lane + 64 * i;
    if (ck < nch) {
      Chunk<T>::load(x + row * C + (long)ck * EPC, v[i]);
    } else {
#pragma unroll
      for (int e = 0; e < EPC; ++e) v[i][e] = 0.f;
    }
  }
  for (int o = 0; o < n_out; ++o) {
    float a = 0.f;
#pragma unroll
    for (int i = 0; i < MAXCH; ++i) {
      const int ck = lane + 64 * i;
      if (ck < nch) {
        const float* wr = w + (long)o * C + ck * EPC;
#pragma unroll
        for (int e = 0; e < EPC; ++e) a = fmaf(v[i][e], wr[e], a);
      }
    }
    a = wave_sum(a);
    if (lane == 0) out[row * n_out + o] = a + b[o];
  }
}

extern "C" int tdeed_heads_fwd(const void* x, int rows, int C, const float* w, const float* b, int n_out,
                               float* out, int dtype, void* stream) {
  TD_CHECK(x && w && b && out, "heads: null pointer");
  TD_CHECK(rows > 0 && C % 8 == 0 && n_out > 0, "heads: bad sizes");
  hipStream_t st = (hipStream_t)stream;
  dim3 grid(cdiv(rows, 4));
  if (dtype == TDEED_F32) {
    TD_CHECK(C <= 1024, "heads: C=%d too wide", C);
    hipLaunchKernelGGL(heads_kernel<float>, grid, dim3(256), 0, st, (const float*)x, rows, C, w, b, n_out, out);
  } else if (dtype == TDEED_BF16) {
    TD_CHECK(C <= 2048, "heads: C=%d too wide", C);
    hipLaunchKernelGGL(heads_kernel<bf16_t>, grid, dim3(256), 0, st, (const bf16_t*)x, rows, C, w, b, n_out, out);
  } else { tdeed_set_error("heads: bad dtype %d", dtype); return TDEED_ERR_ARG; }
  TD_LAUNCH_CHECK("heads");
  return TDEED_OK;
}

// =========================================================================== loss
// single block: rows are B*T <= a few thousand, K1 <= 64.
__global__ __launch_bounds__(256) void loss_kernel(const float* __restrict__ logits, int rows, int ld, int K1,
                                                   const int64_t* __restrict__ hard,
                                                   const float* __restrict__ soft,
                                                   const float* __restrict__ cls_w, int displ_col,
                                                   const float* __restrict__ labelD, float* __restrict__ out) {
  __shared__ float scratch[8];
  float num = 0.f, den = 0.f, se = 0.f, bad = 0.f;
  for (int r = threadIdx.x; r < rows; r += 256) {
    const float* lg = logits + (long)r * ld;
    float m = lg[0];
    for (int k = 1; k < K1; ++k) m = fmaxf(m, lg[k]);
    float s = 0.f;
    for (int k = 0; k < K1; ++k) s += expf(lg[k] - m);
    const float lse = m + logf(s);
    if (soft) {
      const float* p = soft + (long)r * K1;
      float a = 0.f;
      for (int k = 0; k < K1; ++k) a += cls_w[k] * p[k] * (lse - lg[k]);
      num += a;
    } else {
      int yk = (int)hard[r];
      if (yk < 0 || yk >= K1) { bad = 1.f; yk = 0; }               // never indexed; the loss becomes NaN (torch raises here)
      num += cls_w[yk] * (lse - lg[yk]);
      den += cls_w[yk];
    }
    if (displ_col >= 0 && labelD) {
      const float d = lg[displ_col] - labelD[r];
      se += d * d;
    }
  }
  num = block_sum<4>(num, scratch);
  den = block_sum<4>(den, scratch);
  se = block_sum<4>(se, scratch);
  bad = block_sum<4>(bad, scratch);
  if (threadIdx.x == 0) {
    const float ce = bad > 0.f ? __builtin_nanf("") : (soft ? num / (float)rows : num / den);
    const float mse = (displ_col >= 0 && labelD) ? se / (float)rows : 0.f;
    out[0] = ce + mse;
    out[1] = ce;
    out[2] = mse;
  }
}

extern "C" int tdeed_loss_fwd(const float* logits, int rows, int ld, int K1, const int64_t* hard, const float* soft,
                              const float* cls_w, int displ_col, const float* labelD, float* out, void* stream) {
  TD_CHECK(logits && cls_w && out && (hard || soft), "loss: null pointer");
  TD_CHECK(rows > 0 && K1 > 0 && K1 <= ld && displ_col < ld, "loss: bad sizes");
  hipLaunchKernelGGL(loss_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, logits, rows, ld, K1, hard, soft,
                     cls_w, displ_col, labelD, out);
  TD_LAUNCH_CHECK("loss");
  return TDEED_OK;
}

// =========================================================================== process_prediction
// block per clip.  Order-independent restatement of the reference's sequential scatter-max:
// scores[j][k] = max(0, max_{t : clamp(t - rne(d_t)) == j} softmax(logits_t)[k]).
__global__ __launch_bounds__(256) void process_prediction_kernel(const float* __restrict__ head, int T_len, int ld,
                                                                 int K1, int displ_col,
                                                                 float* __restrict__ scores,
                                                                 int64_t* __restrict__ cls) {
  extern __shared__ float sm[];          // prob [T][K1], tgt [T] (as int)
  float* prob = sm;
  int* tgt = reinterpret_cast<int*>(sm + T_len * K1);
  const int b = blockIdx.x;
  const float* hb = head + (long)b * T_len * ld;
  for (int t = threadIdx.x; t < T_len; t += 256) {
    const float* lg = hb + (long)t * ld;
    float m = lg[0];
    for (int k = 1; k < K1; ++k) m = fmaxf(m, lg[k]);
    float s = 0.f;
    for (int k = 0; k < K1; ++k) {
      float e = expf(lg[k] - m);
      prob[t * K1 + k] = e;
      s += e;
    }
    for (int k = 0; k < K1; ++k) prob[t * K1 + k] /= s;
    int d = displ_col >= 0 ? (int)rintf(lg[displ_col]) : 0;   // round-half-to-even like torch.round
    int j = t - d;
    tgt[t] = j < 0 ? 0 : (j > T_len - 1 ? T_len - 1 : j);
  }
  __syncthreads();
  for (int i = threadIdx.x; i < T_len * K1; i += 256) {
    const int j = i / K1, k = i - j * K1;
    float m = 0.f;
    for (int t = 0; t < T_len; ++t)
      if (tgt[t] == j) m = fmaxf(m, prob[t * K1 + k]);
    scores[((long)b * T_len + j) * K1 + k] = m;
  }
  __syncthreads();
  for (int j = threadIdx.x; j < T_len; j += 256) {
    const float* sc = scores + ((long)b * T_len + j) * K1;
    int best = 0;
    float bv = sc[0];
    for (int k = 1; k < K1; ++k)
      if (sc[k] > bv) { bv = sc[k]; best = k; }
    cls[(long)b * T_len + j] = best;
  }
}

extern "C" int tdeed_process_prediction(const float* head_out, int B, int T, int ld, int K1, int displ_col,
                                        float* scores, int64_t* cls, void* stream) {
  TD_CHECK(head_out && scores && cls, "process_prediction: null pointer");
  TD_CHECK(B > 0 && T > 0 && K1 > 0 && K1 <= ld && displ_col < ld, "process_prediction: bad sizes");
  size_t smem = ((size_t)T * K1 + T) * sizeof(float);
  TD_CHECK(smem <= 64 * 1024, "process_prediction: clip too long");
  hipLaunchKernelGGL(process_prediction_kernel, dim3(B), dim3(256), smem, (hipStream_t)stream, head_out, T, ld, K1,
                     displ_col, scores, cls);
  TD_LAUNCH_CHECK("process_prediction");
  return TDEED_OK;
}

// =========================================================================== utilities
__global__ void cast_bf16_kernel(const float* __restrict__ s, bf16_t* __restrict__ d, long n) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
    d[i] = (bf16_t)s[i];
}
extern "C" int tdeed_cast_f32_to_bf16(const float* src, void* dst, long n, void* stream) {
  TD_CHECK(src && dst && n >= 0, "cast: bad args");
  if (n == 0) return TDEED_OK;
  int grid = (int)((n + 255) / 256 < 4096 ? (n + 255) / 256 : 4096);
  hipLaunchKernelGGL(cast_bf16_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, src, (bf16_t*)dst, n);
  TD_LAUNCH_CHECK("cast");
  return TDEED_OK;
}

// The LARGE dense weights' kernel copies in one launch: entry e = an fp32 [R][C] matrix of the flat parameter buffer, its
// copy in the activation dtype (dst, may be null) and the transposed copy [C][R] (dstT, may be null).  A workgroup serves one
// 32 x 32 tile of one entry (binary search over the entries' first tiles).  Replaces one cast + one transpose launch per
// weight (86 launches of 5 us per step for RegNetY-800MF + SGP).
struct CtEnt { const float* src; long R; long C; void* dst; void* dstT; long first_tile; };
template <typename T>
__global__ __launch_bounds__(256) void multi_cast_transpose_kernel(const CtEnt* __restrict__ tab, int nt) {
  __shared__ float tile[32][33];
  const long b = blockIdx.x;
  int lo = 0, hi = nt - 1;
  while (lo < hi) {                                              // last entry with first_tile <= b
    const int mid = (lo + hi + 1) >> 1;
    if (tab[mid].first_tile <= b) lo = mid; else hi = mid - 1;
  }
  const CtEnt e = tab[lo];
  const long w = b - e.first_tile;
  const long tc = (e.C + 31) / 32;
  const long r0 = (w / tc) * 32, c0 = (w % tc) * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  T* d = reinterpret_cast<T*>(e.dst);
  T* dT = reinterpret_cast<T*>(e.dstT);
  for (int i = ty; i < 32; i += 8) {
    const bool ok = r0 + i < e.R && c0 + tx < e.C;
    const float v = ok ? e.src[(r0 + i) * e.C + c0 + tx] : 0.f;
    tile[i][tx] = v;
    if (ok && d) d[(r0 + i) * e.C + c0 + tx] = (T)v;
  }
  if (!dT) return;
  __syncthreads();
  for (int i = ty; i < 32; i += 8)
    if (c0 + i < e.C && r0 + tx < e.R) dT[(c0 + i) * e.R + r0 + tx] = (T)tile[tx][i];
}

// tab: device array of n entries {src, R, C, dst, dstT, first_tile} (six 8-byte fields), tiles = sum of ceil(R/32) * ceil(C/32)
extern "C" int tdeed_multi_cast_transpose(const void* tab, int n, long tiles, int dtype, void* stream) {
  TD_CHECK(tab && n > 0 && tiles > 0 && tiles < 0x7fffffffL, "multi_cast_transpose: bad arguments");
  TD_CHECK(dtype == TDEED_F32 || dtype == TDEED_BF16, "multi_cast_transpose: bad dtype %d", dtype);
  if (dtype == TDEED_F32)
    hipLaunchKernelGGL(multi_cast_transpose_kernel<float>, dim3((unsigned)tiles), dim3(256), 0, (hipStream_t)stream, (const CtEnt*)tab, n);
  else
    hipLaunchKernelGGL(multi_cast_transpose_kernel<bf16_t>, dim3((unsigned)tiles), dim3(256), 0, (hipStream_t)stream, (const CtEnt*)tab, n);
  TD_LAUNCH_CHECK("multi_cast_transpose");
  return TDEED_OK;
}

// Kernel-layout copies of the master parameters in one launch: out[i] = idx[i] ? src[idx[i] - 1] : 0 (fp32 or bf16 out).
// Every packed tensor of the training engine -- bf16 casts, transposes, MFMA fragment orders, zero-padded vectors -- is a
// fixed permutation (with holes) of the flat fp32 parameter buffer, recorded once as an index table (repack.py); a
// training step refreshes all of them with two of these launches instead of ~350 small cast / transpose / gather kernels.
template <typename T>
__global__ __launch_bounds__(256) void gather_cast_kernel(const float* __restrict__ src, const int* __restrict__ idx, long n8,
                                                          T* __restrict__ out) {
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n8; i += (long)gridDim.x * 256) {
    const int4 a = *reinterpret_cast<const int4*>(idx + i * 8), b = *reinterpret_cast<const int4*>(idx + i * 8 + 4);
    const int j[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
    float v[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = src[j[e] > 0 ? j[e] - 1 : 0];     // always a valid address; selected afterwards
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = j[e] > 0 ? v[e] : 0.f;
    if constexpr (sizeof(T) == 2) {
      Chunk<bf16_t>::store(reinterpret_cast<bf16_t*>(out) + i * 8, v);
    } else {
      float lo[4] = {v[0], v[1], v[2], v[3]}, hi[4] = {v[4], v[5], v[6], v[7]};
      Chunk<float>::store(reinterpret_cast<float*>(out) + i * 8, lo);
      Chunk<float>::store(reinterpret_cast<float*>(out) + i * 8 + 4, hi);
    }
  }
}
// n: a multiple of 8 (pad the table with zeros); idx entries are 1-based positions in src, 0 = literal zero
extern "C" int tdeed_gather_cast(const float* src, const int* idx, long n, void* out, int dtype, void* stream) {
  TD_CHECK(src && idx && out && n > 0 && n % 8 == 0, "gather_cast: bad arguments");
  TD_CHECK(dtype == TDEED_F32 || dtype == TDEED_BF16, "gather_cast: bad dtype %d", dtype);
  const long n8 = n / 8;
  const int grid = (int)((n8 + 255) / 256 < 8192 ? (n8 + 255) / 256 : 8192);
  if (dtype == TDEED_F32)
    hipLaunchKernelGGL(gather_cast_kernel<float>, dim3(grid), dim3(256), 0, (hipStream_t)stream, src, idx, n8, (float*)out);
  else
    hipLaunchKernelGGL(gather_cast_kernel<bf16_t>, dim3(grid), dim3(256), 0, (hipStream_t)stream, src, idx, n8, (bf16_t*)out);
  TD_LAUNCH_CHECK("gather_cast");
  return TDEED_OK;
}

// same bytes as tdeed_amd.synth.uint8_clip: word i = splitmix64(splitmix64(i ^ base) + i), little endian
__device__ __forceinline__ uint64_t splitmix64(uint64_t x) {
  x += 0x9E3779B97F4A7C15ull;
  uint64_t z = x;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}
__global__ void fill_u8_hash_kernel(uint8_t* __restrict__ dst, long n, uint64_t base) {
  const long nw = (n + 7) / 8;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < nw; i += (long)gridDim.x * blockDim.x) {
    uint64_t h = splitmix64(splitmix64((uint64_t)i ^ base) + (uint64_t)i);
    if (i * 8 + 8 <= n) {
      *reinterpret_cast<uint64_t*>(dst + i * 8) = h;
    } else {
      for (long j = i * 8; j < n; ++j) dst[j] = (uint8_t)(h >> (8 * (j - i * 8)));
    }
  }
}
extern "C" int tdeed_fill_u8_hash(uint8_t* dst, long n, uint64_t base, void* stream) {
  TD_CHECK(dst && n >= 0, "fill_u8_hash: bad args");
  TD_CHECK(((uintptr_t)dst & 7) == 0, "fill_u8_hash: dst must be 8-byte aligned");
  if (n == 0) return TDEED_OK;
  hipLaunchKernelGGL(fill_u8_hash_kernel, dim3(2048), dim3(256), 0, (hipStream_t)stream, dst, n, base);
  TD_LAUNCH_CHECK("fill_u8_hash");
  return TDEED_OK;
}

// =========================================================================== HIP graphs
extern "C" int tdeed_graph_begin(void* stream) {
  hipError_t e = hipStreamBeginCapture((hipStream_t)stream, hipStreamCaptureModeThreadLocal);
  if (e != hipSuccess) { tdeed_set_error("graph_begin: %s", hipGetErrorString(e)); return TDEED_ERR_RUNTIME; }
  return TDEED_OK;
}
extern "C" int tdeed_graph_end(void* stream, void** graph_exec) {
  hipGraph_t g = nullptr;
  hipError_t e = hipStreamEndCapture((hipStream_t)stream, &g);
  if (e != hipSuccess || !g) { tdeed_set_error("graph_end: %s", hipGetErrorString(e)); return TDEED_ERR_RUNTIME; }
  hipGraphExec_t ex = nullptr;
  e = hipGraphInstantiate(&ex, g, nullptr, nullptr, 0);
  (void)hipGraphDestroy(g);
  if (e != hipSuccess) { tdeed_set_error("graph_instantiate: %s", hipGetErrorString(e)); return TDEED_ERR_RUNTIME; }
  *graph_exec = (void*)ex;
  return TDEED_OK;
}
extern "C" int tdeed_graph_launch(void* graph_exec, void* stream) {
  hipError_t e = hipGraphLaunch((hipGraphExec_t)graph_exec, (hipStream_t)stream);
  if (e != hipSuccess) { tdeed_set_error("graph_launch: %s", hipGetErrorString(e)); return TDEED_ERR_RUNTIME; }
  return TDEED_OK;
}
extern "C" int tdeed_graph_destroy(void* graph_exec) {
  if (graph_exec) (void)hipGraphExecDestroy((hipGraphExec_t)graph_exec);
  return TDEED_OK;
}
