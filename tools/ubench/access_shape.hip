// Micro-benchmark (measurement tool, not product code): HBM rate of the access shapes an MFMA epilogue produces against the
// fully coalesced shape, on a [M][320] bf16 tensor (640-byte rows) -- the s3 activation of RegNetY-800MF at B = 16.
//   mode 0: every wave instruction covers 1 KiB of contiguous bytes (16 B per lane)
//   mode 1: 16 rows x 64 B per wave instruction (lane = row + 16 * piece, 16 B per lane): the 16x16 accumulator with two
//           channel tiles per lane, ten waves of a workgroup cover the 640 B of the rows
//   mode 2: 16 rows x 32 B per wave instruction (8 B per lane): one channel tile per lane
//   mode 3: 4 rows x 256 B (lane = 16 pieces x 4 rows): what an LDS-transposed epilogue with 4-row groups would issue
//   mode 4: 32 rows x 32 B per wave instruction, 16 B per lane (a grouped-conv tile pair after v_permlane16_swap)
//   mode 5: 8 rows x 64 B per wave instruction, 8 B per lane (lane width vs segment size)
// op 0: read only (sum to a sink), 1: write only, 2: copy (read src shape = write shape)
// build: hipcc --offload-arch=gfx950 -O3 -o access_shape access_shape.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

constexpr int ROWB = 640;

template <int MODE, int OP>
__global__ __launch_bounds__(640) void k(const unsigned char* __restrict__ src, unsigned char* __restrict__ dst, long M,
                                         unsigned* sink) {
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const long ntiles = M / 64;
  unsigned acc = 0;
  for (long t = blockIdx.x; t < ntiles; t += gridDim.x) {
    const long base = t * 64 * ROWB;
    if constexpr (MODE == 0) {
      // 64 rows x 640 B = 40 KiB = 2560 pieces; 640 threads x 4
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const long off = base + ((long)j * 640 + tid) * 16;
        u32x4 v = {1u, 2u, 3u, 4u};
        if constexpr (OP != 1) v = *reinterpret_cast<const u32x4*>(src + off);
        if constexpr (OP != 0) *reinterpret_cast<u32x4*>(dst + off) = v;
        else acc += v[0] ^ v[3];
      }
    } else if constexpr (MODE == 1) {
      const int px = lane & 15, q = lane >> 4;
#pragma unroll
      for (int mt = 0; mt < 4; ++mt) {
        const long off = base + (long)(mt * 16 + px) * ROWB + wv * 64 + q * 16;
        u32x4 v = {1u, 2u, 3u, 4u};
        if constexpr (OP != 1) v = *reinterpret_cast<const u32x4*>(src + off);
        if constexpr (OP != 0) *reinterpret_cast<u32x4*>(dst + off) = v;
        else acc += v[0] ^ v[3];
      }
    } else if constexpr (MODE == 2) {
      const int px = lane & 15, q = lane >> 4;
#pragma unroll
      for (int mt = 0; mt < 4; ++mt)
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const long off = base + (long)(mt * 16 + px) * ROWB + wv * 64 + h * 32 + q * 8;
          u32x2 v = {1u, 2u};
          if constexpr (OP != 1) v = *reinterpret_cast<const u32x2*>(src + off);
          if constexpr (OP != 0) *reinterpret_cast<u32x2*>(dst + off) = v;
          else acc += v[0] ^ v[1];
        }
    } else if constexpr (MODE == 4) {
      const int row = lane & 31, pc = lane >> 5;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int unit = j * 10 + wv;                 // 40 units: (32-row half, 32-byte group of 20)
        const int half = unit / 20, g = unit % 20;
        const long off = base + (long)(half * 32 + row) * ROWB + g * 32 + pc * 16;
        u32x4 v = {1u, 2u, 3u, 4u};
        if constexpr (OP != 1) v = *reinterpret_cast<const u32x4*>(src + off);
        if constexpr (OP != 0) *reinterpret_cast<u32x4*>(dst + off) = v;
        else acc += v[0] ^ v[3];
      }
    } else if constexpr (MODE == 5) {
      const int row = lane >> 3, pc = lane & 7;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int unit = j * 10 + wv;                 // 80 units: (8-row group of 8, 64-byte group of 10)
        const int rg = unit / 10, g = unit % 10;
        const long off = base + (long)(rg * 8 + row) * ROWB + g * 64 + pc * 8;
        u32x2 v = {1u, 2u};
        if constexpr (OP != 1) v = *reinterpret_cast<const u32x2*>(src + off);
        if constexpr (OP != 0) *reinterpret_cast<u32x2*>(dst + off) = v;
        else acc += v[0] ^ v[1];
      }
    } else {
      // 4 rows x 256 B per wave instruction: wave wv of 10 takes pieces [16 wv', ...) -- 40 pieces per row = 2.5 waves per
      // row; use a flat split: instruction j of wave wv covers rows 4 * (..), 16 pieces
      const int pc = lane & 15, rr = lane >> 4;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int unit = (j * 10 + wv);               // 40 units of (4-row group g, 16-piece column block cb): 16 groups x 2.5
        const int g = unit % 16, cb = unit / 16;      // cb 0..2 (cb = 2 covers pieces 32..39 only with half the lanes)
        const int piece = cb * 16 + pc;
        if (piece < 40) {
          const long off = base + (long)(g * 4 + rr) * ROWB + piece * 16;
          u32x4 v = {1u, 2u, 3u, 4u};
          if constexpr (OP != 1) v = *reinterpret_cast<const u32x4*>(src + off);
          if constexpr (OP != 0) *reinterpret_cast<u32x4*>(dst + off) = v;
          else acc += v[0] ^ v[3];
        }
      }
    }
  }
  if (OP == 0 && acc == 0x12345678u) *sink = acc;
}

template <int MODE, int OP>
static float run(const unsigned char* s, unsigned char* d, long M, unsigned* sink, int grid) {
  hipEvent_t a, b;
  hipEventCreate(&a); hipEventCreate(&b);
  for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((k<MODE, OP>), dim3(grid), dim3(640), 0, 0, s, d, M, sink);
  hipEventRecord(a);
  const int R = 20;
  for (int i = 0; i < R; ++i) hipLaunchKernelGGL((k<MODE, OP>), dim3(grid), dim3(640), 0, 0, s, d, M, sink);
  hipEventRecord(b);
  hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  return ms / R;
}

int main() {
  const long M = 313600;
  const size_t bytes = (size_t)M * ROWB;
  unsigned char *s, *d; unsigned* sink;
  // several buffers so that consecutive launches do not find their rows in the Infinity Cache: rotate through 2 GB
  hipMalloc(&s, bytes); hipMalloc(&d, bytes); hipMalloc(&sink, 4);
  hipMemset(s, 1, bytes); hipMemset(d, 2, bytes);
  const char* opn[3] = {"read", "write", "copy"};
  for (int grid : {256, 1024}) {
    printf("grid %d (workgroups of 640 threads, 64-row tiles, %.0f MB tensor)\n", grid, bytes / 1e6);
#define RUN(MODE, OP)                                                                     \
  {                                                                                       \
    float ms = run<MODE, OP>(s, d, M, sink, grid);                                        \
    double b = (OP == 2 ? 2.0 : 1.0) * bytes;                                             \
    printf("  mode %d %-5s %8.1f us  %6.2f TB/s\n", MODE, opn[OP], ms * 1e3, b / ms / 1e9); \
  }
    RUN(0, 0) RUN(1, 0) RUN(2, 0) RUN(3, 0) RUN(4, 0) RUN(5, 0)
    RUN(0, 1) RUN(1, 1) RUN(2, 1) RUN(3, 1) RUN(4, 1) RUN(5, 1)
    RUN(0, 2) RUN(1, 2) RUN(2, 2) RUN(3, 2) RUN(4, 2) RUN(5, 2)
  }
  return 0;
}
