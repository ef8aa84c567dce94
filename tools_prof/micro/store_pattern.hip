// What a 256 x 256 output tile costs to write (bf16) or to read-add-write (fp32) from eight waves, by which lanes of a
// store instruction share a row: pattern A = one row per lane (lane (g, c) = row c, 16 columns 16 g ..), pattern B =
// sixteen lanes per row (lane (g, c) = row 4 g + j, 4 columns 4 c ..). Same bytes, same tile, one workgroup per CU
// walking its tiles.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

template <int PAT, bool F32>
__global__ __launch_bounds__(512) void tile_out(void* out, int M, int N, int n_tiles) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, wr = wave >> 2, wc = wave & 3, g = lane >> 4, c = lane & 15;
  const int tiles_n = N / 256;
  for (int t = blockIdx.x; t < n_tiles; t += gridDim.x) {
    const int m0 = (t / tiles_n) * 256, n0 = (t % tiles_n) * 256;
    if (PAT == 0) {
      const int nb = n0 + 64 * wc + 16 * g;
      if (F32) {
#pragma unroll
        for (int half = 0; half < 2; ++half) {
          f32x4 r[4][4];
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const float* p = (const float*)out + (size_t)(m0 + 128 * wr + 16 * (4 * half + q) + c) * N + nb;
#pragma unroll
            for (int j4 = 0; j4 < 4; ++j4) r[q][j4] = *(const f32x4*)(p + 4 * j4);
          }
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            float* p = (float*)out + (size_t)(m0 + 128 * wr + 16 * (4 * half + q) + c) * N + nb;
#pragma unroll
            for (int j4 = 0; j4 < 4; ++j4) { f32x4 x = r[q][j4]; x += 1.0f; *(f32x4*)(p + 4 * j4) = x; }
          }
        }
      } else {
#pragma unroll
        for (int tm = 0; tm < 8; ++tm) {
          unsigned short* p = (unsigned short*)out + (size_t)(m0 + 128 * wr + 16 * tm + c) * N + nb;
          u32x4 v = {(unsigned)t, (unsigned)tm, (unsigned)lane, 7u};
          *(u32x4*)p = v;
          *(u32x4*)(p + 8) = v;
        }
      }
    } else {
      const int nb = n0 + 64 * wc + 4 * c;
      if (F32) {
#pragma unroll
        for (int half = 0; half < 2; ++half) {
          f32x4 r[4][4];
#pragma unroll
          for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int j = 0; j < 4; ++j)
              r[q][j] = *(const f32x4*)((const float*)out + (size_t)(m0 + 128 * wr + 16 * (4 * half + q) + 4 * g + j) * N + nb);
#pragma unroll
          for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              f32x4 x = r[q][j];
              x += 1.0f;
              *(f32x4*)((float*)out + (size_t)(m0 + 128 * wr + 16 * (4 * half + q) + 4 * g + j) * N + nb) = x;
            }
        }
      } else {
#pragma unroll
        for (int tm = 0; tm < 8; ++tm)
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            unsigned short* p = (unsigned short*)out + (size_t)(m0 + 128 * wr + 16 * tm + 4 * g + j) * N + nb;
            u32x2 v = {(unsigned)t, (unsigned)(tm + lane)};
            *(u32x2*)p = v;
          }
      }
    }
  }
}

template <int PAT, bool F32>
void run(void* buf, int M, int N, int grid, const char* name) {
  const int n_tiles = (M / 256) * (N / 256);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((tile_out<PAT, F32>), dim3(grid), dim3(512), 0, 0, buf, M, N, n_tiles);
  hipEventRecord(e0, 0);
  const int it = 20;
  for (int i = 0; i < it; ++i) hipLaunchKernelGGL((tile_out<PAT, F32>), dim3(grid), dim3(512), 0, 0, buf, M, N, n_tiles);
  hipEventRecord(e1, 0);
  hipEventSynchronize(e1);
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  ms /= it;
  const double bytes = (double)M * N * (F32 ? 8.0 : 2.0);
  printf("%-28s M=%d N=%d grid=%d: %8.1f us  %6.2f TB/s  %6.2f us per tile and CU\n", name, M, N, grid, ms * 1e3, bytes / ms / 1e9,
         ms * 1e3 / ((double)n_tiles / grid));
}

int main() {
  const int M = 65536;
  void* buf;
  hipMalloc(&buf, (size_t)M * 4096 * 4);
  hipMemset(buf, 0, (size_t)M * 4096 * 4);
  for (int N : {1024, 4096}) {
    for (int grid : {256, 64}) {
      run<0, false>(buf, M, N, grid, "bf16 row per lane (A)");
      run<1, false>(buf, M, N, grid, "bf16 16 lanes per row (B)");
      run<0, true>(buf, M, N, grid, "fp32 rmw row per lane (A)");
      run<1, true>(buf, M, N, grid, "fp32 rmw 16 lanes per row (B)");
    }
  }
  return 0;
}
