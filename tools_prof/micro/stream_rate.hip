// Microbenchmark: how fast can one workgroup (4 waves, one per SIMD) pull an L2-resident buffer into the CU,
// (a) by LDS-DMA (global_load_lds_dwordx4, 1 KiB per wave instruction) and (b) by global_load_dwordx4 into registers,
// with D instructions in flight per wave. Reports bytes per clock per CU for one workgroup and for a full chip.
#include <hip/hip_runtime.h>
#include <cstdio>
#pragma clang diagnostic ignored "-Wunused-value"
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;

template <int MODE, int D, int NW = 4>
__global__ __launch_bounds__(64 * NW) void k(const unsigned char* src, size_t bytes, int iters, long long* cyc, unsigned* sink) {
  extern __shared__ __attribute__((aligned(1024))) unsigned char lds[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const unsigned voff = lane * 16;
  const size_t n_pieces = bytes / 1024;          // pieces of 1 KiB, wave w takes pieces w, w + 4, ...
  u32x4 ring[D];
  for (int i = 0; i < D; ++i) ring[i] = (u32x4){0, 0, 0, 0};
  unsigned acc = 0;
  const unsigned ldsb = (unsigned)(size_t)(__attribute__((address_space(3))) void*)lds + wave * (65536 / NW);
  long long t0 = __builtin_amdgcn_s_memtime();
  size_t p = wave + (size_t)blockIdx.x * NW % n_pieces;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < D; ++i) {
      const unsigned char* a = src + (p % n_pieces) * 1024;
      p += NW;
      if (MODE == 0) {
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(D - 1) : "memory");
        asm volatile("s_add_u32 m0, %0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, %3" ::"s"(ldsb), "i"(i * 1024), "v"(voff), "s"(a) : "memory", "m0", "scc");
      } else {
        asm volatile("s_waitcnt vmcnt(%1)" : "+v"(ring[i]) : "n"(D - 1));
        acc += ring[i][0];
        asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(ring[i]) : "v"(voff), "s"(a) : "memory");
      }
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  long long t1 = __builtin_amdgcn_s_memtime();
  if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
  if (acc == 0x12345678u) sink[threadIdx.x] = acc + lds[threadIdx.x];
}

template <int MODE, int D>
void run(const unsigned char* src, size_t bytes, long long* cyc, unsigned* sink) {
  const int iters = 4096 / D;
  for (int grid : {1, 256}) {
    hipFuncSetAttribute((const void*)k<MODE, D>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    hipLaunchKernelGGL((k<MODE, D>), dim3(grid), dim3(256), 65536, 0, src, bytes, iters, cyc, sink);
    hipLaunchKernelGGL((k<MODE, D>), dim3(grid), dim3(256), 65536, 0, src, bytes, iters, cyc, sink);
    hipDeviceSynchronize();
    long long h;
    hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
    printf("%s, %2d in flight per wave, %3d workgroups: %.1f B/clk per CU\n", MODE == 0 ? "LDS-DMA  " : "registers", D, grid,
           4.0 * iters * D * 1024 / (double)h);
  }
}
// the same with 8 and 16 waves per workgroup (two and four per SIMD), LDS-DMA, 4 pieces in flight per wave
template <int NW>
void run_waves(const unsigned char* src, size_t bytes, long long* cyc, unsigned* sink) {
  const int iters = 1024;
  for (int grid : {1, 256}) {
    hipFuncSetAttribute((const void*)k<0, 4, NW>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    hipLaunchKernelGGL((k<0, 4, NW>), dim3(grid), dim3(64 * NW), 65536, 0, src, bytes, iters, cyc, sink);
    hipLaunchKernelGGL((k<0, 4, NW>), dim3(grid), dim3(64 * NW), 65536, 0, src, bytes, iters, cyc, sink);
    hipDeviceSynchronize();
    long long h;
    hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
    printf("LDS-DMA  , %2d waves per workgroup, 4 in flight per wave, %3d workgroups: %.1f B/clk per CU\n", NW, grid,
           (double)NW * iters * 4 * 1024 / (double)h);
  }
}
int main() {
  unsigned char* src; long long* cyc; unsigned* sink;
  const size_t bytes = 2u << 20;
  hipMalloc(&src, bytes); hipMemset(src, 1, bytes); hipMalloc(&cyc, 64); hipMalloc(&sink, 4096);
  run<0, 1>(src, bytes, cyc, sink); run<0, 2>(src, bytes, cyc, sink); run<0, 4>(src, bytes, cyc, sink); run<0, 8>(src, bytes, cyc, sink); run<0, 16>(src, bytes, cyc, sink);
  run<1, 1>(src, bytes, cyc, sink); run<1, 2>(src, bytes, cyc, sink); run<1, 4>(src, bytes, cyc, sink); run<1, 8>(src, bytes, cyc, sink); run<1, 16>(src, bytes, cyc, sink);
  run_waves<8>(src, bytes, cyc, sink); run_waves<16>(src, bytes, cyc, sink);
  return 0;
}
