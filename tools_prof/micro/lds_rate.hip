// Microbenchmark: LDS read bandwidth of one workgroup of 4 waves (one per SIMD): ds_read_b128 streams with counted
// waits, optionally with LDS-DMA pieces landing in another region, optionally with an MFMA per read.
#include <hip/hip_runtime.h>
#include <cstdio>
#pragma clang diagnostic ignored "-Wunused-value"
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((address_space(3))) void lds_void;
typedef const __attribute__((address_space(1))) void glb_void;

template <int MFMA, int DMA, int SAME>
__global__ __launch_bounds__(256) void k(float* out, long long* cyc, const unsigned char* src, int iters) {
  extern __shared__ __attribute__((aligned(1024))) unsigned char lds[];
  for (int i = threadIdx.x; i < 16384; i += 256) ((float*)lds)[i] = 0.001f * i;
  __syncthreads();
  f32x16 acc0;
  for (int r = 0; r < 16; ++r) acc0[r] = 0.f;
  bf16x8 a[16], b;
  for (int j = 0; j < 16; ++j)
    for (int e = 0; e < 8; ++e) a[j][e] = (__bf16)(0.01f * (threadIdx.x + e + j));
  for (int e = 0; e < 8; ++e) b[e] = (__bf16)(0.02f * (threadIdx.x - e));
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  // SAME: all waves read the same fragments; else each wave its own 16 KiB
  unsigned la = (unsigned)(size_t)lds + lane * 16 + (SAME ? 0 : wave * 16384);
  long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      if (DMA && u % 4 == 0) {
        const int f = wave + 4 * (u / 4);
        __builtin_amdgcn_global_load_lds((glb_void*)(src + (size_t)(it & 7) * 16384 + f * 1024 + lane * 16),
                                         (lds_void*)(lds + 65536 + f * 1024), 16, 0, 0);
      }
      asm volatile("s_waitcnt lgkmcnt(7)" : "+v"(a[u]));
      asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(a[(u + 8) % 16]) : "v"(la), "i"(((u + 8) % 16) * 1024));
      if (MFMA) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(acc0) : "v"(a[u]), "v"(b));
      else asm volatile("" :: "v"(a[u]));
    }
    if (DMA) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  long long t1 = __builtin_amdgcn_s_memtime();
  float s = 0.f;
  for (int r = 0; r < 16; ++r) s += acc0[r];
  for (int j = 0; j < 16; ++j) s += (float)a[j][0];
  out[1 + threadIdx.x] = s + lds[65536 + threadIdx.x];
  if (threadIdx.x == 0) cyc[0] = t1 - t0;
}
template <int MFMA, int DMA, int SAME>
void run(float* out, long long* cyc, const unsigned char* src) {
  const int iters = 2000;
  hipFuncSetAttribute((const void*)k<MFMA, DMA, SAME>, hipFuncAttributeMaxDynamicSharedMemorySize, 98304);
  hipLaunchKernelGGL((k<MFMA, DMA, SAME>), dim3(1), dim3(256), 98304, 0, out, cyc, src, iters);
  hipLaunchKernelGGL((k<MFMA, DMA, SAME>), dim3(1), dim3(256), 98304, 0, out, cyc, src, iters);
  hipDeviceSynchronize();
  long long h;
  hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
  const double c = (double)h / (iters * 16.0);
  printf("mfma %d dma %d same-address %d: %.1f cycles per read per wave = %.0f B/clk per CU\n", MFMA, DMA, SAME, c, 4096.0 / c);
}
int main() {
  float* out; long long* cyc; unsigned char* src;
  hipMalloc(&out, 4096); hipMalloc(&cyc, 64); hipMemset(out, 0, 4096);
  hipMalloc(&src, 8 * 16384); hipMemset(src, 1, 8 * 16384);
  run<0, 0, 1>(out, cyc, src); run<0, 0, 0>(out, cyc, src); run<0, 1, 1>(out, cyc, src); run<0, 1, 0>(out, cyc, src);
  run<1, 0, 1>(out, cyc, src); run<1, 0, 0>(out, cyc, src); run<1, 1, 1>(out, cyc, src); run<1, 1, 0>(out, cyc, src);
  return 0;
}
