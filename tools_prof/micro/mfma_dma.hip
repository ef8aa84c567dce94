// Microbenchmark: the W R F M step (see mfma_order.hip) plus one LDS-DMA piece (1 KiB per wave) every fourth step,
// placed at position POS of the step: 0 = before W, 1 = after W, 2 = after R, 3 = after F (before M), 4 = none.
#include <hip/hip_runtime.h>
#include <cstdio>
#pragma clang diagnostic ignored "-Wunused-value"
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((address_space(3))) void lds_void;
typedef const __attribute__((address_space(1))) void glb_void;

template <int NF, int POS, int NFD>
__global__ __launch_bounds__(256) void k(float* out, long long* cyc, const unsigned char* src, int iters) {
  extern __shared__ __attribute__((aligned(1024))) unsigned char lds[];
  for (int i = threadIdx.x; i < 4096; i += 256) ((float*)lds)[i] = 0.001f * i;
  __syncthreads();
  f32x16 acc0;
  for (int r = 0; r < 16; ++r) acc0[r] = 0.f;
  bf16x8 a[16], b;
  for (int j = 0; j < 16; ++j)
    for (int e = 0; e < 8; ++e) a[j][e] = (__bf16)(0.01f * (threadIdx.x + e + j));
  for (int e = 0; e < 8; ++e) b[e] = (__bf16)(0.02f * (threadIdx.x - e));
  float d[8] = {1.f, 2.f, 3.f, 4.f, 5.f, 6.f, 7.f, 8.f};
  float c = out[0];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  unsigned la = (unsigned)(size_t)lds + lane * 16;
  long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      auto dma = [&]() {
        if (u % 4 == 0) {
          const int f = wave + 4 * (u / 4);
          __builtin_amdgcn_global_load_lds((glb_void*)(src + (size_t)(it & 7) * 16384 + f * 1024 + lane * 16),
                                           (lds_void*)(lds + 16384 + f * 1024), 16, 0, 0);
        }
      };
      const bool dstep = (u % 4 == 0) && POS != 4;
      if (POS == 0) dma();
      asm volatile("s_waitcnt lgkmcnt(7)" : "+v"(a[u]));
      if (POS == 1) dma();
      asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(a[(u + 8) % 16]) : "v"(la), "i"(((u + 8) % 16) * 1024));
      if (POS == 2) dma();
#pragma unroll
      for (int f = 0; f < (dstep ? NFD : NF); ++f) asm volatile("v_fmaak_f32 %0, %1, %0, 0x3c20659a" : "+v"(d[f % 2]) : "v"(c));
      if (POS == 3) dma();
      asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(acc0) : "v"(a[u]), "v"(b));
    }
    asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  long long t1 = __builtin_amdgcn_s_memtime();
  float s = 0.f;
  for (int r = 0; r < 16; ++r) s += acc0[r];
  for (int f = 0; f < 8; ++f) s += d[f];
  out[1 + threadIdx.x] = s + lds[16384 + threadIdx.x];
  if (threadIdx.x == 0) cyc[0] = t1 - t0;
}
template <int NF, int POS, int NFD>
void run(float* out, long long* cyc, const unsigned char* src) {
  const int iters = 2000;
  hipFuncSetAttribute((const void*)k<NF, POS, NFD>, hipFuncAttributeMaxDynamicSharedMemorySize, 49152);
  hipLaunchKernelGGL((k<NF, POS, NFD>), dim3(1), dim3(256), 49152, 0, out, cyc, src, iters);
  hipLaunchKernelGGL((k<NF, POS, NFD>), dim3(1), dim3(256), 49152, 0, out, cyc, src, iters);
  hipDeviceSynchronize();
  long long h;
  hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
  printf("fillers %d (%d in DMA steps), DMA position %d: %.1f cycles per step\n", NF, NFD, POS, (double)h / (iters * 16.0));
}
int main() {
  float* out; long long* cyc; unsigned char* src;
  hipMalloc(&out, 4096); hipMalloc(&cyc, 64); hipMemset(out, 0, 4096);
  hipMalloc(&src, 8 * 16384); hipMemset(src, 1, 8 * 16384);
  run<0, 4, 0>(out, cyc, src); run<0, 0, 0>(out, cyc, src); run<0, 1, 0>(out, cyc, src); run<0, 2, 0>(out, cyc, src); run<0, 3, 0>(out, cyc, src);
  run<4, 4, 4>(out, cyc, src); run<4, 0, 4>(out, cyc, src); run<4, 1, 4>(out, cyc, src); run<4, 2, 4>(out, cyc, src); run<4, 3, 4>(out, cyc, src);
  run<4, 0, 0>(out, cyc, src); run<4, 1, 0>(out, cyc, src); run<4, 2, 0>(out, cyc, src); run<4, 3, 0>(out, cyc, src);
  return 0;
}
