// Microbenchmark: W R F M steps (see mfma_order.hip) made step by step more like the fused feed-forward loop:
// bit 1 = s_nop 0 before the MFMA, bit 2 = a different B operand per step (24 of them), bit 4 = 24-step unroll with a ring of 32
// A operands, bit 8 = accumulators alternate between 12 tiles in the second half of the unrolled body.
#include <hip/hip_runtime.h>
#include <cstdio>
#pragma clang diagnostic ignored "-Wunused-value"
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;

template <int NF, int FEAT>
__global__ __launch_bounds__(256) void k(float* out, long long* cyc, int iters) {
  __shared__ __attribute__((aligned(16))) unsigned char lds[32768];
  for (int i = threadIdx.x; i < 8192; i += 256) ((float*)lds)[i] = 0.001f * i;
  __syncthreads();
  constexpr int NS = (FEAT & 16) ? 192 : (FEAT & 4) ? 48 : 16;       // steps per unrolled body
  constexpr int RING = (FEAT & 4) ? 32 : 16;
  f32x16 acc[13];
  for (int t = 0; t < 13; ++t)
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
  bf16x8 a[RING], b[24];
  for (int j = 0; j < RING; ++j)
    for (int e = 0; e < 8; ++e) a[j][e] = (__bf16)(0.01f * (threadIdx.x + e + j));
  for (int j = 0; j < 24; ++j)
    for (int e = 0; e < 8; ++e) b[j][e] = (__bf16)(0.02f * (threadIdx.x - e + j));
  float d[8] = {1.f, 2.f, 3.f, 4.f, 5.f, 6.f, 7.f, 8.f};
  float c = out[0];
  unsigned la = (unsigned)(size_t)lds + (threadIdx.x & 63) * 16;
  if (FEAT & 32) for (int k = 0; k < (int)(threadIdx.x >> 6) * 7; ++k) __builtin_amdgcn_s_sleep(3);
  long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < NS; ++u) {
      asm volatile("s_waitcnt lgkmcnt(7)" : "+v"(a[u % RING]));
      asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(a[(u + 8) % RING]) : "v"(la), "i"(((u + 8) % RING) * 1024));
#pragma unroll
      for (int f = 0; f < NF; ++f) {
        if (FEAT & 64) asm volatile("v_fmaak_f32 %0, %1, %0, 0x3c20659a" : "+v"(d[f % 2]) : "v"(c));
        else asm volatile("v_add_f32 %0, %1, %0" : "+v"(d[f]) : "v"(c));
      }
      if (FEAT & 1) asm volatile("s_nop 0");
      const int bi = (FEAT & 2) ? u % 24 : 0;
      const int ti = ((FEAT & 8) && u >= 24) ? 1 + (u - 24) / 2 : 0;
      asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(acc[ti]) : "v"(a[u % RING]), "v"(b[bi]));
    }
  }
  long long t1 = __builtin_amdgcn_s_memtime();
  float s = 0.f;
  for (int t = 0; t < 13; ++t)
    for (int r = 0; r < 16; ++r) s += acc[t][r];
  for (int f = 0; f < 8; ++f) s += d[f];
  out[1 + threadIdx.x] = s;
  if (threadIdx.x == 0) cyc[0] = (t1 - t0) / NS;
}
template <int NF, int FEAT>
void run(float* out, long long* cyc) {
  const int iters = (FEAT & 16) ? 250 : 1000;
  hipLaunchKernelGGL((k<NF, FEAT>), dim3(1), dim3(256), 0, 0, out, cyc, iters);
  hipLaunchKernelGGL((k<NF, FEAT>), dim3(1), dim3(256), 0, 0, out, cyc, iters);
  hipDeviceSynchronize();
  long long h;
  hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
  printf("feat %2d fillers %d: %.1f cycles per step\n", FEAT, NF, (double)h / iters);
}
int main() {
  float* out; long long* cyc;
  hipMalloc(&out, 4096); hipMalloc(&cyc, 64); hipMemset(out, 0, 4096);
  run<0, 79>(out, cyc); run<1, 79>(out, cyc); run<2, 79>(out, cyc); run<3, 79>(out, cyc); run<4, 79>(out, cyc); run<5, 79>(out, cyc); run<6, 79>(out, cyc); run<7, 79>(out, cyc); run<8, 79>(out, cyc);
  run<3, 15>(out, cyc); run<4, 15>(out, cyc); run<5, 15>(out, cyc); run<6, 15>(out, cyc); run<7, 15>(out, cyc); run<8, 15>(out, cyc);
  return 0;
}
