// Microbenchmark: cycles per v_mfma_f32_32x32x16_bf16 with N independent v_add_f32 fillers per MFMA,
// one wave per SIMD, accumulator in AGPRs or VGPRs, one or two accumulation chains.
#include <hip/hip_runtime.h>
#pragma clang diagnostic ignored "-Wunused-value"
#include <cstdio>
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;

template <int NF, bool AG, int CH>
__global__ __launch_bounds__(256) void k(float* out, long long* cyc, int iters) {
  f32x16 acc0, acc1;
  for (int r = 0; r < 16; ++r) { acc0[r] = 0.f; acc1[r] = 0.f; }
  bf16x8 a, b;
  for (int e = 0; e < 8; ++e) { a[e] = (__bf16)(0.01f * (threadIdx.x + e)); b[e] = (__bf16)(0.02f * (threadIdx.x - e)); }
  float d[6] = {1.f, 2.f, 3.f, 4.f, 5.f, 6.f};
  float c = out[0];
  long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      if (AG) {
        if (CH == 2 && (u & 1)) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(acc1) : "v"(a), "v"(b));
        else asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(acc0) : "v"(a), "v"(b));
      } else {
        if (CH == 2 && (u & 1)) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc1) : "v"(a), "v"(b));
        else asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc0) : "v"(a), "v"(b));
      }
#pragma unroll
      for (int f = 0; f < NF; ++f) asm volatile("v_add_f32 %0, %1, %0" : "+v"(d[f]) : "v"(c));
    }
  }
  long long t1 = __builtin_amdgcn_s_memtime();
  float s = 0.f;
  for (int r = 0; r < 16; ++r) s += acc0[r] + acc1[r];
  for (int f = 0; f < 6; ++f) s += d[f];
  out[1 + threadIdx.x] = s;
  if (threadIdx.x == 0) cyc[0] = t1 - t0;
}

// FEAT bits: 1 = s_nop 0 before the MFMA, 2 = s_waitcnt lgkmcnt(8) in the gap, 4 = one ds_read_b128 per gap into a
// rotating set of A operands (read 8 gaps ahead), 8 = fillers placed before the MFMA instead of after
template <int NF, int FEAT>
__global__ __launch_bounds__(256) void k2(float* out, long long* cyc, int iters) {
  __shared__ __attribute__((aligned(16))) unsigned char lds[16384];
  for (int i = threadIdx.x; i < 4096; i += 256) ((float*)lds)[i] = 0.001f * i;
  __syncthreads();
  f32x16 acc0;
  for (int r = 0; r < 16; ++r) acc0[r] = 0.f;
  bf16x8 a[16], b;
  for (int j = 0; j < 16; ++j)
    for (int e = 0; e < 8; ++e) a[j][e] = (__bf16)(0.01f * (threadIdx.x + e + j));
  for (int e = 0; e < 8; ++e) b[e] = (__bf16)(0.02f * (threadIdx.x - e));
  float d[6] = {1.f, 2.f, 3.f, 4.f, 5.f, 6.f};
  float c = out[0];
  unsigned la = (unsigned)(size_t)lds + (threadIdx.x & 63) * 16;
  long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      if (FEAT & 4) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(a[(u + 8) % 16]) : "v"(la), "i"(((u + 8) % 16) * 1024 % 16384));
      if (FEAT & 2) asm volatile("s_waitcnt lgkmcnt(8)" : "+v"(a[u]));
      if (FEAT & 8) {
#pragma unroll
        for (int f = 0; f < NF; ++f) asm volatile("v_add_f32 %0, %1, %0" : "+v"(d[f]) : "v"(c));
      }
      if (FEAT & 1) asm volatile("s_nop 0");
      asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(acc0) : "v"(a[u]), "v"(b));
      if (!(FEAT & 8)) {
#pragma unroll
        for (int f = 0; f < NF; ++f) asm volatile("v_add_f32 %0, %1, %0" : "+v"(d[f]) : "v"(c));
      }
    }
  }
  long long t1 = __builtin_amdgcn_s_memtime();
  float s = 0.f;
  for (int r = 0; r < 16; ++r) s += acc0[r];
  for (int f = 0; f < 6; ++f) s += d[f];
  out[1 + threadIdx.x] = s;
  if (threadIdx.x == 0) cyc[0] = t1 - t0;
}
template <int NF, int FEAT>
void run2(float* out, long long* cyc) {
  const int iters = 2000;
  hipLaunchKernelGGL((k2<NF, FEAT>), dim3(1), dim3(256), 0, 0, out, cyc, iters);
  hipLaunchKernelGGL((k2<NF, FEAT>), dim3(1), dim3(256), 0, 0, out, cyc, iters);
  hipDeviceSynchronize();
  long long h;
  hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
  printf("feat=%d fillers=%d: %.1f cycles per MFMA\n", FEAT, NF, (double)h / (iters * 16.0));
}

template <int NF, bool AG, int CH>
void run(float* out, long long* cyc, const char* name) {
  const int iters = 2000;
  hipLaunchKernelGGL((k<NF, AG, CH>), dim3(1), dim3(256), 0, 0, out, cyc, iters);
  hipLaunchKernelGGL((k<NF, AG, CH>), dim3(1), dim3(256), 0, 0, out, cyc, iters);
  hipDeviceSynchronize();
  long long h;
  hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
  printf("%s fillers=%d: %.1f cycles per MFMA\n", name, NF, (double)h / (iters * 16.0));
}

int main() {
  float* out; long long* cyc;
  hipMalloc(&out, 4096); hipMalloc(&cyc, 64); hipMemset(out, 0, 4096);
  run<0, true, 1>(out, cyc, "agpr 1 chain"); run<2, true, 1>(out, cyc, "agpr 1 chain"); run<4, true, 1>(out, cyc, "agpr 1 chain"); run<6, true, 1>(out, cyc, "agpr 1 chain");
  run<0, false, 1>(out, cyc, "vgpr 1 chain"); run<2, false, 1>(out, cyc, "vgpr 1 chain"); run<4, false, 1>(out, cyc, "vgpr 1 chain"); run<6, false, 1>(out, cyc, "vgpr 1 chain");
  run<0, true, 2>(out, cyc, "agpr 2 chains"); run<4, true, 2>(out, cyc, "agpr 2 chains"); run<6, true, 2>(out, cyc, "agpr 2 chains");
  run<4, false, 2>(out, cyc, "vgpr 2 chains");
  run2<0, 0>(out, cyc); run2<4, 0>(out, cyc);
  run2<0, 1>(out, cyc); run2<4, 1>(out, cyc);
  run2<0, 2>(out, cyc); run2<4, 2>(out, cyc);
  run2<0, 6>(out, cyc); run2<4, 6>(out, cyc); run2<4, 14>(out, cyc);
  run2<0, 7>(out, cyc); run2<4, 7>(out, cyc); run2<4, 15>(out, cyc); run2<2, 7>(out, cyc);
  return 0;
}
