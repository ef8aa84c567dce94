// Microbenchmark: one wave per SIMD, per step one v_mfma_f32_32x32x16_bf16 (M), one ds_read_b128 of the operand
// needed 8 steps later (R), one counted s_waitcnt for this step's operand (W) and NF independent v_add_f32 (F),
// issued in the order given by ORDER (a string over R, W, F, M; F may appear twice = split in halves).
#include <hip/hip_runtime.h>
#include <cstdio>
#pragma clang diagnostic ignored "-Wunused-value"
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;

template <int NF, int O0, int O1, int O2, int O3, int O4, int FT = 0>
__global__ __launch_bounds__(256) void k(float* out, long long* cyc, int iters) {
  __shared__ __attribute__((aligned(16))) unsigned char lds[16384];
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
  unsigned la = (unsigned)(size_t)lds + (threadIdx.x & 63) * 16;
  constexpr int ord[5] = {O0, O1, O2, O3, O4};
  int nfs = 0;
  for (int i = 0; i < 5; ++i) nfs += ord[i] == 'F';
  long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      int fdone = 0;
#pragma unroll
      for (int i = 0; i < 5; ++i) {
        if (ord[i] == 'R') asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(a[(u + 8) % 16]) : "v"(la), "i"(((u + 8) % 16) * 1024 % 16384));
        if (ord[i] == 'W') asm volatile("s_waitcnt lgkmcnt(8)" : "+v"(a[u]));
        if (ord[i] == 'M') asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(acc0) : "v"(a[u]), "v"(b));
        if (ord[i] == 'F') {
          constexpr int per = NF;
#pragma unroll
          for (int f = 0; f < per; ++f) {
            if (FT == 0) asm volatile("v_add_f32 %0, %1, %0" : "+v"(d[(fdone + f) % 8]) : "v"(c));
            if (FT == 1) asm volatile("v_fmaak_f32 %0, %1, %0, 0x3c20659a" : "+v"(d[(fdone + f) % 8]) : "v"(c));
            if (FT == 2) asm volatile("v_fmaak_f32 %0, %1, %0, 0x3c20659a" : "+v"(d[f % 2]) : "v"(c));   // two dependent chains
            if (FT == 3) asm volatile("v_fma_f32 %0, %1, %0, %1" : "+v"(d[f % 2]) : "v"(c));             // same, 8-byte VOP3, no literal
            if (FT == 4) asm volatile("v_add_f32 %0, %1, %0" : "+v"(d[f % 2]) : "v"(c));               // same, 4-byte VOP2
          }
          fdone += per;
        }
      }
    }
  }
  long long t1 = __builtin_amdgcn_s_memtime();
  float s = 0.f;
  for (int r = 0; r < 16; ++r) s += acc0[r];
  for (int f = 0; f < 8; ++f) s += d[f];
  out[1 + threadIdx.x] = s + nfs;
  if (threadIdx.x == 0) cyc[0] = t1 - t0;
}
template <int NF, int O0, int O1, int O2, int O3, int O4 = 0, int FT = 0>
void run(float* out, long long* cyc) {
  const int iters = 2000;
  hipLaunchKernelGGL((k<NF, O0, O1, O2, O3, O4, FT>), dim3(1), dim3(256), 0, 0, out, cyc, iters);
  hipLaunchKernelGGL((k<NF, O0, O1, O2, O3, O4, FT>), dim3(1), dim3(256), 0, 0, out, cyc, iters);
  hipDeviceSynchronize();
  long long h;
  hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
  printf("order %c%c%c%c%c filler type %d, per F=%d: %.1f cycles per step\n", O0, O1, O2, O3, O4 ? O4 : ' ', FT, NF, (double)h / (iters * 16.0));
}
#define ALL(NF) run<NF,'R','W','F','M'>(out,cyc); run<NF,'R','W','M','F'>(out,cyc); run<NF,'R','F','W','M'>(out,cyc); \
  run<NF,'W','F','R','M'>(out,cyc); run<NF,'W','R','M','F'>(out,cyc); run<NF,'W','R','F','M'>(out,cyc);
int main() {
  float* out; long long* cyc;
  hipMalloc(&out, 4096); hipMalloc(&cyc, 64); hipMemset(out, 0, 4096);
  run<4,'W','R','F','M',0,0>(out,cyc); run<4,'W','R','F','M',0,1>(out,cyc); run<4,'W','R','F','M',0,2>(out,cyc);
  run<4,'W','R','F','M',0,3>(out,cyc); run<4,'W','R','F','M',0,4>(out,cyc);
  run<5,'W','R','F','M',0,0>(out,cyc); run<5,'W','R','F','M',0,1>(out,cyc);
  run<3,'W','R','F','M',0,1>(out,cyc); run<2,'W','R','F','M',0,1>(out,cyc);
  return 0;
}
