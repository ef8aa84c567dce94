// Microbenchmark: what one 1-KiB LDS-DMA piece costs the issuing wave beside its MFMAs, for the forms the ISA offers:
//   0 none   1 global_load_lds_dwordx4 (lane offset in a VGPR, scalar base)   2 buffer_load_dwordx4 ... lds, lane offset
//   in a VGPR
// One workgroup, one wave per SIMD; a step = wait, ds_read_b128, MFMA; one piece every fourth step. Measured on an
// otherwise idle chip: 33.3 cycles per step without pieces, 35.5 with form 1, 34.8 with form 2: a piece costs its wave
// 6-9 cycles when the memory system is idle; the 58 cycles seen inside the full-chip kernels are the memory system's
// back-pressure, not the instruction. (A resource with the lane term from ADD_TID_ENABLE instead of a VGPR faulted the
// GPU as encoded here and is not part of this file.)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((address_space(3))) void lds_void;
typedef const __attribute__((address_space(1))) void glb_void;

template <int FORM>
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
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  unsigned la = (unsigned)(size_t)lds + lane * 16;
  const unsigned ldsw = (unsigned)(size_t)(lds_void*)lds + 16384 + wave * 1024;
  unsigned voff = lane * 16 + wave * 1024;
  asm volatile("" : "+v"(voff));
  // buffer resource over the 128-KiB source
  __amdgpu_buffer_rsrc_t r_plain = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, 8 * 16384, 0x00027000);
  long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
    const unsigned char* unit = src + (size_t)(it & 7) * 16384;
    const unsigned soff = (unsigned)((it & 7) * 16384);
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      if (u % 4 == 0 && FORM != 0) {
        const int kpc = u / 4;
        if (FORM == 1) {
          asm volatile("s_add_u32 m0, %0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, %3"
                       ::"s"(ldsw), "i"(kpc * 4096), "v"(voff), "s"(unit) : "memory", "m0", "scc");
        } else {
          asm volatile("s_add_u32 m0, %0, %1\n\ts_nop 0\n\tbuffer_load_dwordx4 %2, %3, %4 offen offset:%5 lds"
                       ::"s"(ldsw), "i"(kpc * 4096), "v"(voff), "s"(r_plain), "s"(soff), "i"(0) : "memory", "m0", "scc");
        }
      }
      asm volatile("s_waitcnt lgkmcnt(7)" : "+v"(a[u]));
      asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(a[(u + 8) % 16]) : "v"(la), "i"(((u + 8) % 16) * 1024));
      asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(acc0) : "v"(a[u]), "v"(b));
    }
    asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  long long t1 = __builtin_amdgcn_s_memtime();
  float s = 0.f;
  for (int r = 0; r < 16; ++r) s += acc0[r];
  out[1 + threadIdx.x] = s + lds[16384 + threadIdx.x] + lds[16384 + 4096 + threadIdx.x];
  if (threadIdx.x == 0) cyc[0] = t1 - t0;
}
template <int FORM>
void run(float* out, long long* cyc, const unsigned char* src) {
  const int iters = 2000;
  hipFuncSetAttribute((const void*)k<FORM>, hipFuncAttributeMaxDynamicSharedMemorySize, 49152);
  hipLaunchKernelGGL((k<FORM>), dim3(1), dim3(256), 49152, 0, out, cyc, src, iters);
  hipLaunchKernelGGL((k<FORM>), dim3(1), dim3(256), 49152, 0, out, cyc, src, iters);
  hipDeviceSynchronize();
  long long h; float chk;
  hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
  hipMemcpy(&chk, out + 1, 4, hipMemcpyDeviceToHost);
  printf("form %d: %.1f cycles per step (check %g)\n", FORM, (double)h / (iters * 16.0), chk);
}
int main() {
  float* out; long long* cyc; unsigned char* src;
  hipMalloc(&out, 4096); hipMalloc(&cyc, 64); hipMemset(out, 0, 4096);
  hipMalloc(&src, 8 * 16384); hipMemset(src, 1, 8 * 16384);
  run<0>(out, cyc, src); run<1>(out, cyc, src); run<2>(out, cyc, src);
  return 0;
}
