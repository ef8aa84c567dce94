// Shared device/host helpers for the bidiff HIP library (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define BD_OK 0
#define BD_ERR_ARG (-1)      // bad shape / unsupported configuration / null pointer
#define BD_ERR_LAUNCH (-2)   // HIP runtime reported an error at launch
#define BD_ERR_UNSUPPORTED (-3)

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;

// Record an error message retrievable through bd_last_error(); returns `code`.
int bd_fail(int code, const char* fmt, ...);
int bd_check_launch(const char* what);

#define BD_REQUIRE(cond, ...)                                  \
  do {                                                         \
    if (!(cond)) return bd_fail(BD_ERR_ARG, __VA_ARGS__);      \
  } while (0)

// Raise a kernel's dynamic-LDS limit to `bytes`: one runtime call per kernel instantiation, device and size
// reached, not one per launch (the attribute is per device and only ever needs to grow).
#define BD_SET_MAX_LDS(kernel, bytes)                                                                   \
  do {                                                                                                  \
    static int bd_lds_set_[64];                                                                         \
    int bd_dev_ = 0;                                                                                    \
    (void)hipGetDevice(&bd_dev_);                                                                       \
    int* bd_slot_ = &bd_lds_set_[bd_dev_ & 63];                                                         \
    if (__atomic_load_n(bd_slot_, __ATOMIC_ACQUIRE) < (int)(bytes)) {                                   \
      (void)hipFuncSetAttribute((const void*)(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)(bytes)); \
      __atomic_store_n(bd_slot_, (int)(bytes), __ATOMIC_RELEASE);                                       \
    }                                                                                                   \
  } while (0)

__device__ __forceinline__ float bd_bf16_to_f32(unsigned short b) {
  return __uint_as_float(((unsigned int)b) << 16);
}
// Round-to-nearest-even f32 -> bf16 through the hardware conversion (keeps NaN a NaN).
__device__ __forceinline__ unsigned short bd_f32_to_bf16(float f) {
  __bf16 h = (__bf16)f;
  return __builtin_bit_cast(unsigned short, h);
}
__device__ __forceinline__ unsigned int bd_pack_bf16x2(float lo, float hi) {
  return (unsigned int)bd_f32_to_bf16(lo) | ((unsigned int)bd_f32_to_bf16(hi) << 16);
}

__device__ __forceinline__ float bd_wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float bd_wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}
__device__ __forceinline__ int bd_wave_sum_i(int v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
