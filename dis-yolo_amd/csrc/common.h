// Shared helpers for the gfx950 kernel library (device + host side).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include "../../include/disyolo.h"

#ifndef DY_HOST_ONLY   // (`make asan`: host-side sources as plain C++, no device types)
typedef __bf16 bf16;
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

#endif
#define DY_WAVE 64

void disyolo_set_error(const char* fmt, ...);

#define DY_REQUIRE(cond, ...)                 \
  do {                                        \
    if (!(cond)) {                            \
      disyolo_set_error(__VA_ARGS__);         \
      return DISYOLO_E_ARG;                   \
    }                                         \
  } while (0)

#define DY_CHECK_LAUNCH()                                                    \
  do {                                                                       \
    hipError_t e_ = hipGetLastError();                                       \
    if (e_ != hipSuccess) {                                                  \
      disyolo_set_error("%s:%d launch failed: %s", __FILE__, __LINE__,       \
                        hipGetErrorString(e_));                              \
      return DISYOLO_E_HIP;                                                  \
    }                                                                        \
  } while (0)

#ifndef DY_HOST_ONLY
__device__ __forceinline__ float bf2f(bf16 v) { return (float)v; }
__device__ __forceinline__ float bfbits2f(unsigned short b) {
  return __builtin_bit_cast(float, (unsigned)b << 16);
}
// 8 bf16 packed in a uint4 -> 8 floats
__device__ __forceinline__ void unpack8(const uint4& u, float* f) {
  f[0] = __builtin_bit_cast(float, u.x << 16);
  f[1] = __builtin_bit_cast(float, u.x & 0xffff0000u);
  f[2] = __builtin_bit_cast(float, u.y << 16);
  f[3] = __builtin_bit_cast(float, u.y & 0xffff0000u);
  f[4] = __builtin_bit_cast(float, u.z << 16);
  f[5] = __builtin_bit_cast(float, u.z & 0xffff0000u);
  f[6] = __builtin_bit_cast(float, u.w << 16);
  f[7] = __builtin_bit_cast(float, u.w & 0xffff0000u);
}
__device__ __forceinline__ unsigned pack2(float lo, float hi) {
  typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
  bf16x2 v = {(bf16)lo, (bf16)hi};
  return __builtin_bit_cast(unsigned, v);
}
__device__ __forceinline__ uint4 pack8(const float* f) {
  uint4 u;
  u.x = pack2(f[0], f[1]);
  u.y = pack2(f[2], f[3]);
  u.z = pack2(f[4], f[5]);
  u.w = pack2(f[6], f[7]);
  return u;
}
__device__ __forceinline__ float leaky(float v, float alpha) { return fmaxf(alpha * v, v); }

// sum across the 64 lanes of a wave (result in every lane)
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

#endif  // DY_HOST_ONLY

static inline int ceil_div(int64_t a, int64_t b) { return (int)((a + b - 1) / b); }
