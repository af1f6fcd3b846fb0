// Shared helpers for libsacv_hip.so (gfx950 / MI355X only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#define SACV_OK 0
#define SACV_ERR_ARG -1
#define SACV_ERR_LAUNCH -2
#define SACV_ERR_UNSUPPORTED -3

void sacv_set_error(const char* fmt, ...);

#define SACV_CHECK_ARG(cond, ...)          \
  do {                                     \
    if (!(cond)) {                         \
      sacv_set_error(__VA_ARGS__);         \
      return SACV_ERR_ARG;                 \
    }                                      \
  } while (0)

#define SACV_CHECK_LAUNCH(name)                                            \
  do {                                                                     \
    hipError_t e__ = hipGetLastError();                                    \
    if (e__ != hipSuccess) {                                               \
      sacv_set_error("%s: launch failed: %s", name, hipGetErrorString(e__)); \
      return SACV_ERR_LAUNCH;                                              \
    }                                                                      \
  } while (0)

static inline int sacv_cdiv(long long a, long long b) { return (int)((a + b - 1) / b); }

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
  return v;
}
__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
  return v;
}
