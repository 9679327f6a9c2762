// Shared declarations for the gfx950 kernels behind include/capsyolo_hip.h.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "capsyolo_hip.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

int cy_set_error(int code, const char* fmt, ...);

#define CY_REQUIRE(cond, ...)                                      \
  do {                                                             \
    if (!(cond)) return cy_set_error(CY_EINVAL, __VA_ARGS__);      \
  } while (0)

#define CY_LAUNCH_CHECK(name)                                                          \
  do {                                                                                 \
    hipError_t e__ = hipGetLastError();                                                \
    if (e__ != hipSuccess) return cy_set_error((int)e__, "%s: %s", name, hipGetErrorString(e__)); \
  } while (0)

static inline long long cy_ceil_div(long long a, long long b) { return (a + b - 1) / b; }

// opt a kernel in to more than the default dynamic-LDS limit (gfx950 has 160 KiB per CU)
template <typename K>
static inline int cy_allow_lds(K kernel, size_t bytes) {
  if (bytes <= 48 * 1024) return 0;
  if (bytes > 160 * 1024) return cy_set_error(CY_EINVAL, "kernel needs %zu bytes of LDS (> 160 KiB)", bytes);
  hipError_t e = hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
  if (e != hipSuccess) return cy_set_error((int)e, "hipFuncSetAttribute(LDS=%zu): %s", bytes, hipGetErrorString(e));
  return 0;
}

// wave64 butterfly sum (all lanes end with the total)
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) v = fmaxf(v, __shfl_xor(v, m, 64));
  return v;
}
