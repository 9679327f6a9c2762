// Shared declarations for the gfx950 kernels behind include/capsyolo_hip.h.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "capsyolo_hip.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
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

// ---- internal interface of routing_rows.hip (the row-stationary routing pass for C > 1), used by routing.hip
typedef struct {
  const float* u; const float* Wp;                       // Wp: W repacked by cyi_rows_pack_w (the LDS image layout)
  const float* V; const float* ds; float* slab;          // one pass of a phased plan: V_t (null for t = 0), ds^t, partial sums
  float* s_hist; float* v_out;                           // fused forward (s_hist is read by the fused backward)
  const float* dv; float* ds_all; float* V_all;          // fused backward
  float* cdb;                                            // fused backward, optional: c^t and db^t of every (t >= 1, row, i, j) for routing_caps.hip
  int R, N, C, n_iter, it, ic, g, B, fused;
} cyi_rows_args_t;
typedef struct { int slots, nj, wps, waves, rows_per_block, row_blocks, nch, ic, phased; } cyi_rows_plan_t;
void cyi_rows_plan(int R, int N, int C, int Dout, int mode, cyi_rows_plan_t* p);
int cyi_rows_launch(int mode, const cyi_rows_args_t* a, const cyi_rows_plan_t* p, int Dout, hipStream_t s);
long long cyi_rows_wp_floats(int N, int C, int Dout);                  // floats of the packed W image
int cyi_rows_pack_w(const float* W, float* Wp, int N, int C, int Dout, hipStream_t s);

// ---- routing_mfma.hip: the forward pass with u_hat = u W on v_mfma_f32_16x16x4_f32 (row tiles of 16; C <= 48, Dout 16 / 21)
bool cyi_mfma_ok(int C, int Dout);
long long cyi_mfma_wp_floats(int N, int C, int Dout);                  // floats of its packed W image
int cyi_mfma_pack_w(const float* W, float* Wp, int N, int C, int Dout, hipStream_t s);
int cyi_mfma_launch(const cyi_rows_args_t* a, int nch, int Dout, hipStream_t s);   // a->fused: all iterations; else one phased iteration over nch chunks of a->ic

int cyi_caps_bwd_launch(const cy_routing_bwd_t* a, const float* cdb, hipStream_t s);   // routing_caps.hip: du and dW (cdb: saved couplings or NULL)

#ifdef __HIPCC__
// offset (in floats) of the 8-vector of input capsule i of capsule row `row`: contiguous rows [R][N][8], or (g != 0)
// the DarkCapsuleNet cell gather (models.py:393-398) read in place from the NHWC feature map [B][4g][4g][256]
__device__ __forceinline__ long long u_offset_g(int row, int i, int N, int g, int B) {
  if (g == 0) return ((long long)row * N + i) * 8;
  const int k = row / B, b = row - k * B;
  const int pos = i >> 5, chg = i & 31;
  const long long pix = (long long)b * 16 * g * g + (long long)(pos >> 2) * 4 * g * g + 4 * k + (pos & 3);
  return pix * 256 + chg * 8;
}
#endif

// wave64 reductions on the DPP cross-lane network (no LDS round trips; __shfl_xor lowers to ds_bpermute_b32,
// ~100 cycles each and serialised by its s_waitcnt): quad swaps, half-row / row mirrors, then row broadcasts
// 15 and 31 leave the total in lane 63, which v_readlane returns to every lane as a scalar.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_get(float fallback, float src) {
  return __int_as_float(
      __builtin_amdgcn_update_dpp(__float_as_int(fallback), __float_as_int(src), CTRL, ROW_MASK, 0xF, false));
}
__device__ __forceinline__ float wave_sum(float v) {
  v += dpp_get<0xB1, 0xF>(0.f, v);     // quad_perm [1,0,3,2]
  v += dpp_get<0x4E, 0xF>(0.f, v);     // quad_perm [2,3,0,1]
  v += dpp_get<0x141, 0xF>(0.f, v);    // row_half_mirror
  v += dpp_get<0x140, 0xF>(0.f, v);    // row_mirror          -> every lane: sum of its 16-lane row
  v += dpp_get<0x142, 0xA>(0.f, v);    // row_bcast:15 into rows 1 and 3
  v += dpp_get<0x143, 0xC>(0.f, v);    // row_bcast:31 into rows 2 and 3 -> lane 63: total
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}
// sum over each 16-lane row only (4 DPP steps); every lane of a row ends with its row's sum
__device__ __forceinline__ float row16_sum(float v) {
  v += dpp_get<0xB1, 0xF>(0.f, v);
  v += dpp_get<0x4E, 0xF>(0.f, v);
  v += dpp_get<0x141, 0xF>(0.f, v);
  v += dpp_get<0x140, 0xF>(0.f, v);
  return v;
}
#ifdef __HIPCC__
// all-reduce over the wave with every lane holding the result: 4 DPP steps inside the 16-lane rows, then the two gfx950
// row / half swaps (v_permlane16_swap, v_permlane32_swap) -- no v_readlane, so no VALU -> SALU -> VALU round trip
__device__ __forceinline__ float swap_add16(float v) {
  typedef unsigned u32x2_ __attribute__((ext_vector_type(2)));
  const u32x2_ r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
__device__ __forceinline__ float swap_add32(float v) {
  typedef unsigned u32x2_ __attribute__((ext_vector_type(2)));
  const u32x2_ r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
__device__ __forceinline__ float wave_allsum(float v) { return swap_add32(swap_add16(row16_sum(v))); }
__device__ __forceinline__ float wave_allmax(float v) {
  typedef unsigned u32x2_ __attribute__((ext_vector_type(2)));
  v = fmaxf(v, dpp_get<0xB1, 0xF>(v, v));
  v = fmaxf(v, dpp_get<0x4E, 0xF>(v, v));
  v = fmaxf(v, dpp_get<0x141, 0xF>(v, v));
  v = fmaxf(v, dpp_get<0x140, 0xF>(v, v));
  u32x2_ r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  v = fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
  r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  return fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
}
#endif
__device__ __forceinline__ float wave_max(float v) {
  v = fmaxf(v, dpp_get<0xB1, 0xF>(v, v));
  v = fmaxf(v, dpp_get<0x4E, 0xF>(v, v));
  v = fmaxf(v, dpp_get<0x141, 0xF>(v, v));
  v = fmaxf(v, dpp_get<0x140, 0xF>(v, v));
  v = fmaxf(v, dpp_get<0x142, 0xA>(v, v));
  v = fmaxf(v, dpp_get<0x143, 0xC>(v, v));
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}
