// Does side work issued by the SAME wave overlap with its in-flight MFMA?  One wave per SIMD (256-thread blocks,
// one block per CU); after every v_mfma_f32_32x32x2_f32 the wave issues NV independent VALU FMAs, NW ds_write_b64,
// NR ds_read_b32 or NG global loads (L2 hits).  Prints time per MFMA in cycles at the sustained clock.
// build: hipcc --offload-arch=gfx950 -O3 -o mfma_overlap mfma_overlap.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int NV, int NW, int NR, int NG>
__global__ __launch_bounds__(256) void k(float* out, const float* in, int iters, float a, float b) {
  __shared__ float lds[256 * 8 + 64];
  f32x16 c0 = {0}, c1 = {0}, c2 = {0}, c3 = {0};
  float v[8];
  for (int i = 0; i < 8; ++i) v[i] = a * (i + threadIdx.x);
  float* lp = lds + threadIdx.x * 4;
  const f32x4* gp = (const f32x4*)in + threadIdx.x;
  f32x4 g = {0, 0, 0, 0};
  float r = 0.f;
  float rr[16] = {0};
  f32x4 gg[16];
  for (int i = 0; i < 16; ++i) gg[i] = g;
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      f32x16* cc = (u & 3) == 0 ? &c0 : (u & 3) == 1 ? &c1 : (u & 3) == 2 ? &c2 : &c3;
      *cc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, *cc, 0, 0, 0);
#pragma unroll
      for (int q = 0; q < NV; ++q) v[q & 7] = __builtin_fmaf(v[q & 7], a, b);
#pragma unroll
      for (int q = 0; q < NW; ++q) *(float2*)(lp + 1024 * (q & 1)) = make_float2(v[q & 7], v[(q + 1) & 7]);
#pragma unroll
      for (int q = 0; q < NR; ++q) rr[(u * NR + q) & 15] = *(volatile float*)(lp + 64 * q);
#pragma unroll
      for (int q = 0; q < NG; ++q) gg[(u * NG + q) & 15] = __builtin_nontemporal_load(gp + 256 * ((i + q + u) & 63));
      __builtin_amdgcn_sched_barrier(0);
    }
    if (NR) for (int q = 0; q < 16; ++q) r += rr[q];
    if (NG) for (int q = 0; q < 16; ++q) g += gg[q];
  }
  float s = r + g[0] + g[1] + g[2] + g[3];
  for (int i = 0; i < 8; ++i) s += v[i];
  for (int q = 0; q < 16; ++q) s += c0[q] + c1[q] + c2[q] + c3[q];
  if (s == 12345.678f) out[threadIdx.x] = s + lds[threadIdx.x];
}
template <int NV, int NW, int NR, int NG>
void run(float* out, const float* in, int cus, int wps = 1) {
  const int blocks = cus * wps;
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  const int iters = 20000;
  k<NV, NW, NR, NG><<<blocks, 256>>>(out, in, 2000, 1.0f, 0.5f);
  (void)hipEventRecord(e0);
  k<NV, NW, NR, NG><<<blocks, 256>>>(out, in, iters, 1.0f, 0.5f);
  (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  printf("waves/SIMD %d  VALU %2d  ds_write_b64 %2d  ds_read_b32 %2d  global_load_b128 %2d : %7.3f ms  %6.1f cycles per MFMA per SIMD (2.37 GHz)\n", wps, NV, NW, NR, NG,
         ms, ms * 1e-3 * 2.37e9 / (iters * 8.0 * wps));
}
int main() {
  float *out, *in; (void)hipMalloc(&out, 4096); (void)hipMalloc(&in, 1 << 20); (void)hipMemset(in, 0, 1 << 20);
  hipDeviceProp_t p; (void)hipGetDeviceProperties(&p, 0);
  const int nb = p.multiProcessorCount;
  run<0, 0, 0, 0>(out, in, nb);
  run<4, 0, 0, 0>(out, in, nb);
  run<8, 0, 0, 0>(out, in, nb);
  run<12, 0, 0, 0>(out, in, nb);
  run<16, 0, 0, 0>(out, in, nb);
  run<24, 0, 0, 0>(out, in, nb);
  run<0, 1, 0, 0>(out, in, nb);
  run<0, 2, 0, 0>(out, in, nb);
  run<0, 4, 0, 0>(out, in, nb);
  run<0, 0, 2, 0>(out, in, nb);
  run<0, 0, 4, 0>(out, in, nb);
  run<0, 0, 0, 1>(out, in, nb);
  run<0, 0, 0, 2>(out, in, nb);
  run<0, 0, 0, 0>(out, in, nb, 2);
  run<8, 0, 0, 0>(out, in, nb, 2);
  run<12, 0, 0, 0>(out, in, nb, 2);
  run<16, 0, 0, 0>(out, in, nb, 2);
  run<24, 0, 0, 0>(out, in, nb, 2);
  run<12, 2, 0, 0>(out, in, nb, 2);
  run<8, 0, 0, 0>(out, in, nb, 3);
  run<16, 0, 0, 0>(out, in, nb, 3);
  return 0;
}
