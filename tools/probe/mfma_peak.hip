// Sustained v_mfma_f32_32x32x2_f32 rate of the whole chip (4 waves per CU, one per SIMD, 4 independent
// accumulator tiles per wave): the practical ceiling the fp32 MFMA kernels are measured against.
// build: hipcc --offload-arch=gfx950 -O3 -o mfma_peak mfma_peak.hip ; run: ./mfma_peak [waves_per_simd]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x16 __attribute__((ext_vector_type(16)));
__global__ __launch_bounds__(256) void mfma_loop(float* out, const float* in, int iters, float a0, float b0) {
  f32x16 c0 = {0}, c1 = {0}, c2 = {0}, c3 = {0};
  // in == nullptr: trivial operands (1.0, 0.5); else per-lane pseudo-random operands in [-1, 1) -- the chip holds a
  // lower clock on non-trivial data, which is what real kernels see
  float a = a0, b = b0;
  if (in) { a = in[threadIdx.x + 256 * (blockIdx.x & 63)]; b = in[16384 + threadIdx.x + 256 * (blockIdx.x & 63)]; }
  unsigned long long t0 = 0, r0 = 0;
  if (threadIdx.x == 0) { t0 = __builtin_amdgcn_s_memtime(); r0 = __builtin_amdgcn_s_memrealtime(); }
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      c0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c0, 0, 0, 0);
      c1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c1, 0, 0, 0);
      c2 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c2, 0, 0, 0);
      c3 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c3, 0, 0, 0);
    }
  }
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    out[1024] = (float)((double)(t1 - t0) / (double)(r1 - r0) * 100.0);        // in-kernel clock, MHz
  }
  float s = 0.f;
  for (int r = 0; r < 16; ++r) s += c0[r] + c1[r] + c2[r] + c3[r];
  if (s == 12345.678f) out[threadIdx.x] = s;
}
int main(int argc, char** argv) {
  const int wps = argc > 1 ? atoi(argv[1]) : 1;
  float* out; hipMalloc(&out, 8192);
  float* in; hipMalloc(&in, 32768 * 4);
  { float h[32768]; unsigned x = 12345u; for (int i = 0; i < 32768; ++i) { x = x * 1664525u + 1013904223u; h[i] = (float)(int)(x >> 8) / 8388608.0f - 1.0f; } hipMemcpy(in, h, sizeof(h), hipMemcpyHostToDevice); }
  hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
  const int blocks = p.multiProcessorCount * wps;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int rep = 0; rep < 6; ++rep) {
    const int iters = rep == 0 ? 1000 : 40000;
    const float* src = rep >= 3 ? in : nullptr;
    hipEventRecord(e0);
    mfma_loop<<<blocks, 256>>>(out, src, iters, 1.0f, 0.5f);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double fl = (double)blocks * 4 * iters * 64.0 * 2.0 * 32 * 32 * 2;
    float mhz = 0; hipMemcpy(&mhz, out + 1024, 4, hipMemcpyDeviceToHost);
    printf("CUs %d  waves/SIMD %d  %s operands  iters %d  %.3f ms  %.1f TFLOP/s  in-kernel clock %.0f MHz\n", p.multiProcessorCount,
           wps, src ? "random " : "trivial", iters, ms, fl / ms / 1e9, mhz);
  }
  return 0;
}
