// Sustained v_mfma_f32_32x32x2_f32 rate of the whole chip (4 waves per CU, one per SIMD, 4 independent
// accumulator tiles per wave): the practical ceiling the fp32 MFMA kernels are measured against.
// build: hipcc --offload-arch=gfx950 -O3 -o mfma_peak mfma_peak.hip ; run: ./mfma_peak [waves_per_simd]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x16 __attribute__((ext_vector_type(16)));
__global__ __launch_bounds__(256) void mfma_loop(float* out, int iters, float a, float b) {
  f32x16 c0 = {0}, c1 = {0}, c2 = {0}, c3 = {0};
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      c0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c0, 0, 0, 0);
      c1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c1, 0, 0, 0);
      c2 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c2, 0, 0, 0);
      c3 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c3, 0, 0, 0);
    }
  }
  float s = 0.f;
  for (int r = 0; r < 16; ++r) s += c0[r] + c1[r] + c2[r] + c3[r];
  if (s == 12345.678f) out[threadIdx.x] = s;
}
int main(int argc, char** argv) {
  const int wps = argc > 1 ? atoi(argv[1]) : 1;
  float* out; hipMalloc(&out, 4096);
  hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
  const int blocks = p.multiProcessorCount * wps;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int rep = 0; rep < 4; ++rep) {
    const int iters = rep == 0 ? 1000 : 40000 * rep;       // ~ 4, 8, 12 ms at peak for 1 wave/SIMD
    hipEventRecord(e0);
    mfma_loop<<<blocks, 256>>>(out, iters, 1.0f, 0.5f);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double fl = (double)blocks * 4 * iters * 64.0 * 2.0 * 32 * 32 * 2;
    printf("CUs %d clock %d MHz  waves/SIMD %d  iters %d  %.3f ms  %.1f TFLOP/s  -> %.0f MHz effective\n", p.multiProcessorCount,
           p.clockRate / 1000, wps, iters, ms, fl / ms / 1e9, fl / ms / 1e9 / 157.3 * 2400);
  }
  return 0;
}
