// Achievable 1R+1W and 2R+1W streaming rates (16 B per lane) for different grid sizes / unroll depths / cache policies:
// the ceiling for the BatchNorm / activation elementwise passes.
// build: hipcc --offload-arch=gfx950 -O3 -o tools/probe/copy_probe tools/probe/copy_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int U, bool NT, int NR>
__global__ __launch_bounds__(256) void k(const f32x4* __restrict__ a, const f32x4* __restrict__ b, f32x4* __restrict__ o, long long n) {
  const long long stride = (long long)gridDim.x * 256;
  long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  for (; i + (U - 1) * stride < n; i += U * stride) {
    f32x4 x[U], y[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      x[u] = NT ? __builtin_nontemporal_load(a + i + u * stride) : a[i + u * stride];
      if (NR == 2) y[u] = NT ? __builtin_nontemporal_load(b + i + u * stride) : b[i + u * stride];
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      f32x4 r = x[u] * 1.5f;
      if (NR == 2) r += y[u];
      if (NT) __builtin_nontemporal_store(r, o + i + u * stride); else o[i + u * stride] = r;
    }
  }
}
template <int U, bool NT, int NR>
void run(const f32x4* a, const f32x4* b, f32x4* o, long long n, int blocks) {
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  k<U, NT, NR><<<blocks, 256>>>(a, b, o, n);
  (void)hipEventRecord(e0);
  for (int r = 0; r < 3; ++r) k<U, NT, NR><<<blocks, 256>>>(a, b, o, n);
  (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1); ms /= 3;
  printf("%dR1W  unroll %d  %s  blocks %5d : %7.3f ms  %6.0f GB/s\n", NR, U, NT ? "nontemporal" : "default    ", blocks, ms,
         (NR + 1) * n * 16.0 / ms / 1e6);
}
int main() {
  const long long n = 32ll * 416 * 416 * 256 / 4;      // float4 elements of a conv_2-sized tensor (5.67 GB)
  f32x4 *a, *b, *o;
  (void)hipMalloc(&a, n * 16); (void)hipMalloc(&b, n * 16); (void)hipMalloc(&o, n * 16);
  (void)hipMemset(a, 0, n * 16); (void)hipMemset(b, 0, n * 16);
  for (int blocks : {1024, 2048, 4096, 8192, 16384, 65536}) {
    run<1, false, 1>(a, b, o, n, blocks);
    run<4, false, 1>(a, b, o, n, blocks);
    run<4, true, 1>(a, b, o, n, blocks);
    run<8, true, 1>(a, b, o, n, blocks);
  }
  for (int blocks : {2048, 4096, 16384}) {
    run<2, false, 2>(a, b, o, n, blocks);
    run<2, true, 2>(a, b, o, n, blocks);
    run<4, true, 2>(a, b, o, n, blocks);
  }
  return 0;
}
