// Floor finder for the C=1 routing stream: how fast can 256..1024 blocks read R rows of 16 KiB (HBM / MALL)?
// build: hipcc --offload-arch=gfx950 -O3 -o tools/probe/stream_probe tools/probe/stream_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int NSLOT, int DEPTH>
__global__ __launch_bounds__(256 * NSLOT) void probe(const float* __restrict__ u, float* __restrict__ out, int R) {
  const int t = threadIdx.x, lane = t & 63, w = t >> 6, q = w & 3, sl = w >> 2;
  const int lo = (int)((long long)R * blockIdx.x / gridDim.x), hi = (int)((long long)R * (blockIdx.x + 1) / gridDim.x);
  f32x4 acc = {0, 0, 0, 0};
  for (int row = lo + sl; row < hi; row += NSLOT * DEPTH) {
    f32x4 x[DEPTH][4];
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) {
      const int r = row + d * NSLOT;
      if (r < hi) {
        const f32x4* src = (const f32x4*)(u + (long long)r * 4096 + q * 1024) + lane;
#pragma unroll
        for (int j = 0; j < 4; ++j) x[d][j] = src[j * 64];
      } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) x[d][j] = f32x4{0, 0, 0, 0};
      }
    }
#pragma unroll
    for (int d = 0; d < DEPTH; ++d)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc += x[d][j];
  }
  if (acc[0] + acc[1] + acc[2] + acc[3] == 12345.678f) out[blockIdx.x * blockDim.x + t] = acc[0];
}
__global__ void empty_kernel(float* out) { if (out == nullptr) out[0] = 1.f; }

template <int NSLOT, int DEPTH>
void run(const float* u, float* out, int R, int blocks, const char* tag) {
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  for (int i = 0; i < 3; ++i) probe<NSLOT, DEPTH><<<blocks, 256 * NSLOT>>>(u, out, R);
  hipEventRecord(a);
  const int reps = 50;
  for (int i = 0; i < reps; ++i) probe<NSLOT, DEPTH><<<blocks, 256 * NSLOT>>>(u, out, R);
  hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  const double us = ms * 1e3 / reps, gb = 4.0 * R * 4096 / 1e9;
  printf("%-34s blocks %5d  %7.2f us/launch (back-to-back)  %7.1f GB/s\n", tag, blocks, us, gb / (us * 1e-6));
}
int main(int argc, char** argv) {
  const int R = argc > 1 ? atoi(argv[1]) : 5408;
  float *u, *out;
  hipMalloc(&u, (size_t)R * 4096 * 4); hipMalloc(&out, 1 << 22);
  hipMemset(u, 0, (size_t)R * 4096 * 4);
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  hipEventRecord(a); for (int i = 0; i < 100; ++i) empty_kernel<<<256, 256>>>(out); hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b); printf("empty kernel back-to-back: %.2f us\n", ms * 10);
  run<1, 2>(u, out, R, 256, "4 waves depth2");
  run<2, 2>(u, out, R, 256, "8 waves depth2");
  run<2, 4>(u, out, R, 256, "8 waves depth4");
  run<3, 2>(u, out, R, 256, "12 waves depth2");
  run<4, 2>(u, out, R, 256, "16 waves depth2");
  run<4, 1>(u, out, R, 256, "16 waves depth1");
  run<2, 2>(u, out, R, 512, "8 waves depth2");
  run<1, 2>(u, out, R, 1024, "4 waves depth2");
  run<1, 4>(u, out, R, 1352, "4 waves depth4 (1 iter)");
  run<1, 2>(u, out, R, 2048, "4 waves depth2");
  return 0;
}
