// torch.optim.Adam step (main.py:72,280) for a whole list of tensors in one launch (gfx950).
// table[k] = {param*, grad*, exp_avg*, exp_avg_sq*, numel}; blockmap[b] = {tensor, chunk}.
// Same update order as torch's single-tensor path: m, v, denom = sqrt(v)/sqrt(bc2) + eps,
// p -= (lr/bc1) * m / denom.  HBM-bound: 4 reads + 3 writes of 4 B per element.
#include "common.h"

namespace {

struct AdamEntry {
  float* p; const float* g; float* m; float* v; long long n;
};

__global__ __launch_bounds__(256) void adam_multi_kernel(const AdamEntry* __restrict__ table, const int2* __restrict__ blockmap,
                                                         int chunk, float step_size, float beta1, float beta2, float eps,
                                                         float inv_bc2_sqrt) {
  const int2 bm = blockmap[blockIdx.x];
  const AdamEntry e = table[bm.x];
  const long long begin = (long long)bm.y * chunk;
  long long end = begin + chunk;
  if (end > e.n) end = e.n;
  for (long long i = begin + threadIdx.x; i < end; i += 256) {
    const float g = e.g[i];
    const float m = beta1 * e.m[i] + (1.f - beta1) * g;
    const float v = beta2 * e.v[i] + (1.f - beta2) * g * g;
    e.m[i] = m;
    e.v[i] = v;
    const float denom = sqrtf(v) * inv_bc2_sqrt + eps;
    e.p[i] -= step_size * (m / denom);
  }
}

// The same step with its scalars read from DEVICE memory: hyper = {lr, beta1, beta2, eps, bias_corr1, bias_corr2}.  A captured HIP
// graph freezes kernel arguments; the step count (bias corrections) and a scheduler's learning rate change from step to step, so
// the host refreshes the six floats with one small copy in front of every replay (graph_step.py).
__global__ __launch_bounds__(256) void adam_multi_dev_kernel(const AdamEntry* __restrict__ table, const int2* __restrict__ blockmap,
                                                             int chunk, const float* __restrict__ hyper) {
  const float beta1 = hyper[1], beta2 = hyper[2], eps = hyper[3];
  const float step_size = hyper[0] / hyper[4], inv_bc2_sqrt = 1.f / sqrtf(hyper[5]);
  const int2 bm = blockmap[blockIdx.x];
  const AdamEntry e = table[bm.x];
  const long long begin = (long long)bm.y * chunk;
  long long end = begin + chunk;
  if (end > e.n) end = e.n;
  for (long long i = begin + threadIdx.x; i < end; i += 256) {
    const float g = e.g[i];
    const float m = beta1 * e.m[i] + (1.f - beta1) * g;
    const float v = beta2 * e.v[i] + (1.f - beta2) * g * g;
    e.m[i] = m;
    e.v[i] = v;
    const float denom = sqrtf(v) * inv_bc2_sqrt + eps;
    e.p[i] -= step_size * (m / denom);
  }
}

// flat[offset_k + i] = scale * tensor_k[i] (pack) or tensor_k[i] = scale * flat[offset_k + i] (unpack) for a whole list
// of tensors in one launch: table[k] = {tensor*, offset into flat, numel}, blockmap as above
struct CopyEntry { float* t; long long off; long long n; };
__global__ __launch_bounds__(256) void multi_copy_kernel(const CopyEntry* __restrict__ table, const int2* __restrict__ blockmap,
                                                         int chunk, float* __restrict__ flat, int unpack, float scale) {
  const int2 bm = blockmap[blockIdx.x];
  const CopyEntry e = table[bm.x];
  const long long begin = (long long)bm.y * chunk;
  long long end = begin + chunk;
  if (end > e.n) end = e.n;
  for (long long i = begin + threadIdx.x; i < end; i += 256) {
    if (unpack) e.t[i] = scale * flat[e.off + i];
    else flat[e.off + i] = scale * e.t[i];
  }
}

}  // namespace

extern "C" int cy_multi_copy(const void* table, const void* blockmap, int n_blocks, int chunk, float* flat, int unpack,
                             float scale, void* stream) {
  CY_REQUIRE(table && blockmap && flat && n_blocks > 0 && chunk > 0, "cy_multi_copy: bad arguments");
  multi_copy_kernel<<<n_blocks, 256, 0, (hipStream_t)stream>>>((const CopyEntry*)table, (const int2*)blockmap, chunk, flat,
                                                               unpack, scale);
  CY_LAUNCH_CHECK("cy_multi_copy");
  return 0;
}

extern "C" int cy_adam_multi(const void* table, const void* blockmap, int n_blocks, int chunk, float lr, float beta1,
                             float beta2, float eps, float bias_corr1, float bias_corr2, void* stream) {
  CY_REQUIRE(table && blockmap && n_blocks > 0 && chunk > 0, "cy_adam_multi: bad arguments");
  CY_REQUIRE(bias_corr1 > 0.f && bias_corr2 > 0.f, "cy_adam_multi: bias corrections must be positive");
  adam_multi_kernel<<<n_blocks, 256, 0, (hipStream_t)stream>>>((const AdamEntry*)table, (const int2*)blockmap, chunk,
                                                               lr / bias_corr1, beta1, beta2, eps,
                                                               1.f / sqrtf(bias_corr2));
  CY_LAUNCH_CHECK("cy_adam_multi");
  return 0;
}

extern "C" int cy_adam_multi_dev(const void* table, const void* blockmap, int n_blocks, int chunk, const float* hyper, void* stream) {
  CY_REQUIRE(table && blockmap && hyper && n_blocks > 0 && chunk > 0, "cy_adam_multi_dev: bad arguments");
  adam_multi_dev_kernel<<<n_blocks, 256, 0, (hipStream_t)stream>>>((const AdamEntry*)table, (const int2*)blockmap, chunk, hyper);
  CY_LAUNCH_CHECK("cy_adam_multi_dev");
  return 0;
}
