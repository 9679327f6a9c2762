// BatchNorm2d (+LeakyReLU) on NHWC activations, training and eval mode, forward and backward.
// Replaces nn.BatchNorm2d / nn.LeakyReLU of models.py:132-223, 347-365.  HBM-bound streaming
// kernels: 16-byte accesses, one pass per tensor.  The batch statistics themselves come out of the
// convolution epilogue (conv.hip) as double sum / sum-of-squares.
#include "common.h"

namespace {

__global__ void bn_finalize_kernel(const double* __restrict__ stats, double count, const float* __restrict__ gamma,
                                   const float* __restrict__ beta, float* running_mean, float* running_var,
                                   float momentum, float eps, float* scale, float* shift, float* mean,
                                   float* invstd, int N, long long* num_batches_tracked) {
  const int n = blockIdx.x * blockDim.x + threadIdx.x;
  if (n == 0 && num_batches_tracked != nullptr) *num_batches_tracked += 1;     // nn.BatchNorm2d's step counter
  if (n >= N) return;
  double s1 = 0.0, s2 = 0.0;
  for (int cp = 0; cp < CY_STATS_COPIES; ++cp) {            // fixed order: deterministic given the copies
    s1 += stats[((size_t)cp * N + n) * 2];
    s2 += stats[((size_t)cp * N + n) * 2 + 1];
  }
  const double m = s1 / count;
  double var = s2 / count - m * m;                         // biased variance normalises the batch
  if (var < 0.0) var = 0.0;
  const float is = (float)(1.0 / sqrt(var + (double)eps));
  const float sc = gamma[n] * is;
  mean[n] = (float)m;
  invstd[n] = is;
  scale[n] = sc;
  shift[n] = beta[n] - (float)m * sc;
  if (running_mean != nullptr) {
    const double unbiased = count > 1.0 ? var * count / (count - 1.0) : var;
    running_mean[n] = (1.f - momentum) * running_mean[n] + momentum * (float)m;
    running_var[n] = (1.f - momentum) * running_var[n] + momentum * (float)unbiased;
  }
}

__global__ void bn_eval_kernel(const float* __restrict__ gamma, const float* __restrict__ beta,
                               const float* __restrict__ rm, const float* __restrict__ rv, float eps, float* scale,
                               float* shift, int N) {
  const int n = blockIdx.x * blockDim.x + threadIdx.x;
  if (n >= N) return;
  const float sc = gamma[n] / sqrtf(rv[n] + eps);
  scale[n] = sc;
  shift[n] = beta[n] - rm[n] * sc;
}

__device__ __forceinline__ float lrelu(float y, float slope) { return y > 0.f ? y : y * slope; }

// A = lrelu(Z*scale + shift); float4 path when N % 4 == 0.
// HOIST (N divides 1024): the grid stride is a multiple of N, so a thread always sees the same 4 channels: their
// scale / shift live in registers and the loop body is 4 independent 16-byte loads, the arithmetic, 4 stores --
// no 64-bit modulo, no parameter reloads per element (those, not HBM, limited the first version to 4.6 TB/s).
// NT: streaming (nontemporal) accesses for tensors that cannot stay in the 256 MiB Infinity Cache anyway; a small
// activation (conv_5's 88.6 MB at the headline shape) is stored normally so that its consumer -- the routing
// kernel, launched next -- finds it on-die instead of in HBM (caps1_fwd_kernel 21.8 -> 18.8 us inside the step).
template <bool AFFINE, bool HOIST, bool NT = true>
__global__ void affine_act_kernel(const float* __restrict__ Z, float* __restrict__ A, const float* __restrict__ scale,
                                  const float* __restrict__ shift, float slope, long long n4, int N) {
  const long long stride = (long long)gridDim.x * blockDim.x;
  long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (HOIST) {
    float4 sc = make_float4(1.f, 1.f, 1.f, 1.f), sh = make_float4(0.f, 0.f, 0.f, 0.f);
    if (AFFINE) {
      const int c = (int)((threadIdx.x * 4u) % (unsigned)N);
      sc = *(const float4*)(scale + c); sh = *(const float4*)(shift + c);
    }
    const f32x4* Zv = (const f32x4*)Z;
    f32x4* Av = (f32x4*)A;
    for (; i + 3 * stride < n4; i += 4 * stride) {
      f32x4 z[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) z[u] = NT ? __builtin_nontemporal_load(Zv + i + u * stride) : Zv[i + u * stride];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        f32x4 y;
        y[0] = lrelu(z[u][0] * sc.x + sh.x, slope); y[1] = lrelu(z[u][1] * sc.y + sh.y, slope);
        y[2] = lrelu(z[u][2] * sc.z + sh.z, slope); y[3] = lrelu(z[u][3] * sc.w + sh.w, slope);
        if (NT) __builtin_nontemporal_store(y, Av + i + u * stride);
        else Av[i + u * stride] = y;
      }
    }
    for (; i < n4; i += stride) {
      const f32x4 z = Zv[i];
      f32x4 y;
      y[0] = lrelu(z[0] * sc.x + sh.x, slope); y[1] = lrelu(z[1] * sc.y + sh.y, slope);
      y[2] = lrelu(z[2] * sc.z + sh.z, slope); y[3] = lrelu(z[3] * sc.w + sh.w, slope);
      Av[i] = y;
    }
    return;
  }
  for (; i < n4; i += stride) {
    float4 z = ((const float4*)Z)[i];
    if (AFFINE) {
      const int c = (int)((i * 4) % N);
      const float4 sc = *(const float4*)(scale + c), sh = *(const float4*)(shift + c);
      z.x = z.x * sc.x + sh.x; z.y = z.y * sc.y + sh.y; z.z = z.z * sc.z + sh.z; z.w = z.w * sc.w + sh.w;
    }
    z.x = lrelu(z.x, slope); z.y = lrelu(z.y, slope); z.z = lrelu(z.z, slope); z.w = lrelu(z.w, slope);
    ((float4*)A)[i] = z;
  }
}
__global__ void affine_act_scalar_kernel(const float* __restrict__ Z, float* __restrict__ A,
                                         const float* __restrict__ scale, const float* __restrict__ shift,
                                         float slope, long long n, int N) {
  const long long stride = (long long)gridDim.x * blockDim.x;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    float z = Z[i];
    if (scale != nullptr) { const int c = (int)(i % N); z = z * scale[c] + shift[c]; }
    A[i] = lrelu(z, slope);
  }
}

// pass 1 of the backward: per-channel sums of dYhat and dYhat*xhat.
// Thread -> 4 channels (float4) of one row; a block works a SLICE of CW = min(N, 64) channels (LPR = CW/4 lanes per row,
// 256/LPR rows in parallel) over a range of rows: blockIdx = row block * (N / CW) + slice.  (One block per row range over
// ALL channels meant 2 N double atomics per block: at 13 x 13 x 1024 channels 676 blocks sent 1.4 M atomics to 2048
// addresses for 22 MB of data -- DarkNet's 17 such launches took 1.09 ms per step.)
__global__ void bn_bwd_reduce_kernel(const float* __restrict__ Z, const float* __restrict__ dA,
                                     const float* __restrict__ scale, const float* __restrict__ shift,
                                     const float* __restrict__ mean, const float* __restrict__ invstd, float slope,
                                     double* red, long long P, int N, long long rows_per_block) {
  extern __shared__ __attribute__((aligned(16))) double smd[];   // [256][8]
  const int CW = N < 64 ? N : 64, nsl = N / CW;
  const int LPR = CW >> 2;
  const int rpar = 256 / LPR;
  const int t = threadIdx.x;
  const int u = t % LPR, rsub = t / LPR;
  const int c0 = (int)(blockIdx.x % nsl) * CW;
  const int c = c0 + u * 4;
  const float4 sc = *(const float4*)(scale + c), sh = *(const float4*)(shift + c);
  const float4 mu = *(const float4*)(mean + c), is = *(const float4*)(invstd + c);
  const long long r0 = (long long)(blockIdx.x / nsl) * rows_per_block;
  long long r1 = r0 + rows_per_block;
  if (r1 > P) r1 = P;
  // per-thread sums in double (the kernel waits for memory, not for the adds): the two sums then carry the rounding of
  // their fp32 terms only, whatever the number of rows a thread walks
  struct D4 { double x, y, z, w; };
  D4 s1 = {0.0, 0.0, 0.0, 0.0}, s2 = s1;
#define CY_ACC(z, g, f)                                             \
    {                                                               \
      const float y = z.f * sc.f + sh.f;                            \
      const float d = y > 0.f ? g.f : g.f * slope;                  \
      s1.f += (double)d;                                            \
      s2.f += (double)(d * ((z.f - mu.f) * is.f));                  \
    }
  long long r = r0 + rsub;
  const long long step = (long long)rpar * N;
  const float* zp = Z + r * N + c;
  const float* gp = dA + r * N + c;
  for (; r + 3 * rpar < r1; r += 4 * rpar, zp += 4 * step, gp += 4 * step) {   // 8 independent 16-byte loads in flight
    const float4 z0 = *(const float4*)(zp), z1 = *(const float4*)(zp + step), z2 = *(const float4*)(zp + 2 * step),
                 z3 = *(const float4*)(zp + 3 * step);
    const float4 g0 = *(const float4*)(gp), g1 = *(const float4*)(gp + step), g2 = *(const float4*)(gp + 2 * step),
                 g3 = *(const float4*)(gp + 3 * step);
    CY_ACC(z0, g0, x) CY_ACC(z0, g0, y) CY_ACC(z0, g0, z) CY_ACC(z0, g0, w)
    CY_ACC(z1, g1, x) CY_ACC(z1, g1, y) CY_ACC(z1, g1, z) CY_ACC(z1, g1, w)
    CY_ACC(z2, g2, x) CY_ACC(z2, g2, y) CY_ACC(z2, g2, z) CY_ACC(z2, g2, w)
    CY_ACC(z3, g3, x) CY_ACC(z3, g3, y) CY_ACC(z3, g3, z) CY_ACC(z3, g3, w)
  }
  for (; r < r1; r += rpar, zp += step, gp += step) {
    const float4 z = *(const float4*)(zp);
    const float4 g = *(const float4*)(gp);
    CY_ACC(z, g, x) CY_ACC(z, g, y) CY_ACC(z, g, z) CY_ACC(z, g, w)
  }
#undef CY_ACC
  double* my = smd + t * 8;
  my[0] = s1.x; my[1] = s1.y; my[2] = s1.z; my[3] = s1.w;
  my[4] = s2.x; my[5] = s2.y; my[6] = s2.z; my[7] = s2.w;
  __syncthreads();
  // thread t < 2*CW handles (channel n = c0 + t % CW, which = t / CW)
  for (int idx = t; idx < 2 * CW; idx += 256) {
    const int n = idx % CW, which = idx / CW;
    const int uu = n >> 2, e = n & 3;
    double acc = 0.0;
    for (int rs = 0; rs < rpar; ++rs) acc += smd[(rs * LPR + uu) * 8 + which * 4 + e];
    atomicAdd(red + 2 * (c0 + n) + which, acc);
  }
}

template <bool HOIST>
__global__ void bn_bwd_apply_kernel(const float* __restrict__ Z, const float* __restrict__ dA, float* __restrict__ dZ,
                                    const float* __restrict__ scale, const float* __restrict__ shift,
                                    const float* __restrict__ mean, const float* __restrict__ invstd, float slope,
                                    const double* __restrict__ red, double inv_count, long long n4, int N) {
  const long long stride = (long long)gridDim.x * blockDim.x;
  long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (HOIST) {                              // N divides 1024: per-channel constants in registers (see affine_act_kernel)
    const int c = (int)((threadIdx.x * 4u) % (unsigned)N);
    const float4 sc = *(const float4*)(scale + c), sh = *(const float4*)(shift + c);
    const float4 mu = *(const float4*)(mean + c), is = *(const float4*)(invstd + c);
    float m1[4], m2[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      m1[k] = (float)(red[2 * (c + k)] * inv_count);
      m2[k] = (float)(red[2 * (c + k) + 1] * inv_count);
    }
    const float scv[4] = {sc.x, sc.y, sc.z, sc.w}, shv[4] = {sh.x, sh.y, sh.z, sh.w};
    const float muv[4] = {mu.x, mu.y, mu.z, mu.w}, isv[4] = {is.x, is.y, is.z, is.w};
    const f32x4* Zv = (const f32x4*)Z;
    const f32x4* Gv = (const f32x4*)dA;
    f32x4* Ov = (f32x4*)dZ;
    auto one = [&](const f32x4 z, const f32x4 g) {
      f32x4 o;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const float y = z[k] * scv[k] + shv[k];
        const float d = y > 0.f ? g[k] : g[k] * slope;
        const float xh = (z[k] - muv[k]) * isv[k];
        o[k] = scv[k] * (d - m1[k] - xh * m2[k]);
      }
      return o;
    };
    for (; i + stride < n4; i += 2 * stride) {
      const f32x4 z0 = __builtin_nontemporal_load(Zv + i), z1 = __builtin_nontemporal_load(Zv + i + stride);
      const f32x4 g0 = __builtin_nontemporal_load(Gv + i), g1 = __builtin_nontemporal_load(Gv + i + stride);
      __builtin_nontemporal_store(one(z0, g0), Ov + i);
      __builtin_nontemporal_store(one(z1, g1), Ov + i + stride);
    }
    for (; i < n4; i += stride) Ov[i] = one(Zv[i], Gv[i]);
    return;
  }
  for (; i < n4; i += stride) {
    const int c = (int)((i * 4) % N);
    const float4 z = ((const float4*)Z)[i], g = ((const float4*)dA)[i];
    const float4 sc = *(const float4*)(scale + c), sh = *(const float4*)(shift + c);
    const float4 mu = *(const float4*)(mean + c), is = *(const float4*)(invstd + c);
    float4 o;
#define CY_APPLY(f, k)                                                       \
    {                                                                        \
      const float y = z.f * sc.f + sh.f;                                     \
      const float d = y > 0.f ? g.f : g.f * slope;                           \
      const float xh = (z.f - mu.f) * is.f;                                  \
      const float m1 = (float)(red[2 * (c + k)] * inv_count);                \
      const float m2 = (float)(red[2 * (c + k) + 1] * inv_count);            \
      o.f = sc.f * (d - m1 - xh * m2);                                       \
    }
    CY_APPLY(x, 0) CY_APPLY(y, 1) CY_APPLY(z, 2) CY_APPLY(w, 3)
#undef CY_APPLY
    ((float4*)dZ)[i] = o;
  }
}

// red_out[c][k] = scale * sum over the striped copies; optional float copies of the two columns (dbeta, dgamma)
__global__ void bn_red_fold_kernel(const double* __restrict__ red_copies, int copies, double scale,
                                   double* __restrict__ red_out, float* dgamma, float* dbeta, int N) {
  const int n = blockIdx.x * blockDim.x + threadIdx.x;
  if (n >= N) return;
  double s0 = 0.0, s1 = 0.0;
  for (int cp = 0; cp < copies; ++cp) {                    // fixed order: deterministic given the copies
    s0 += red_copies[((size_t)cp * N + n) * 2];
    s1 += red_copies[((size_t)cp * N + n) * 2 + 1];
  }
  s0 *= scale; s1 *= scale;
  if (red_out != nullptr) { red_out[2 * n] = s0; red_out[2 * n + 1] = s1; }
  if (dbeta != nullptr) dbeta[n] = (float)s0;
  if (dgamma != nullptr) dgamma[n] = (float)s1;
}

__global__ void bn_param_grad_kernel(const double* __restrict__ red, float* dgamma, float* dbeta, int N) {
  const int n = blockIdx.x * blockDim.x + threadIdx.x;
  if (n >= N) return;
  dbeta[n] = (float)red[2 * n];
  dgamma[n] = (float)red[2 * n + 1];
}

__global__ void act_bwd_kernel(const float* __restrict__ Z, const float* __restrict__ dA, float* __restrict__ dZ,
                               float slope, long long n) {
  const long long stride = (long long)gridDim.x * blockDim.x;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    const float g = dA[i];
    dZ[i] = Z[i] > 0.f ? g : g * slope;
  }
}

// grid of a grid-stride streaming kernel.  Measured with tools/probe/copy_probe.hip on a 5.67 GB tensor: 4096 blocks
// 5.8 TB/s, 16384 blocks 6.3 TB/s, 65536 blocks 6.6 TB/s (1R+1W, 4 x 16 B in flight per thread, nontemporal): many
// short blocks balance the 8 XCDs better than a persistent-sized grid.
inline unsigned stream_grid(long long work_items) {
  long long b = cy_ceil_div(work_items, 256 * 4);
  if (b > 65536) b = 65536;
  if (b < 1) b = 1;
  return (unsigned)b;
}
inline bool pow2(int v) { return v > 0 && (v & (v - 1)) == 0; }

}  // namespace

extern "C" int cy_bn_finalize(const double* stats, long long count, const float* gamma, const float* beta,
                              float* running_mean, float* running_var, float momentum, float eps, float* scale,
                              float* shift, float* mean, float* invstd, int N, long long* num_batches_tracked,
                              void* stream) {
  CY_REQUIRE(stats && gamma && beta && scale && shift && mean && invstd && N > 0 && count > 0,
             "cy_bn_finalize: bad arguments");
  CY_REQUIRE((running_mean == nullptr) == (running_var == nullptr), "cy_bn_finalize: running stats must come in pairs");
  bn_finalize_kernel<<<(N + 255) / 256, 256, 0, (hipStream_t)stream>>>(stats, (double)count, gamma, beta, running_mean,
                                                                       running_var, momentum, eps, scale, shift, mean,
                                                                       invstd, N, num_batches_tracked);
  CY_LAUNCH_CHECK("cy_bn_finalize");
  return 0;
}

extern "C" int cy_bn_red_fold(const double* red_copies, int copies, double scale, double* red_out, float* dgamma,
                              float* dbeta, int N, void* stream) {
  CY_REQUIRE(red_copies && copies > 0 && N > 0 && (red_out || dgamma || dbeta), "cy_bn_red_fold: bad arguments");
  bn_red_fold_kernel<<<(N + 255) / 256, 256, 0, (hipStream_t)stream>>>(red_copies, copies, scale, red_out, dgamma, dbeta, N);
  CY_LAUNCH_CHECK("cy_bn_red_fold");
  return 0;
}

extern "C" int cy_bn_eval_scale_shift(const float* gamma, const float* beta, const float* running_mean,
                                      const float* running_var, float eps, float* scale, float* shift, int N,
                                      void* stream) {
  CY_REQUIRE(gamma && beta && running_mean && running_var && scale && shift && N > 0, "cy_bn_eval_scale_shift: bad arguments");
  bn_eval_kernel<<<(N + 255) / 256, 256, 0, (hipStream_t)stream>>>(gamma, beta, running_mean, running_var, eps, scale,
                                                                   shift, N);
  CY_LAUNCH_CHECK("cy_bn_eval_scale_shift");
  return 0;
}

// Eval-mode BatchNorm folded into the convolution in front of it (predict_fns.py:38-43, 65-69 run models.py:132-223, 347-365 in
// eval mode): BN(conv(x; W) + b) = conv(x; W') + b' with s = gamma / sqrt(running_var + eps), W'[co] = W[co] s[co],
// b'[co] = (b[co] - running_mean[co]) s[co] + beta[co].  One thread per weight; thread 0 of each channel writes b'.
__global__ __launch_bounds__(256) void bn_fold_eval_kernel(const float* __restrict__ W, const float* __restrict__ bias,
                                                           const float* __restrict__ gamma, const float* __restrict__ beta,
                                                           const float* __restrict__ rmean, const float* __restrict__ rvar, float eps,
                                                           float* __restrict__ Wf, float* __restrict__ bf, int Cout, int per_out) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= (long long)Cout * per_out) return;
  const int co = (int)(i / per_out);
  const float sc = gamma[co] / sqrtf(rvar[co] + eps);
  Wf[i] = W[i] * sc;
  if (i == (long long)co * per_out) bf[co] = ((bias != nullptr ? bias[co] : 0.f) - rmean[co]) * sc + beta[co];
}

extern "C" int cy_bn_fold_eval(const float* W, const float* bias, const float* gamma, const float* beta, const float* running_mean,
                               const float* running_var, float eps, float* Wf, float* bf, int Cout, int per_out, void* stream) {
  CY_REQUIRE(W && gamma && beta && running_mean && running_var && Wf && bf && Cout > 0 && per_out > 0, "cy_bn_fold_eval: bad arguments");
  const long long n = (long long)Cout * per_out;
  bn_fold_eval_kernel<<<(unsigned)((n + 255) / 256), 256, 0, (hipStream_t)stream>>>(W, bias, gamma, beta, running_mean, running_var,
                                                                                    eps, Wf, bf, Cout, per_out);
  CY_LAUNCH_CHECK("cy_bn_fold_eval");
  return 0;
}

extern "C" int cy_affine_act(const float* Z, float* A, const float* scale, const float* shift, float slope,
                             long long P, int N, void* stream) {
  CY_REQUIRE(Z && A && P > 0 && N > 0, "cy_affine_act: bad arguments");
  CY_REQUIRE((scale == nullptr) == (shift == nullptr), "cy_affine_act: scale/shift must come in pairs");
  hipStream_t s = (hipStream_t)stream;
  const long long n = P * N;
  const bool v4 = (N % 4 == 0) && (((uintptr_t)Z & 15) == 0) && (((uintptr_t)A & 15) == 0);
  const bool hoist = v4 && N <= 1024 && (1024 % N) == 0;
  const bool nt = n * 4 > (128ll << 20);
  if (hoist && scale && !nt) affine_act_kernel<true, true, false><<<stream_grid(n / 4), 256, 0, s>>>(Z, A, scale, shift, slope, n / 4, N);
  else if (hoist && scale) affine_act_kernel<true, true><<<stream_grid(n / 4), 256, 0, s>>>(Z, A, scale, shift, slope, n / 4, N);
  else if (hoist) affine_act_kernel<false, true><<<stream_grid(n / 4), 256, 0, s>>>(Z, A, scale, shift, slope, n / 4, N);
  else if (v4 && scale) affine_act_kernel<true, false><<<stream_grid(n / 4), 256, 0, s>>>(Z, A, scale, shift, slope, n / 4, N);
  else if (v4) affine_act_kernel<false, false><<<stream_grid(n / 4), 256, 0, s>>>(Z, A, scale, shift, slope, n / 4, N);
  else affine_act_scalar_kernel<<<stream_grid(n), 256, 0, s>>>(Z, A, scale, shift, slope, n, N);
  CY_LAUNCH_CHECK("cy_affine_act");
  return 0;
}

extern "C" int cy_bn_bwd_reduce(const float* Z, const float* dA, const float* scale, const float* shift,
                                const float* mean, const float* invstd, float slope, double* red, long long P, int N,
                                void* stream) {
  CY_REQUIRE(Z && dA && scale && shift && mean && invstd && red && P > 0, "cy_bn_bwd_reduce: bad arguments");
  CY_REQUIRE(N % 4 == 0 && pow2(N / 4) && N / 4 <= 256, "cy_bn_bwd_reduce: N=%d must be 4*2^k <= 1024", N);
  hipStream_t s = (hipStream_t)stream;
  hipError_t e = hipMemsetAsync(red, 0, (size_t)N * 2 * sizeof(double), s);
  if (e != hipSuccess) return cy_set_error((int)e, "cy_bn_bwd_reduce: memset: %s", hipGetErrorString(e));
  const int cw = N < 64 ? N : 64, nsl = N / cw;
  const int rpar = 256 / (cw / 4);
  long long rows_per_block = cy_ceil_div(P, cy_ceil_div(2048, nsl));     // about 2048 blocks
  if (rows_per_block < 4 * rpar) rows_per_block = 4 * rpar;
  rows_per_block = cy_ceil_div(rows_per_block, rpar) * rpar;
  const long long blocks = cy_ceil_div(P, rows_per_block) * nsl;
  bn_bwd_reduce_kernel<<<(unsigned)blocks, 256, 256 * 8 * 8, s>>>(Z, dA, scale, shift, mean, invstd, slope, red, P, N,
                                                                 rows_per_block);
  CY_LAUNCH_CHECK("cy_bn_bwd_reduce");
  return 0;
}

namespace {
// ---- conv -> BatchNorm -> LeakyReLU -> MaxPool2d(2) blocks (DarkNet, models.py:135 ... 195), round 4: the activation never exists at full
// resolution.  Forward: y = max over the 2 x 2 window of lrelu(z * scale + shift), idx = the winning position (0 .. 3, the first of equals
// in row-major order: nn.MaxPool2d's choice); one thread = 4 channels of one pooled pixel.
__global__ void affine_act_maxpool2_kernel(const float* __restrict__ Z, const float* __restrict__ scale, const float* __restrict__ shift,
                                           float slope, float* __restrict__ Y, unsigned char* __restrict__ idx, long long n4, int Ho, int Wo,
                                           int C) {
  const int c4n = C >> 2;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long long)gridDim.x * blockDim.x) {
    long long r = i;
    const int c = (int)(r % c4n) * 4; r /= c4n;
    const int ox = (int)(r % Wo); r /= Wo;
    const int oy = (int)(r % Ho); r /= Ho;
    const long long base = ((r * (2 * Ho) + 2 * oy) * (2 * Wo) + 2 * ox) * C + c;
    const long long rs = (long long)2 * Wo * C;
    const float4 sc = *(const float4*)(scale + c), sh = *(const float4*)(shift + c);
    const float4 z0 = *(const float4*)(Z + base), z1 = *(const float4*)(Z + base + C), z2 = *(const float4*)(Z + base + rs),
                 z3 = *(const float4*)(Z + base + rs + C);
    float4 best;
    uchar4 bi;
#define CY_POOL(f)                                                          \
    {                                                                       \
      float a = z0.f * sc.f + sh.f; a = fmaxf(a, a * slope);                \
      float b = z1.f * sc.f + sh.f; b = fmaxf(b, b * slope);                \
      float c_ = z2.f * sc.f + sh.f; c_ = fmaxf(c_, c_ * slope);            \
      float d = z3.f * sc.f + sh.f; d = fmaxf(d, d * slope);                \
      float m = a; unsigned char k = 0;                                     \
      if (b > m) { m = b; k = 1; }                                          \
      if (c_ > m) { m = c_; k = 2; }                                        \
      if (d > m) { m = d; k = 3; }                                          \
      best.f = m; bi.f = k;                                                 \
    }
    CY_POOL(x) CY_POOL(y) CY_POOL(z) CY_POOL(w)
#undef CY_POOL
    *(float4*)(Y + i * 4) = best;
    *(uchar4*)(idx + i * 4) = bi;
  }
}

// Backward of the same: dP (gradient of the pooled activation) goes to the winning position, times the activation's derivative there --
// d = [pos == idx] dP lrelu'(z * scale + shift), the PREMASKED gradient the producer block's backward takes -- and the block's
// BatchNorm-backward sums (sum d, sum d xhat over the pixels, the cy_bn_bwd_reduce contract) are added on the way: red[N][2] doubles,
// zeroed by the caller.  Thread layout of bn_bwd_reduce_kernel with pooled pixels as the rows.
__global__ void maxpool2_bwd_bn_kernel(const float* __restrict__ dP, const unsigned char* __restrict__ idx, const float* __restrict__ Z,
                                       const float* __restrict__ scale, const float* __restrict__ shift, const float* __restrict__ mean,
                                       const float* __restrict__ invstd, float slope, float* __restrict__ D, double* red, long long Pp,
                                       int Ho, int Wo, int N, long long rows_per_block) {
  extern __shared__ __attribute__((aligned(16))) double smd[];   // [256][8]
  const int CW = N < 64 ? N : 64, nsl = N / CW;
  const int LPR = CW >> 2;
  const int rpar = 256 / LPR;
  const int t = threadIdx.x;
  const int u = t % LPR, rsub = t / LPR;
  const int c0 = (int)(blockIdx.x % nsl) * CW;
  const int c = c0 + u * 4;
  const float4 sc = *(const float4*)(scale + c), sh = *(const float4*)(shift + c);
  const float4 mu = *(const float4*)(mean + c), is = *(const float4*)(invstd + c);
  const long long r0 = (long long)(blockIdx.x / nsl) * rows_per_block;
  long long r1 = r0 + rows_per_block;
  if (r1 > Pp) r1 = Pp;
  struct D4 { double x, y, z, w; };
  D4 s1 = {0.0, 0.0, 0.0, 0.0}, s2 = s1;
  const long long rs = (long long)2 * Wo * N;
  for (long long r = r0 + rsub; r < r1; r += rpar) {
    long long q = r;
    const int ox = (int)(q % Wo); q /= Wo;
    const int oy = (int)(q % Ho); q /= Ho;
    const long long base = ((q * (2 * Ho) + 2 * oy) * (2 * Wo) + 2 * ox) * N + c;
    const float4 g = *(const float4*)(dP + r * N + c);
    const uchar4 bi = *(const uchar4*)(idx + r * N + c);
    const float4 z0 = *(const float4*)(Z + base), z1 = *(const float4*)(Z + base + N), z2 = *(const float4*)(Z + base + rs),
                 z3 = *(const float4*)(Z + base + rs + N);
    float4 d0, d1, d2, d3;
#define CY_PB(f)                                                                                   \
    {                                                                                              \
      const float zw = bi.f == 0 ? z0.f : bi.f == 1 ? z1.f : bi.f == 2 ? z2.f : z3.f;              \
      const float y = zw * sc.f + sh.f;                                                            \
      const float d = y > 0.f ? g.f : g.f * slope;                                                 \
      d0.f = bi.f == 0 ? d : 0.f; d1.f = bi.f == 1 ? d : 0.f; d2.f = bi.f == 2 ? d : 0.f; d3.f = bi.f == 3 ? d : 0.f; \
      s1.f += (double)d;                                                                           \
      s2.f += (double)(d * ((zw - mu.f) * is.f));                                                  \
    }
    CY_PB(x) CY_PB(y) CY_PB(z) CY_PB(w)
#undef CY_PB
    *(float4*)(D + base) = d0; *(float4*)(D + base + N) = d1; *(float4*)(D + base + rs) = d2; *(float4*)(D + base + rs + N) = d3;
  }
  double* my = smd + t * 8;
  my[0] = s1.x; my[1] = s1.y; my[2] = s1.z; my[3] = s1.w;
  my[4] = s2.x; my[5] = s2.y; my[6] = s2.z; my[7] = s2.w;
  __syncthreads();
  for (int id = t; id < 2 * CW; id += 256) {
    const int n = id % CW, which = id / CW;
    const int uu = n >> 2, e = n & 3;
    double acc = 0.0;
    for (int k = 0; k < rpar; ++k) acc += smd[(k * LPR + uu) * 8 + which * 4 + e];
    atomicAdd(red + 2 * (c0 + n) + which, acc);
  }
}
}  // namespace

extern "C" int cy_affine_act_maxpool2(const float* Z, const float* scale, const float* shift, float slope, float* Y, unsigned char* idx,
                                      int B, int Ho, int Wo, int C, void* stream) {
  CY_REQUIRE(Z && scale && shift && Y && idx && B > 0 && Ho > 0 && Wo > 0, "cy_affine_act_maxpool2: bad arguments");
  CY_REQUIRE(C > 0 && C % 4 == 0 && slope >= 0.f && slope <= 1.f, "cy_affine_act_maxpool2: C=%d must be a multiple of 4, slope=%g in [0, 1]", C, (double)slope);
  CY_REQUIRE((((uintptr_t)Z | (uintptr_t)Y | (uintptr_t)idx) & 15) == 0, "cy_affine_act_maxpool2: operands must be 16-byte aligned");
  const long long n4 = (long long)B * Ho * Wo * (C / 4);
  affine_act_maxpool2_kernel<<<stream_grid(n4), 256, 0, (hipStream_t)stream>>>(Z, scale, shift, slope, Y, idx, n4, Ho, Wo, C);
  CY_LAUNCH_CHECK("cy_affine_act_maxpool2");
  return 0;
}

extern "C" int cy_maxpool2_bwd_bn(const float* dP, const unsigned char* idx, const float* Z, const float* scale, const float* shift,
                                  const float* mean, const float* invstd, float slope, float* D, double* red, int B, int Ho, int Wo,
                                  int C, void* stream) {
  CY_REQUIRE(dP && idx && Z && scale && shift && mean && invstd && D && red && B > 0 && Ho > 0 && Wo > 0, "cy_maxpool2_bwd_bn: bad arguments");
  CY_REQUIRE(C > 0 && C % 4 == 0 && (C < 64 ? 256 % (C / 4) == 0 : C % 64 == 0) && slope >= 0.f && slope <= 1.f,
             "cy_maxpool2_bwd_bn: C=%d must be a multiple of 64 (or 4, 8, 16, 32), slope=%g in [0, 1]", C, (double)slope);
  CY_REQUIRE((((uintptr_t)dP | (uintptr_t)idx | (uintptr_t)Z | (uintptr_t)D) & 15) == 0, "cy_maxpool2_bwd_bn: operands must be 16-byte aligned");
  const int CW = C < 64 ? C : 64, nsl = C / CW, rpar = 256 / (CW / 4);
  const long long Pp = (long long)B * Ho * Wo;
  long long rows_per_block = cy_ceil_div(Pp, cy_ceil_div(2048, nsl));     // about 2048 blocks
  if (rows_per_block < 4 * rpar) rows_per_block = 4 * rpar;
  rows_per_block = cy_ceil_div(rows_per_block, rpar) * rpar;
  const long long blocks = cy_ceil_div(Pp, rows_per_block) * nsl;
  maxpool2_bwd_bn_kernel<<<(unsigned)blocks, 256, 256 * 8 * 8, (hipStream_t)stream>>>(dP, idx, Z, scale, shift, mean, invstd, slope, D, red, Pp,
                                                                                    Ho, Wo, C, rows_per_block);
  CY_LAUNCH_CHECK("cy_maxpool2_bwd_bn");
  return 0;
}

extern "C" int cy_bn_bwd_apply(const float* Z, const float* dA, float* dZ, const float* scale, const float* shift,
                               const float* mean, const float* invstd, const float* gamma, float slope,
                               const double* red, float* dgamma, float* dbeta, long long P, int N, void* stream) {
  CY_REQUIRE(Z && dA && dZ && scale && shift && mean && invstd && red && P > 0, "cy_bn_bwd_apply: bad arguments");
  CY_REQUIRE(N % 4 == 0, "cy_bn_bwd_apply: N=%d must be a multiple of 4", N);
  (void)gamma;
  hipStream_t s = (hipStream_t)stream;
  const long long n4 = P * N / 4;
  if (N <= 1024 && (1024 % N) == 0)
    bn_bwd_apply_kernel<true><<<stream_grid(n4), 256, 0, s>>>(Z, dA, dZ, scale, shift, mean, invstd, slope, red,
                                                              1.0 / (double)P, n4, N);
  else
    bn_bwd_apply_kernel<false><<<stream_grid(n4), 256, 0, s>>>(Z, dA, dZ, scale, shift, mean, invstd, slope, red,
                                                               1.0 / (double)P, n4, N);
  CY_LAUNCH_CHECK("cy_bn_bwd_apply");
  if (dgamma && dbeta) {
    bn_param_grad_kernel<<<(N + 255) / 256, 256, 0, s>>>(red, dgamma, dbeta, N);
    CY_LAUNCH_CHECK("cy_bn_bwd_apply(param)");
  }
  return 0;
}

extern "C" int cy_bn_param_grad(const double* red, float* dgamma, float* dbeta, int N, void* stream) {
  CY_REQUIRE(red && dgamma && dbeta && N > 0, "cy_bn_param_grad: bad arguments");
  bn_param_grad_kernel<<<(N + 255) / 256, 256, 0, (hipStream_t)stream>>>(red, dgamma, dbeta, N);
  CY_LAUNCH_CHECK("cy_bn_param_grad");
  return 0;
}

extern "C" int cy_act_bwd(const float* Z, const float* dA, float* dZ, float slope, long long n, void* stream) {
  CY_REQUIRE(Z && dA && dZ && n > 0, "cy_act_bwd: bad arguments");
  act_bwd_kernel<<<stream_grid(n), 256, 0, (hipStream_t)stream>>>(Z, dA, dZ, slope, n);
  CY_LAUNCH_CHECK("cy_act_bwd");
  return 0;
}
