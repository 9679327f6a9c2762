// Capsule dynamic routing, all iterations in one launch, forward and backward (gfx950).
//
// Replaces CapsuleLayer.forward's routing branch (models.py:70-79) and its autograd backward.
//   u_hat_ij = u_i W_ij ;  b^t_ij = u_hat_ij . V_t[j],  V_t = sum_{tau<t} v^tau   (the reference
//   accumulates logits b += u_hat.v, which is the same sum -- logits are never stored);
//   c^t = softmax_j(b^t) ; s^t_j = sum_i c^t_ij u_hat_ij ; v^t = squash(s^t).
// u_hat is recomputed every iteration from (u, W) instead of being stored (the reference keeps a
// [R,N,C,1,Dout] tensor and makes ~20 passes over it).
//
// Two code paths:
//  * C == 1 (DarkCapsuleNet head, models.py:368-370): coupling == 1 exactly, the layer is
//    v = squash(sum_i u_i W_i): a pure HBM stream over u (R rows of 4096 floats).  One block of 4
//    waves walks rows; wave w owns quarter w of the row, whose 16 elements per lane meet W values
//    held in registers.  With the NHWC feature map the "cell gather" (models.py:393-398) makes each
//    quarter one contiguous 4 KiB segment, so the gather costs nothing.
//  * general C <= 64: lanes <-> output capsule j, one wave per row.  The softmax over j is a
//    wavefront reduction, s_j / V_j / the running sums are lane-local registers, W_i is staged
//    once per block through LDS and shared by the block's rows.
// Backward (general): B1 (same row decomposition) walks t = T-1..1 producing ds^t and V_t per
// row; B2 (one wave per input capsule i, lanes <-> j) turns them into du and dW with the dW_i
// tile accumulated in registers across all rows.
#include "common.h"
#include <stdlib.h>

namespace {

// ------------------------------------------------------------------------------------------------ helpers
__device__ __forceinline__ void squash_vec(const float* s, float* v, int D) {
  float n2 = 0.f;
  for (int o = 0; o < D; ++o) n2 += s[o] * s[o];
  const float f = (n2 / (1.f + n2)) / sqrtf(n2);          // no epsilon: 0 -> NaN like the reference
  for (int o = 0; o < D; ++o) v[o] = f * s[o];
}
// ds = J^T dv for v = s * n/(1+n^2), n = |s|
__device__ __forceinline__ void squash_bwd_vec(const float* s, const float* dv, float* ds, int D) {
  float n2 = 0.f, sd = 0.f;
  for (int o = 0; o < D; ++o) { n2 += s[o] * s[o]; sd += s[o] * dv[o]; }
  const float n = sqrtf(n2);
  const float h = n / (1.f + n2);
  const float hp = (1.f - n2) / ((1.f + n2) * (1.f + n2));
  const float k = sd * hp / n;
  for (int o = 0; o < D; ++o) ds[o] = h * dv[o] + k * s[o];
}

// offset (in floats) of the Din-vector of input capsule i of row `row`
__device__ __forceinline__ long long u_offset(int row, int i, int N, int Din, int g, int B) {
  if (g == 0) return ((long long)row * N + i) * Din;
  const int k = row / B, b = row - k * B;
  const int pos = i >> 5, chg = i & 31;
  const long long pix = (long long)b * 16 * g * g + (long long)(pos >> 2) * 4 * g * g + 4 * k + (pos & 3);
  return pix * 256 + chg * 8;
}
// v_out / dv row index: with the cell gather the result is laid out [B][g*g] (what
// view(g,g,B,.).permute(2,0,1,3) of models.py:399 presents), otherwise row order
__device__ __forceinline__ long long out_row(int row, int g, int B) {
  if (g == 0) return row;
  const int k = row / B, b = row - k * B;
  return (long long)b * g * g + k;
}
// offset of quarter w (1024 floats) of row `row` when N*Din == 4096 (GATHER chosen at compile time so that the
// streaming loops stay straight-line code)
template <bool GATHER>
__device__ __forceinline__ long long quarter_offset(int row, int w, int g, int B) {
  if (!GATHER) return (long long)row * 4096 + w * 1024;
  const unsigned k = (unsigned)row / (unsigned)B, b = (unsigned)row - k * (unsigned)B;
  return ((long long)b * 16 * g * g + (long long)w * 4 * g * g + 4 * k) * 256;
}
template <bool GATHER>
__device__ __forceinline__ long long out_row_t(int row, int g, int B) {
  if (!GATHER) return row;
  const unsigned k = (unsigned)row / (unsigned)B, b = (unsigned)row - k * (unsigned)B;
  return (long long)b * g * g + k;
}

// ================================================================================================ C == 1
// One persistent block of 8 waves per CU; block b owns a contiguous range of rows.  Wave (q = w & 3, s = w >> 2):
// quarter q of the rows lo+s, lo+s+2, ...  The wave's 80 W values are loaded straight into registers (20
// float4 per lane, 5 KiB contiguous per wave-instruction).  There is NO block barrier in the row loop: a wave
// reduces its partial dot products to 16-lane row sums with DPP, lanes 15/31/47/63 drop them into the row's LDS
// slot, and the wave whose LDS ticket is the fourth of that row finishes it (sum, squash, store).
constexpr int C1_THREADS = 512;            // 8 waves: 4 quarters x 2 row slots
constexpr int C1_SLOTS = C1_THREADS / 256;
constexpr int C1_MAXROWS = 64;             // rows per block the LDS slot table can hold

template <int DOUT>
__device__ __forceinline__ void c1_load_w(const float* __restrict__ W, float (&wr)[4][4][DOUT], int q, int lane) {
  // element e = q*1024 + j*256 + lane*4 + qq  -> W[e][o] at W[e*DOUT + o]; 4 consecutive e = 4*DOUT floats
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const float* src = W + (size_t)(q * 1024 + j * 256 + lane * 4) * DOUT;
    float tmp[4 * DOUT];
#pragma unroll
    for (int k = 0; k < DOUT; ++k) {
      const f32x4 v = *(const f32x4*)(src + 4 * k);
      tmp[4 * k] = v[0]; tmp[4 * k + 1] = v[1]; tmp[4 * k + 2] = v[2]; tmp[4 * k + 3] = v[3];
    }
#pragma unroll
    for (int qq = 0; qq < 4; ++qq)
#pragma unroll
      for (int o = 0; o < DOUT; ++o) wr[j][qq][o] = tmp[qq * DOUT + o];
  }
}

template <int DOUT, bool GATHER>
__global__ __launch_bounds__(C1_THREADS, 2) void caps1_fwd_kernel(const float* __restrict__ u, const float* __restrict__ W,
                                                                  float* __restrict__ v_out, float* __restrict__ s_out,
                                                                  int R, int g, int B) {
  __shared__ float part[C1_MAXROWS][16][8];       // [row of the block][4 quarters x 4 lane-rows][DOUT (padded)]
  const int t = threadIdx.x, lane = t & 63;
  const int w = __builtin_amdgcn_readfirstlane(t >> 6);
  const int q = w & 3, sl = w >> 2;
  const int lo = (int)((long long)R * blockIdx.x / gridDim.x), hi = (int)((long long)R * (blockIdx.x + 1) / gridDim.x);
  // The row loop is straight-line VMEM: loads are unconditional (rows past the end are clamped to the block's
  // last row and their results dropped) and nothing is stored to global memory inside it, so the compiler's
  // s_waitcnt vmcnt(N) stays exact and the next row set remains in flight while this one is consumed.
  f32x4 xa[4], xb[4];
#define C1_LOAD_ROW(X, ROW)                                                                        \
  {                                                                                                \
    const int r_ = (ROW) < hi ? (ROW) : hi - 1;                                                    \
    const f32x4* src_ = (const f32x4*)(u + quarter_offset<GATHER>(r_, q, g, B)) + lane;                    \
    _Pragma("unroll") for (int j = 0; j < 4; ++j) X[j] = src_[j * 64];                             \
  }
#define C1_CONSUME(X, ROW)                                                                         \
  {                                                                                                \
    float acc[DOUT];                                                                               \
    _Pragma("unroll") for (int o = 0; o < DOUT; ++o) acc[o] = 0.f;                                 \
    _Pragma("unroll") for (int j = 0; j < 4; ++j)                                                  \
      _Pragma("unroll") for (int qq = 0; qq < 4; ++qq)                                             \
        _Pragma("unroll") for (int o = 0; o < DOUT; ++o) acc[o] += X[j][qq] * wr[j][qq][o];        \
    _Pragma("unroll") for (int o = 0; o < DOUT; ++o) acc[o] = row16_sum(acc[o]);                   \
    if ((lane & 15) == 15 && (ROW) < hi) {                                                         \
      float* dst_ = &part[(ROW) - lo][q * 4 + (lane >> 4)][0];                                     \
      _Pragma("unroll") for (int o = 0; o < DOUT; ++o) dst_[o] = acc[o];                           \
    }                                                                                              \
  }
  const int first = lo + sl;
  C1_LOAD_ROW(xa, first)
  C1_LOAD_ROW(xb, first + C1_SLOTS)
  float wr[4][4][DOUT];
  c1_load_w<DOUT>(W, wr, q, lane);
  // drain the prologue's loads here, once: otherwise the loop header inherits "W may be pending" from the
  // preheader and hipcc waits vmcnt(0) in every iteration, which would also drain the prefetched row set
  __builtin_amdgcn_s_waitcnt(0x0F70);       // vmcnt(0), expcnt/lgkmcnt untouched
  for (int row = first; row < hi; row += 2 * C1_SLOTS) {
    C1_CONSUME(xa, row)
    C1_LOAD_ROW(xa, row + 2 * C1_SLOTS)
    C1_CONSUME(xb, row + C1_SLOTS)
    C1_LOAD_ROW(xb, row + 3 * C1_SLOTS)
  }
#undef C1_LOAD_ROW
#undef C1_CONSUME
  __syncthreads();
  // finish: thread -> (row = t/8, o = t%8): sum the 16 partial sums, squash over the row's DOUT values
  {
    const int rr = t >> 3, o = t & 7;
    const int row = lo + rr;
    if (rr < C1_MAXROWS) {
      float sv = 0.f;
      if (row < hi && o < DOUT) {
#pragma unroll
        for (int k = 0; k < 16; ++k) sv += part[rr][k][o];
      }
      float n2 = sv * sv;                   // lanes o >= DOUT contribute 0
      n2 += dpp_get<0xB1, 0xF>(0.f, n2);    // 8-lane groups: quad swaps + half-row mirror
      n2 += dpp_get<0x4E, 0xF>(0.f, n2);
      n2 += dpp_get<0x141, 0xF>(0.f, n2);
      if (row < hi && o < DOUT) {
        const float f = (n2 / (1.f + n2)) / sqrtf(n2);
        s_out[(long long)row * DOUT + o] = sv;
        v_out[out_row_t<GATHER>(row, g, B) * DOUT + o] = f * sv;
      }
    }
  }
}

constexpr int C1B_THREADS = 512;
// backward: du = W ds (written in place of the gather), dW = sum_rows u (x) ds accumulated in 80 registers per
// lane; the two slots of a quarter are combined through LDS and each block writes ONE 4096*DOUT slab.
template <int DOUT, bool GATHER>
__global__ __launch_bounds__(C1B_THREADS, 2) void caps1_bwd_kernel(const float* __restrict__ u, const float* __restrict__ W,
                                                                  const float* __restrict__ s_in, const float* __restrict__ dv,
                                                                  float* __restrict__ du, float* __restrict__ slabs, int R,
                                                                  int g, int B) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int t = threadIdx.x, lane = t & 63;
  const int w = __builtin_amdgcn_readfirstlane(t >> 6);
  const int q = w & 3, sl = w >> 2;
  const int lo = (int)((long long)R * blockIdx.x / gridDim.x), hi = (int)((long long)R * (blockIdx.x + 1) / gridDim.x);
  float wr[4][4][DOUT], dw[4][4][DOUT];
#pragma unroll
  for (int j = 0; j < 4; ++j)
#pragma unroll
    for (int qq = 0; qq < 4; ++qq)
#pragma unroll
      for (int o = 0; o < DOUT; ++o) dw[j][qq][o] = 0.f;

  // this wave's rows: first, first+2, ... (n of them).  Same straight-line discipline as the forward kernel
  // (unconditional clamped loads, every store belongs to a valid row, W drained once before the loop), but with
  // ONE register set (wr + dw already take 160 VGPRs): as soon as the FMAs of float4 j are done, the same
  // registers are re-loaded with float4 j of the wave's next row, so each load has 3/4 of a row's work to land.
  const int first = lo + sl;
  const int n = first < hi ? (hi - first + 1) / 2 : 0;
  f32x4 x[4];
  {
    const f32x4* src = (const f32x4*)(u + quarter_offset<GATHER>(first < R ? first : R - 1, q, g, B)) + lane;
#pragma unroll
    for (int j = 0; j < 4; ++j) x[j] = src[j * 64];
  }
  c1_load_w<DOUT>(W, wr, q, lane);
  __builtin_amdgcn_s_waitcnt(0x0F70);       // vmcnt(0): see caps1_fwd_kernel
  for (int i = 0; i < n; ++i) {
    const int row = first + 2 * i;
    const int nrow = (i + 1 < n) ? row + 2 : row;           // clamped: the last row is simply read again
    float sv[DOUT], dvv[DOUT], ds[DOUT];
#pragma unroll
    for (int o = 0; o < DOUT; ++o) {
      sv[o] = s_in[(long long)row * DOUT + o];
      dvv[o] = dv[out_row_t<GATHER>(row, g, B) * DOUT + o];
    }
    squash_bwd_vec(sv, dvv, ds, DOUT);
    f32x4* dst = (f32x4*)(du + quarter_offset<GATHER>(row, q, g, B)) + lane;
    const f32x4* nsrc = (const f32x4*)(u + quarter_offset<GATHER>(nrow, q, g, B)) + lane;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      f32x4 gq;
#pragma unroll
      for (int qq = 0; qq < 4; ++qq) {
        float a0 = 0.f;
#pragma unroll
        for (int o = 0; o < DOUT; ++o) {
          a0 += wr[j][qq][o] * ds[o];
          dw[j][qq][o] += x[j][qq] * ds[o];
        }
        gq[qq] = a0;
      }
      x[j] = nsrc[j * 64];
      dst[j * 64] = gq;
    }
  }
  // slot 1 -> LDS, slot 0 adds and writes the block's slab (element-major: e*DOUT + o, like W)
  __syncthreads();
  if (sl == 1) {
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int qq = 0; qq < 4; ++qq)
#pragma unroll
        for (int o = 0; o < DOUT; ++o) lds[(size_t)(q * 1024 + j * 256 + lane * 4 + qq) * DOUT + o] = dw[j][qq][o];
  }
  __syncthreads();
  if (sl == 0) {
    float* slab = slabs + (long long)blockIdx.x * 4096 * DOUT;
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int qq = 0; qq < 4; ++qq)
#pragma unroll
        for (int o = 0; o < DOUT; ++o) {
          const size_t idx = (size_t)(q * 1024 + j * 256 + lane * 4 + qq) * DOUT + o;
          slab[idx] = dw[j][qq][o] + lds[idx];
        }
  }
}

// out[i] = sum_k slabs[k][i]: block = 64 outputs x 16 slab phases (1024 threads), coalesced 256-byte rows,
// 4 independent loads in flight per thread
__global__ __launch_bounds__(1024) void slab_sum_kernel(const float* __restrict__ slabs, float* __restrict__ out, int nslabs,
                                                        long long n) {
  __shared__ float red[16][64];
  const int lane = threadIdx.x & 63, ph = threadIdx.x >> 6;
  const long long i = (long long)blockIdx.x * 64 + lane;
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  if (i < n) {
    int k = ph;
    for (; k + 48 < nslabs; k += 64) {
      s0 += slabs[(long long)k * n + i];
      s1 += slabs[(long long)(k + 16) * n + i];
      s2 += slabs[(long long)(k + 32) * n + i];
      s3 += slabs[(long long)(k + 48) * n + i];
    }
    for (; k < nslabs; k += 16) s0 += slabs[(long long)k * n + i];
  }
  red[ph][lane] = (s0 + s1) + (s2 + s3);
  __syncthreads();
  if (ph == 0 && i < n) {
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < 16; ++k) s += red[k][lane];
    out[i] = s;
  }
}

// ================================================================================================ general C
// Forward and the row part of the backward run on the row-stationary pass of routing_rows.hip, du / dW on the
// input-capsule-stationary kernel of routing_caps.hip.

// ------------------------------------------------------------------------------------------------ small vector ops
__global__ void squash_fwd_kernel(const float* __restrict__ s, float* __restrict__ v, long long rows, int D) {
  const long long r = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= rows) return;
  float n2 = 0.f;
  for (int o = 0; o < D; ++o) { const float x = s[r * D + o]; n2 += x * x; }
  const float f = (n2 / (1.f + n2)) / sqrtf(n2);
  for (int o = 0; o < D; ++o) v[r * D + o] = f * s[r * D + o];
}
__global__ void squash_bwd_kernel(const float* __restrict__ s, const float* __restrict__ dv, float* __restrict__ ds,
                                  long long rows, int D) {
  const long long r = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= rows) return;
  float n2 = 0.f, sd = 0.f;
  for (int o = 0; o < D; ++o) { const float x = s[r * D + o]; n2 += x * x; sd += x * dv[r * D + o]; }
  const float n = sqrtf(n2), h = n / (1.f + n2), hp = (1.f - n2) / ((1.f + n2) * (1.f + n2));
  const float k = sd * hp / n;
  for (int o = 0; o < D; ++o) ds[r * D + o] = h * dv[r * D + o] + k * s[r * D + o];
}
__global__ void length_fwd_kernel(const float* __restrict__ v, float* __restrict__ len, long long rows, int D) {
  const long long r = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= rows) return;
  float n2 = 0.f;
  for (int o = 0; o < D; ++o) { const float x = v[r * D + o]; n2 += x * x; }
  len[r] = sqrtf(n2);
}
__global__ void length_bwd_kernel(const float* __restrict__ v, const float* __restrict__ len,
                                  const float* __restrict__ dlen, float* __restrict__ dv, long long rows, int D) {
  const long long r = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= rows) return;
  const float k = dlen[r] / len[r];          // 0/0 -> NaN at a zero capsule, like (x**2).sum()**0.5
  for (int o = 0; o < D; ++o) dv[r * D + o] = k * v[r * D + o];
}

// ================================================================================================ small R: phased
// With few rows the row decomposition cannot fill the chip (R = 32 is one row tile), so the input capsules are split
// over blocks as well: grid = (row tiles) x (chunks of i).  The sum over i then crosses blocks, which is a grid-wide
// dependency once per routing iteration; on this chip a kernel boundary (~1.7 us) is cheaper than an in-kernel grid
// barrier (5-7 us), so each iteration is one pass launch (routing_rows.hip) writing per-chunk partial sums plus ONE
// finish launch (slab_fin_kernel: sum over chunks, squash, V += v).
// slab_sum_kernel and the finish of an iteration in ONE launch (few rows: every launch boundary of the six-launch head costs ~2-5 us
// next to 25 us of work): block = CPB = 64 / DOUT whole capsules x 16 slab phases; the partial sums are added in slab_sum_kernel's
// order and the finish calls the same squash helpers on the gathered capsule vector, so the results equal the two-launch path's bit
// for bit.  BWD = false: s_hist[it] = sum, v = squash(s), V (+)= v, last: v_out.  BWD = true: A_t = sum, SA += A_t,
// ds_all[t - 1] = squash_bwd(s^{t-1}, SA).
template <int DOUT, bool BWD>
__global__ __launch_bounds__(1024) void slab_fin_kernel(const float* __restrict__ slabs, int nslabs, long long plane,
                                                        float* __restrict__ s_hist_it, float* __restrict__ V, float* __restrict__ v_out,
                                                        const float* __restrict__ s_prev, float* __restrict__ ds_prev, float* __restrict__ SA,
                                                        int C, int it, int last, int g, int B) {
  constexpr int CPB = 64 / DOUT, OPB = CPB * DOUT;
  __shared__ float red[16][64];
  __shared__ float sh[64], sh2[64];
  const int lane = threadIdx.x & 63, ph = threadIdx.x >> 6;
  const long long i = (long long)blockIdx.x * OPB + lane;
  const bool act = lane < OPB && i < plane;
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  if (act) {
    int k = ph;
    for (; k + 48 < nslabs; k += 64) {
      s0 += slabs[(long long)k * plane + i];
      s1 += slabs[(long long)(k + 16) * plane + i];
      s2 += slabs[(long long)(k + 32) * plane + i];
      s3 += slabs[(long long)(k + 48) * plane + i];
    }
    for (; k < nslabs; k += 16) s0 += slabs[(long long)k * plane + i];
  }
  red[ph][lane] = (s0 + s1) + (s2 + s3);
  __syncthreads();
  if (ph != 0) return;
  float sum = 0.f;
#pragma unroll
  for (int k = 0; k < 16; ++k) sum += red[k][lane];
  const int cl = lane / DOUT, o = lane - cl * DOUT;     // capsule of the block, component
  if constexpr (!BWD) {
    if (act) s_hist_it[i] = sum;
    sh[lane] = sum;
    __builtin_amdgcn_s_waitcnt(0xC07F);                 // (one wave: LDS visibility is program order + lgkmcnt)
    float sv[DOUT], vv[DOUT];
#pragma unroll
    for (int q = 0; q < DOUT; ++q) sv[q] = sh[(cl < CPB ? cl : 0) * DOUT + q];
    squash_vec(sv, vv, DOUT);
    if (act) {
      const float v = vv[o];
      V[i] = (it == 0 ? 0.f : V[i]) + v;
      if (last) {
        const long long cap = i / DOUT;
        const int row = (int)(cap / C), j = (int)(cap - (long long)row * C);
        v_out[(out_row(row, g, B) * C + j) * DOUT + o] = v;
      }
    }
  } else {
    float sa = 0.f, sp = 0.f;
    if (act) { sa = SA[i] + sum; SA[i] = sa; sp = s_prev[i]; }
    sh[lane] = sp; sh2[lane] = sa;
    __builtin_amdgcn_s_waitcnt(0xC07F);
    float sv[DOUT], dvv[DOUT], ds[DOUT];
#pragma unroll
    for (int q = 0; q < DOUT; ++q) { sv[q] = sh[(cl < CPB ? cl : 0) * DOUT + q]; dvv[q] = sh2[(cl < CPB ? cl : 0) * DOUT + q]; }
    squash_bwd_vec(sv, dvv, ds, DOUT);
    if (act) ds_prev[i] = ds[o];
  }
}

// backward preparation: V_all[t] = sum_{tau<t} squash(s^tau), ds_all[T-1] = squash_bwd(s^{T-1}, dv), SA = 0
template <int DOUT>
__global__ void routing_bwd_prep_kernel(const float* __restrict__ s_hist, const float* __restrict__ dv,
                                        float* __restrict__ ds_all, float* __restrict__ V_all, float* __restrict__ SA,
                                        int R, int C, int NT, int g, int B) {
  const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (long long)R * C) return;
  const int row = (int)(idx / C), j = (int)(idx - (long long)row * C);
  const long long plane = (long long)R * C * DOUT, my = idx * DOUT;
  float Vt[DOUT], sv[DOUT], vv[DOUT];
#pragma unroll
  for (int o = 0; o < DOUT; ++o) Vt[o] = 0.f;
  for (int t = 0; t < NT; ++t) {
#pragma unroll
    for (int o = 0; o < DOUT; ++o) { V_all[(long long)t * plane + my + o] = Vt[o]; sv[o] = s_hist[(long long)t * plane + my + o]; }
    if (t == NT - 1) {
      float d[DOUT], ds[DOUT];
      const float* dp = dv + (out_row(row, g, B) * C + j) * DOUT;
#pragma unroll
      for (int o = 0; o < DOUT; ++o) d[o] = dp[o];
      squash_bwd_vec(sv, d, ds, DOUT);
#pragma unroll
      for (int o = 0; o < DOUT; ++o) { ds_all[(long long)t * plane + my + o] = ds[o]; SA[my + o] = 0.f; }
    } else {
      squash_vec(sv, vv, DOUT);
#pragma unroll
      for (int o = 0; o < DOUT; ++o) Vt[o] += vv[o];
    }
  }
}

constexpr int C1_BLOCKS_DEFAULT = 256;     // one persistent block per CU
inline int c1_blocks() {                   // CY_C1_BLOCKS: developer knob (512 = two co-resident blocks per CU)
  static int v = [] { const char* e = getenv("CY_C1_BLOCKS"); const int k = e ? atoi(e) : 0; return (k >= 64 && k <= 1024) ? k : C1_BLOCKS_DEFAULT; }();
  return v;
}
#define C1_BLOCKS c1_blocks()
bool fast_c1(int N, int C, int Din, int Dout) { return C == 1 && N * Din == 4096 && Dout == 5; }

cyi_rows_args_t rows_args(const float* u, const float* Wp, int R, int N, int C, int n_iter, int g, int B) {
  cyi_rows_args_t r{};
  r.u = u; r.Wp = Wp; r.R = R; r.N = N; r.C = C; r.n_iter = n_iter; r.g = g; r.B = B;
  return r;
}
inline long long align4(long long n) { return (n + 3) & ~3ll; }

// CY_ROUTING_MFMA=1: the forward of C > 1 heads on routing_mfma.hip (u_hat on v_mfma_f32_16x16x4_f32) instead of the vector kernel
// of routing_rows.hip.  OFF by default: measured slower (round 3, tools/ab_routing_mfma.py: DarkCapsuleNet3 head 5.67 ms against
// 3.78 ms, CapsuleNet head 0.178 against 0.130 ms) -- with u_hat on the matrix cores the step is bound by what stays on the vector
// pipe next to them (logits, softmax, weighted sums, accumulator moves: 464 vector instructions per input capsule and wave against 36
// MFMAs) and the 16-row tiles need two rounds on 256 CUs; DESIGN section 4 has the census and what it would take.
inline bool mfma_enabled() {
  const char* e = getenv("CY_ROUTING_MFMA");
  return e && e[0] == '1';
}
inline long long fwd_wp_floats(int N, int C, int Dout) {          // the larger of the two packed W images
  const long long a = cyi_rows_wp_floats(N, C, Dout), b = cyi_mfma_ok(C, Dout) ? cyi_mfma_wp_floats(N, C, Dout) : 0;
  return a > b ? a : b;
}
// chunks of input capsules per row tile of the MFMA kernel's phased plan (few rows): about one block per CU, never more chunks
// than the vector plan has (the workspace holds p.nch slabs)
inline void mfma_chunks(int R, int N, const cyi_rows_plan_t& p, int* nch, int* ic) {
  const int row_tiles = (R + 15) / 16;
  int n = 256 / row_tiles;
  if (n > p.nch) n = p.nch;
  if (n > (N + 1) / 2) n = (N + 1) / 2;
  if (n < 1) n = 1;
  *ic = (N + n - 1) / n;
  *nch = (N + *ic - 1) / *ic;
}

// workspace of the forward: [packed W image][phased plans: V_t, partial-sum slabs]
template <int DOUT>
int launch_fwd(const cy_routing_fwd_t* a, hipStream_t s) {
  cyi_rows_plan_t p;
  cyi_rows_plan(a->R, a->N, a->C, DOUT, 0, &p);
  if (a->ws == nullptr) return cy_set_error(CY_EINVAL, "cy_routing_fwd: this shape needs the workspace (ws) of cy_routing_fwd_ws_floats()");
  float* Wp = a->ws;
  const bool mfma = mfma_enabled() && cyi_mfma_ok(a->C, DOUT);
  int rc = mfma ? cyi_mfma_pack_w(a->W, Wp, a->N, a->C, DOUT, s) : cyi_rows_pack_w(a->W, Wp, a->N, a->C, DOUT, s);
  if (rc) return rc;
  cyi_rows_args_t r = rows_args(a->u, Wp, a->R, a->N, a->C, a->n_iter, a->gather_g, a->gather_B);
  r.s_hist = a->s_hist; r.v_out = a->v_out;
  if (!p.phased) {
    r.fused = 1; r.ic = a->N;
    return mfma ? cyi_mfma_launch(&r, 1, DOUT, s) : cyi_rows_launch(0, &r, &p, DOUT, s);
  }
  const long long plane = (long long)a->R * a->C * DOUT;
  float* V = a->ws + fwd_wp_floats(a->N, a->C, DOUT);
  float* slab = V + align4(plane);
  int nch = p.nch, ic = p.ic;
  if (mfma) mfma_chunks(a->R, a->N, p, &nch, &ic);
  for (int it = 0; it < a->n_iter; ++it) {
    r.fused = 0; r.it = it; r.ic = ic; r.V = it > 0 ? V : nullptr; r.slab = slab;
    rc = mfma ? cyi_mfma_launch(&r, nch, DOUT, s) : cyi_rows_launch(0, &r, &p, DOUT, s);
    if (rc) return rc;
    constexpr int OPB = (64 / DOUT) * DOUT;
    slab_fin_kernel<DOUT, false><<<(unsigned)cy_ceil_div(plane, OPB), 1024, 0, s>>>(slab, nch, plane, a->s_hist + (long long)it * plane, V, a->v_out,
                                                                                     nullptr, nullptr, nullptr, a->C, it, it == a->n_iter - 1,
                                                                                     a->gather_g, a->gather_B);
  }
  return 0;
}
// workspace of the backward: [ds_all][V_all] (read by routing_caps.hip) [phased plans: SA, A_t, slabs] [+4] [packed W image]
inline long long bwd_ws_head(const cy_routing_bwd_t* a, const cyi_rows_plan_t& p) {
  const long long plane = (long long)a->R * a->C * a->Dout;
  return align4(2ll * a->n_iter * plane + (p.phased ? (2ll + p.nch) * plane : 0) + 4);   // + 4: routing_caps.hip reads whole 16-byte pieces
}
template <int DOUT>
int launch_bwd(const cy_routing_bwd_t* a, hipStream_t s) {
  cyi_rows_plan_t p;
  cyi_rows_plan(a->R, a->N, a->C, DOUT, 1, &p);
  const long long plane = (long long)a->R * a->C * DOUT;
  float* Wp = a->ws + bwd_ws_head(a, p);
  int rc = cyi_rows_pack_w(a->W, Wp, a->N, a->C, DOUT, s);
  if (rc) return rc;
  cyi_rows_args_t r = rows_args(a->u, Wp, a->R, a->N, a->C, a->n_iter, a->gather_g, a->gather_B);
  float* ds_all = a->ws;
  float* V_all = a->ws + (long long)a->n_iter * plane;
  float* cdb = nullptr;                             // fused plans (many rows): c^t, db^t of every (t >= 1, row, i, j), behind the W image
  if (!p.phased && a->n_iter > 1 && DOUT <= 21) cdb = Wp + cyi_rows_wp_floats(a->N, a->C, DOUT);   // (Dout = 48: dW alone overflows the registers)
  if (!p.phased) {
    r.fused = 1; r.ic = a->N; r.s_hist = const_cast<float*>(a->s_hist); r.dv = a->dv; r.ds_all = ds_all; r.V_all = V_all;
    r.cdb = cdb;
    rc = cyi_rows_launch(1, &r, &p, DOUT, s);
    if (rc) return rc;
  } else {
    float* SA = a->ws + 2ll * a->n_iter * plane;
    float* At = SA + plane;
    float* slab = At + plane;
    const int fin_blocks = (int)cy_ceil_div((long long)a->R * a->C, 128);
    routing_bwd_prep_kernel<DOUT><<<fin_blocks, 128, 0, s>>>(a->s_hist, a->dv, ds_all, V_all, SA, a->R, a->C, a->n_iter,
                                                             a->gather_g, a->gather_B);
    for (int t = a->n_iter - 1; t >= 1; --t) {
      r.fused = 0; r.it = t; r.ic = p.ic; r.V = V_all + (long long)t * plane; r.ds = ds_all + (long long)t * plane; r.slab = slab;
      rc = cyi_rows_launch(1, &r, &p, DOUT, s);
      if (rc) return rc;
      constexpr int OPB = (64 / DOUT) * DOUT;
      slab_fin_kernel<DOUT, true><<<(unsigned)cy_ceil_div(plane, OPB), 1024, 0, s>>>(slab, p.nch, plane, nullptr, nullptr, nullptr,
                                                                                      a->s_hist + (long long)(t - 1) * plane,
                                                                                      ds_all + (long long)(t - 1) * plane, SA, a->C, t, 0, 0, 1);
      (void)At;
    }
  }
  return cyi_caps_bwd_launch(a, cdb, s);
}

int check_shape(const char* fn, int R, int N, int C, int Din, int Dout, int n_iter, int g, int B) {
  if (R <= 0 || N <= 0 || C <= 0 || n_iter <= 0) return cy_set_error(CY_EINVAL, "%s: non-positive dimension", fn);
  if (g != 0 && (N != 512 || Din != 8 || B <= 0 || R != g * g * B))
    return cy_set_error(CY_EINVAL, "%s: cell gather needs N=512, Din=8, R=g*g*B (got N=%d Din=%d R=%d g=%d B=%d)", fn, N,
                        Din, R, g, B);
  if (fast_c1(N, C, Din, Dout)) return 0;
  if (Din != 8 || !(Dout == 5 || Dout == 16 || Dout == 21 || Dout == 48) || C > 64)
    return cy_set_error(CY_EINVAL, "%s: unsupported capsule shape C=%d Din=%d Dout=%d (built: Din=8, Dout in {5,16,21,48}, C<=64)",
                        fn, C, Din, Dout);
  return 0;
}

}  // namespace

extern "C" int cy_routing_fwd(const cy_routing_fwd_t* a, void* stream) {
  CY_REQUIRE(a && a->u && a->W && a->v_out && a->s_hist, "cy_routing_fwd: null pointer");
  int rc = check_shape("cy_routing_fwd", a->R, a->N, a->C, a->Din, a->Dout, a->n_iter, a->gather_g, a->gather_B);
  if (rc) return rc;
  hipStream_t s = (hipStream_t)stream;
  if (fast_c1(a->N, a->C, a->Din, a->Dout)) {
    int blocks = a->R < C1_BLOCKS ? a->R : C1_BLOCKS;
    while ((a->R + blocks - 1) / blocks + 1 > C1_MAXROWS) blocks *= 2;   // keep rows per block within the slot table
    float* s_last = a->s_hist + (long long)(a->n_iter - 1) * a->R * 5;
    if (a->gather_g)
      caps1_fwd_kernel<5, true><<<blocks, C1_THREADS, 0, s>>>(a->u, a->W, a->v_out, s_last, a->R, a->gather_g, a->gather_B);
    else
      caps1_fwd_kernel<5, false><<<blocks, C1_THREADS, 0, s>>>(a->u, a->W, a->v_out, s_last, a->R, 0, 1);
  } else if (a->Dout == 5) rc = launch_fwd<5>(a, s);
  else if (a->Dout == 16) rc = launch_fwd<16>(a, s);
  else if (a->Dout == 21) rc = launch_fwd<21>(a, s);
  else rc = launch_fwd<48>(a, s);
  if (rc) return rc;
  CY_LAUNCH_CHECK("cy_routing_fwd");
  return 0;
}

extern "C" long long cy_routing_fwd_ws_floats(const cy_routing_fwd_t* a) {
  if (!a || fast_c1(a->N, a->C, a->Din, a->Dout)) return 0;
  cyi_rows_plan_t p;
  cyi_rows_plan(a->R, a->N, a->C, a->Dout, 0, &p);
  const long long plane = (long long)a->R * a->C * a->Dout;
  return fwd_wp_floats(a->N, a->C, a->Dout) + (p.phased ? align4(plane) + p.nch * plane : 0);
}

extern "C" long long cy_routing_bwd_ws_floats(const cy_routing_bwd_t* a) {
  if (!a) return 0;
  if (fast_c1(a->N, a->C, a->Din, a->Dout)) return (long long)C1_BLOCKS * 4096 * 5;
  cyi_rows_plan_t p;
  cyi_rows_plan(a->R, a->N, a->C, a->Dout, 1, &p);
  const long long cdb = (!p.phased && a->n_iter > 1 && a->Dout <= 21) ? 2ll * (a->n_iter - 1) * a->R * a->N * a->C : 0;
  return bwd_ws_head(a, p) + cyi_rows_wp_floats(a->N, a->C, a->Dout) + cdb;
}

extern "C" int cy_routing_bwd(const cy_routing_bwd_t* a, void* stream) {
  CY_REQUIRE(a && a->u && a->W && a->s_hist && a->dv && a->du && a->dW && a->ws, "cy_routing_bwd: null pointer");
  int rc = check_shape("cy_routing_bwd", a->R, a->N, a->C, a->Din, a->Dout, a->n_iter, a->gather_g, a->gather_B);
  if (rc) return rc;
  hipStream_t s = (hipStream_t)stream;
  if (fast_c1(a->N, a->C, a->Din, a->Dout)) {
    const int blocks = a->R < C1_BLOCKS ? a->R : C1_BLOCKS;
    const size_t lds = (size_t)4096 * 5 * 4;
    const float* s_last = a->s_hist + (long long)(a->n_iter - 1) * a->R * 5;
    if (a->gather_g) {
      rc = cy_allow_lds(caps1_bwd_kernel<5, true>, lds);
      if (rc) return rc;
      caps1_bwd_kernel<5, true><<<blocks, C1B_THREADS, lds, s>>>(a->u, a->W, s_last, a->dv, a->du, a->ws, a->R, a->gather_g, a->gather_B);
    } else {
      rc = cy_allow_lds(caps1_bwd_kernel<5, false>, lds);
      if (rc) return rc;
      caps1_bwd_kernel<5, false><<<blocks, C1B_THREADS, lds, s>>>(a->u, a->W, s_last, a->dv, a->du, a->ws, a->R, 0, 1);
    }
    CY_LAUNCH_CHECK("cy_routing_bwd(c1)");
    const long long n = 4096ll * 5;
    slab_sum_kernel<<<(unsigned)cy_ceil_div(n, 64), 1024, 0, s>>>(a->ws, a->dW, blocks, n);
    CY_LAUNCH_CHECK("cy_routing_bwd(c1 reduce)");
    return 0;
  }
  hipError_t e = hipMemsetAsync(a->dW, 0, (size_t)a->N * a->C * a->Din * a->Dout * 4, s);
  if (e != hipSuccess) return cy_set_error((int)e, "cy_routing_bwd: memset: %s", hipGetErrorString(e));
  if (a->Dout == 5) rc = launch_bwd<5>(a, s);
  else if (a->Dout == 16) rc = launch_bwd<16>(a, s);
  else if (a->Dout == 21) rc = launch_bwd<21>(a, s);
  else rc = launch_bwd<48>(a, s);
  if (rc) return rc;
  CY_LAUNCH_CHECK("cy_routing_bwd");
  return 0;
}

#define CY_ROWS_LAUNCH(kernel, ...)                                                              \
  kernel<<<(unsigned)cy_ceil_div(rows, 256), 256, 0, (hipStream_t)stream>>>(__VA_ARGS__, rows, D)

extern "C" int cy_squash_fwd(const float* s, float* v, long long rows, int D, void* stream) {
  CY_REQUIRE(s && v && rows > 0 && D > 0, "cy_squash_fwd: bad arguments");
  CY_ROWS_LAUNCH(squash_fwd_kernel, s, v);
  CY_LAUNCH_CHECK("cy_squash_fwd");
  return 0;
}
extern "C" int cy_squash_bwd(const float* s, const float* dv, float* ds, long long rows, int D, void* stream) {
  CY_REQUIRE(s && dv && ds && rows > 0 && D > 0, "cy_squash_bwd: bad arguments");
  CY_ROWS_LAUNCH(squash_bwd_kernel, s, dv, ds);
  CY_LAUNCH_CHECK("cy_squash_bwd");
  return 0;
}
extern "C" int cy_length_fwd(const float* v, float* len, long long rows, int D, void* stream) {
  CY_REQUIRE(v && len && rows > 0 && D > 0, "cy_length_fwd: bad arguments");
  CY_ROWS_LAUNCH(length_fwd_kernel, v, len);
  CY_LAUNCH_CHECK("cy_length_fwd");
  return 0;
}
extern "C" int cy_length_bwd(const float* v, const float* len, const float* dlen, float* dv, long long rows, int D,
                             void* stream) {
  CY_REQUIRE(v && len && dlen && dv && rows > 0 && D > 0, "cy_length_bwd: bad arguments");
  CY_ROWS_LAUNCH(length_bwd_kernel, v, len, dlen, dv);
  CY_LAUNCH_CHECK("cy_length_bwd");
  return 0;
}
