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
template <int DIN, int DOUT>
struct WTile {
  static constexpr int DD = DIN * DOUT;
  static constexpr bool V4 = (DOUT % 4 == 0);
  static constexpr int WS = V4 ? DD + 4 : (DD | 1);        // LDS row stride (floats) per output capsule j
  static constexpr int MAXC = 64;
  static constexpr int NREG = V4 ? (MAXC * DD / 4 + 255) / 256 : (MAXC * DD + 255) / 256;
};

// cooperative global -> register -> LDS staging of one W_i tile ([C][DD] contiguous in global)
template <int DIN, int DOUT>
struct WStage {
  using T = WTile<DIN, DOUT>;
  float4 r4[T::V4 ? T::NREG : 1];
  float r1[T::V4 ? 1 : T::NREG];
  __device__ __forceinline__ void load(const float* __restrict__ Wi, int C, int t) {
    if (T::V4) {
      const int n4 = C * T::DD / 4;
#pragma unroll
      for (int k = 0; k < T::NREG; ++k) {
        const int idx = t + 256 * k;
        r4[k] = (idx < n4) ? ((const float4*)Wi)[idx] : make_float4(0.f, 0.f, 0.f, 0.f);
      }
    } else {
      const int n = C * T::DD;
#pragma unroll
      for (int k = 0; k < T::NREG; ++k) { const int idx = t + 256 * k; r1[k] = (idx < n) ? Wi[idx] : 0.f; }
    }
  }
  __device__ __forceinline__ void store(float* __restrict__ Wl, int C, int t) const {
    if (T::V4) {
      const int n4 = C * T::DD / 4;
#pragma unroll
      for (int k = 0; k < T::NREG; ++k) {
        const int idx = t + 256 * k;
        if (idx < n4) { const int j = (idx * 4) / T::DD, rem = (idx * 4) % T::DD; *(float4*)(Wl + j * T::WS + rem) = r4[k]; }
      }
    } else {
      const int n = C * T::DD;
#pragma unroll
      for (int k = 0; k < T::NREG; ++k) {
        const int idx = t + 256 * k;
        if (idx < n) { const int j = idx / T::DD, rem = idx % T::DD; Wl[j * T::WS + rem] = r1[k]; }
      }
    }
  }
};

// u_hat[rr][o] = sum_d u[rr][d] * W_l[d*DOUT + o] for the wave's rows (W row read once for all rows)
template <int DIN, int DOUT, int RW>
__device__ __forceinline__ void predict(const float* __restrict__ Wl, const float (&uv)[RW][DIN], float (&uh)[RW][DOUT]) {
#pragma unroll
  for (int rr = 0; rr < RW; ++rr)
#pragma unroll
    for (int o = 0; o < DOUT; ++o) uh[rr][o] = 0.f;
#pragma unroll
  for (int d = 0; d < DIN; ++d) {
    float wrow[DOUT];
    if (DOUT % 4 == 0) {
#pragma unroll
      for (int o4 = 0; o4 < DOUT / 4; ++o4) {
        const float4 v = *(const float4*)(Wl + d * DOUT + o4 * 4);
        wrow[o4 * 4] = v.x; wrow[o4 * 4 + 1] = v.y; wrow[o4 * 4 + 2] = v.z; wrow[o4 * 4 + 3] = v.w;
      }
    } else {
#pragma unroll
      for (int o = 0; o < DOUT; ++o) wrow[o] = Wl[d * DOUT + o];
    }
#pragma unroll
    for (int rr = 0; rr < RW; ++rr)
#pragma unroll
      for (int o = 0; o < DOUT; ++o) uh[rr][o] += uv[rr][d] * wrow[o];
  }
}

template <int DIN, int DOUT, int RW>
__global__ __launch_bounds__(256) void routing_fwd_kernel(cy_routing_fwd_t a) {
  using T = WTile<DIN, DOUT>;
  extern __shared__ __attribute__((aligned(16))) float smem[];   // [2][C][WS]
  const int t = threadIdx.x, lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int C = a.C, N = a.N, R = a.R;
  const int tile = C * T::WS;
  const bool jv = lane < C;
  const int jl = jv ? lane : 0;
  const int row0 = (blockIdx.x * 4 + wave) * RW;
  const long long CD = (long long)C * DOUT;
  const float invC = 1.0f / (float)C;

  float V[RW][DOUT];
#pragma unroll
  for (int rr = 0; rr < RW; ++rr)
#pragma unroll
    for (int o = 0; o < DOUT; ++o) V[rr][o] = 0.f;

  WStage<DIN, DOUT> stage;
  for (int it = 0; it < a.n_iter; ++it) {
    float sacc[RW][DOUT];
#pragma unroll
    for (int rr = 0; rr < RW; ++rr)
#pragma unroll
      for (int o = 0; o < DOUT; ++o) sacc[rr][o] = 0.f;

    __syncthreads();                       // previous iteration's readers are done with both buffers
    stage.load(a.W, C, t);
    stage.store(smem, C, t);
    __syncthreads();
    for (int i = 0; i < N; ++i) {
      const int cur = i & 1;
      if (i + 1 < N) stage.load(a.W + (long long)(i + 1) * C * T::DD, C, t);
      const float* Wl = smem + cur * tile + jl * T::WS;
      float uv[RW][DIN], uh[RW][DOUT];
#pragma unroll
      for (int rr = 0; rr < RW; ++rr) {
        const int row = row0 + rr;
        if (row < R) {
          const float* up = a.u + u_offset(row, i, N, DIN, a.gather_g, a.gather_B);
#pragma unroll
          for (int d = 0; d < DIN; ++d) uv[rr][d] = up[d];
        } else {
#pragma unroll
          for (int d = 0; d < DIN; ++d) uv[rr][d] = 0.f;
        }
      }
      predict<DIN, DOUT, RW>(Wl, uv, uh);
#pragma unroll
      for (int rr = 0; rr < RW; ++rr) {
        float c = invC;
        if (it > 0) {
          float b = 0.f;
#pragma unroll
          for (int o = 0; o < DOUT; ++o) b += uh[rr][o] * V[rr][o];
          b = jv ? b : -INFINITY;
          const float m = wave_max(b);
          const float e = jv ? expf(b - m) : 0.f;
          c = e / wave_sum(e);
        }
#pragma unroll
        for (int o = 0; o < DOUT; ++o) sacc[rr][o] += c * uh[rr][o];
      }
      if (i + 1 < N) stage.store(smem + (cur ^ 1) * tile, C, t);
      __syncthreads();
    }
#pragma unroll
    for (int rr = 0; rr < RW; ++rr) {
      const int row = row0 + rr;
      if (row < R && jv) {
        float v[DOUT];
        squash_vec(sacc[rr], v, DOUT);
        float* sh = a.s_hist + ((long long)it * R + row) * CD + (long long)lane * DOUT;
#pragma unroll
        for (int o = 0; o < DOUT; ++o) { sh[o] = sacc[rr][o]; V[rr][o] += v[o]; }
        if (it == a.n_iter - 1) {
          float* vo = a.v_out + out_row(row, a.gather_g, a.gather_B) * CD + (long long)lane * DOUT;
#pragma unroll
          for (int o = 0; o < DOUT; ++o) vo[o] = v[o];
        }
      }
    }
  }
}

// ---- backward B1: per row, t = T-1..0 -> ds_all[t], V_all[t]   (ws = [2][T][R][C][DOUT])
template <int DIN, int DOUT>
__global__ __launch_bounds__(256) void routing_bwd_rows_kernel(cy_routing_bwd_t a) {
  using T = WTile<DIN, DOUT>;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int t = threadIdx.x, lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int C = a.C, N = a.N, R = a.R, NT = a.n_iter;
  const int tile = C * T::WS;
  const bool jv = lane < C;
  const int jl = jv ? lane : 0;
  const int row = blockIdx.x * 4 + wave;
  const bool rv = row < R;
  const long long CD = (long long)C * DOUT;
  const long long plane = (long long)R * CD;
  float* ds_all = a.ws;
  float* V_all = a.ws + (long long)NT * plane;
  const long long my = (long long)(rv ? row : 0) * CD + (long long)jl * DOUT;

  float SA[DOUT];
#pragma unroll
  for (int o = 0; o < DOUT; ++o) SA[o] = 0.f;
  WStage<DIN, DOUT> stage;

  for (int it = NT - 1; it >= 0; --it) {
    float Vt[DOUT], ds[DOUT];
#pragma unroll
    for (int o = 0; o < DOUT; ++o) { Vt[o] = 0.f; ds[o] = 0.f; }
    if (rv && jv) {
      for (int tau = 0; tau < it; ++tau) {
        float s[DOUT], v[DOUT];
#pragma unroll
        for (int o = 0; o < DOUT; ++o) s[o] = a.s_hist[(long long)tau * plane + my + o];
        squash_vec(s, v, DOUT);
#pragma unroll
        for (int o = 0; o < DOUT; ++o) Vt[o] += v[o];
      }
      float s[DOUT], dvv[DOUT];
#pragma unroll
      for (int o = 0; o < DOUT; ++o) {
        s[o] = a.s_hist[(long long)it * plane + my + o];
        dvv[o] = SA[o] + (it == NT - 1 ? a.dv[out_row(row, a.gather_g, a.gather_B) * CD + (long long)jl * DOUT + o] : 0.f);
      }
      squash_bwd_vec(s, dvv, ds, DOUT);
#pragma unroll
      for (int o = 0; o < DOUT; ++o) { ds_all[(long long)it * plane + my + o] = ds[o]; V_all[(long long)it * plane + my + o] = Vt[o]; }
    }
    if (it == 0) break;

    float A[DOUT];
#pragma unroll
    for (int o = 0; o < DOUT; ++o) A[o] = 0.f;
    __syncthreads();
    stage.load(a.W, C, t);
    stage.store(smem, C, t);
    __syncthreads();
    for (int i = 0; i < N; ++i) {
      const int cur = i & 1;
      if (i + 1 < N) stage.load(a.W + (long long)(i + 1) * C * T::DD, C, t);
      const float* Wl = smem + cur * tile + jl * T::WS;
      float uv[1][DIN], uh[1][DOUT];
      if (rv) {
        const float* up = a.u + u_offset(row, i, N, DIN, a.gather_g, a.gather_B);
#pragma unroll
        for (int d = 0; d < DIN; ++d) uv[0][d] = up[d];
      } else {
#pragma unroll
        for (int d = 0; d < DIN; ++d) uv[0][d] = 0.f;
      }
      predict<DIN, DOUT, 1>(Wl, uv, uh);
      float b = 0.f, dc = 0.f;
#pragma unroll
      for (int o = 0; o < DOUT; ++o) { b += uh[0][o] * Vt[o]; dc += uh[0][o] * ds[o]; }
      b = jv ? b : -INFINITY;
      const float m = wave_max(b);
      const float e = jv ? expf(b - m) : 0.f;
      const float c = e / wave_sum(e);
      const float dot = wave_sum(c * dc);
      const float db = c * (dc - dot);
#pragma unroll
      for (int o = 0; o < DOUT; ++o) A[o] += db * uh[0][o];
      if (i + 1 < N) stage.store(smem + (cur ^ 1) * tile, C, t);
      __syncthreads();
    }
#pragma unroll
    for (int o = 0; o < DOUT; ++o) SA[o] += A[o];
  }
}

// ---- backward B2: wave <-> input capsule i, lanes <-> j; dW_i accumulated in registers over the rows
constexpr int B2_WAVES = 4;                // input capsules (waves) per block; dW_i + u_hat need > 256 registers: one wave per SIMD
template <int DIN, int DOUT>
__global__ __launch_bounds__(64 * B2_WAVES, 1) void routing_bwd_caps_kernel(cy_routing_bwd_t a, int rows_per_chunk) {
  using T = WTile<DIN, DOUT>;
  extern __shared__ __attribute__((aligned(16))) float smem[];   // [B2_WAVES][C][WS]
  const int t = threadIdx.x, lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int C = a.C, N = a.N, R = a.R, NT = a.n_iter;
  const int i = blockIdx.x * B2_WAVES + wave;
  const bool iv = i < N;
  const bool jv = lane < C;
  const int jl = jv ? lane : 0;
  const long long CD = (long long)C * DOUT;
  const long long plane = (long long)R * CD;
  const float* ds_all = a.ws;
  const float* V_all = a.ws + (long long)NT * plane;
  const float invC = 1.0f / (float)C;
  float* Wme = smem + wave * C * T::WS;

  // this wave's W_i -> its private LDS region (wave-local, but a block barrier keeps it simple)
  if (iv) {
    const float* Wi = a.W + (long long)i * C * T::DD;
    if (T::V4) {                            // 8 independent float4 loads in flight per lane
      const int n4 = C * T::DD / 4;
      for (int base = 0; base < n4; base += 64 * 8) {
        f32x4 tmp[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          const int idx = base + k * 64 + lane;
          tmp[k] = idx < n4 ? ((const f32x4*)Wi)[idx] : f32x4{0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          const int idx = base + k * 64 + lane;
          if (idx < n4) *(f32x4*)(Wme + ((idx * 4) / T::DD) * T::WS + (idx * 4) % T::DD) = tmp[k];
        }
      }
    } else {
      for (int base = 0; base < C * T::DD; base += 64 * 8) {
        float tmp[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) { const int idx = base + k * 64 + lane; tmp[k] = idx < C * T::DD ? Wi[idx] : 0.f; }
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          const int idx = base + k * 64 + lane;
          if (idx < C * T::DD) Wme[(idx / T::DD) * T::WS + idx % T::DD] = tmp[k];
        }
      }
    }
  }
  __syncthreads();
  const float* Wl = Wme + jl * T::WS;

  float dw[DIN][DOUT];
#pragma unroll
  for (int d = 0; d < DIN; ++d)
#pragma unroll
    for (int o = 0; o < DOUT; ++o) dw[d][o] = 0.f;

  const int r0 = blockIdx.y * rows_per_chunk;
  const int r1 = min(R, r0 + rows_per_chunk);
  if (iv) {
    for (int row = r0; row < r1; ++row) {
      const long long uoff = u_offset(row, i, N, DIN, a.gather_g, a.gather_B);
      float uv[1][DIN], uh[1][DOUT], duh[DOUT];
#pragma unroll
      for (int d = 0; d < DIN; ++d) uv[0][d] = a.u[uoff + d];
      predict<DIN, DOUT, 1>(Wl, uv, uh);
      const long long my = (long long)row * CD + (long long)jl * DOUT;
#pragma unroll
      for (int o = 0; o < DOUT; ++o) duh[o] = jv ? invC * ds_all[my + o] : 0.f;
      for (int it = 1; it < NT; ++it) {
        float Vt[DOUT], ds[DOUT];
        if (DOUT % 4 == 0) {
#pragma unroll
          for (int o4 = 0; o4 < DOUT / 4; ++o4) {
            const f32x4 a4 = *(const f32x4*)(V_all + (long long)it * plane + my + o4 * 4);
            const f32x4 b4 = *(const f32x4*)(ds_all + (long long)it * plane + my + o4 * 4);
#pragma unroll
            for (int e = 0; e < 4; ++e) { Vt[o4 * 4 + e] = a4[e]; ds[o4 * 4 + e] = b4[e]; }
          }
        } else {
#pragma unroll
          for (int o = 0; o < DOUT; ++o) { Vt[o] = V_all[(long long)it * plane + my + o]; ds[o] = ds_all[(long long)it * plane + my + o]; }
        }
        float b = 0.f, dc = 0.f;
#pragma unroll
        for (int o = 0; o < DOUT; ++o) { b += uh[0][o] * Vt[o]; dc += uh[0][o] * ds[o]; }
        b = jv ? b : -INFINITY;
        const float m = wave_max(b);
        const float e = jv ? expf(b - m) : 0.f;
        const float c = e / wave_sum(e);
        const float dot = wave_sum(c * dc);
        const float db = c * (dc - dot);
#pragma unroll
        for (int o = 0; o < DOUT; ++o) duh[o] += jv ? (c * ds[o] + db * Vt[o]) : 0.f;
      }
      // du_i[d] = sum_j sum_o W[j][d][o] * duh_j[o]
      float mine = 0.f;
#pragma unroll
      for (int d = 0; d < DIN; ++d) {
        float p = 0.f;
#pragma unroll
        for (int o = 0; o < DOUT; ++o) {
          const float wv = Wl[d * DOUT + o];
          p += wv * duh[o];
          dw[d][o] += uv[0][d] * duh[o];
        }
        p = wave_sum(jv ? p : 0.f);
        if (lane == d) mine = p;
      }
      if (lane < DIN) a.du[uoff + lane] = mine;
    }
  }
  // dW_i: registers -> wave's LDS region -> coalesced atomics (row chunks add into the same tile)
  __syncthreads();
  if (iv) {
    if (jv) {
#pragma unroll
      for (int d = 0; d < DIN; ++d)
#pragma unroll
        for (int o = 0; o < DOUT; ++o) Wme[lane * T::WS + d * DOUT + o] = dw[d][o];
    }
  }
  __syncthreads();
  if (iv) {
    float* dWi = a.dW + (long long)i * C * T::DD;
    for (int idx = lane; idx < C * T::DD; idx += 64) atomicAdd(dWi + idx, Wme[(idx / T::DD) * T::WS + idx % T::DD]);
  }
}

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
// With few rows the row decomposition above cannot fill the chip (R=32 -> 4 blocks), so the input capsules are
// split into chunks as well: grid = (row blocks) x (chunks).  The sum over i then crosses blocks, which is a
// grid-wide dependency once per routing iteration; on this chip a kernel boundary (~1.5 us) is cheaper than an
// in-kernel grid barrier (4-10 us), so each iteration is one "phase" launch writing per-chunk partial sums plus
// a tiny finish launch (sum over chunks, squash, V += v).  Same lane/wave roles and W staging as above.
struct PhaseArgs {
  const float* u; const float* W; const float* V; const float* ds; float* slab;
  int R, N, C, it, ic, g, B;
};

template <int DIN, int DOUT, int RW, int MODE>   // MODE 0: forward partial s^t;  MODE 1: backward partial A_t
__global__ __launch_bounds__(256, 2) void routing_phase_kernel(PhaseArgs a) {
  using T = WTile<DIN, DOUT>;
  extern __shared__ __attribute__((aligned(16))) float smem[];   // [2][C][WS]
  const int t = threadIdx.x, lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int C = a.C, N = a.N, R = a.R;
  const int tile = C * T::WS;
  const bool jv = lane < C;
  const int jl = jv ? lane : 0;
  const int row0 = (blockIdx.x * 4 + wave) * RW;
  const int i0 = blockIdx.y * a.ic;
  const int i1 = min(N, i0 + a.ic);
  const long long CD = (long long)C * DOUT;
  const float invC = 1.0f / (float)C;

  float V[RW][DOUT], dsv[MODE == 1 ? RW : 1][DOUT], acc[RW][DOUT];
#pragma unroll
  for (int rr = 0; rr < RW; ++rr) {
    const int row = row0 + rr;
    const bool ok = row < R && jv;
    const long long my = (long long)(ok ? row : 0) * CD + (long long)jl * DOUT;
#pragma unroll
    for (int o = 0; o < DOUT; ++o) {
      V[rr][o] = (ok && a.V != nullptr) ? a.V[my + o] : 0.f;
      if (MODE == 1) dsv[rr][o] = ok ? a.ds[my + o] : 0.f;
      acc[rr][o] = 0.f;
    }
  }
  WStage<DIN, DOUT> stage;
  stage.load(a.W + (long long)i0 * C * T::DD, C, t);
  stage.store(smem, C, t);
  __syncthreads();
  for (int i = i0; i < i1; ++i) {
    const int cur = (i - i0) & 1;
    if (i + 1 < i1) stage.load(a.W + (long long)(i + 1) * C * T::DD, C, t);
    const float* Wl = smem + cur * tile + jl * T::WS;
    float uv[RW][DIN], uh[RW][DOUT];
#pragma unroll
    for (int rr = 0; rr < RW; ++rr) {
      const int row = row0 + rr;
      if (row < R) {
        const float* up = a.u + u_offset(row, i, N, DIN, a.g, a.B);
#pragma unroll
        for (int d = 0; d < DIN; ++d) uv[rr][d] = up[d];
      } else {
#pragma unroll
        for (int d = 0; d < DIN; ++d) uv[rr][d] = 0.f;
      }
    }
    predict<DIN, DOUT, RW>(Wl, uv, uh);
#pragma unroll
    for (int rr = 0; rr < RW; ++rr) {
      if (MODE == 0) {
        float c = invC;
        if (a.it > 0) {
          float b = 0.f;
#pragma unroll
          for (int o = 0; o < DOUT; ++o) b += uh[rr][o] * V[rr][o];
          b = jv ? b : -INFINITY;
          const float m = wave_max(b);
          const float e = jv ? expf(b - m) : 0.f;
          c = e / wave_sum(e);
        }
#pragma unroll
        for (int o = 0; o < DOUT; ++o) acc[rr][o] += c * uh[rr][o];
      } else {
        float b = 0.f, dc = 0.f;
#pragma unroll
        for (int o = 0; o < DOUT; ++o) { b += uh[rr][o] * V[rr][o]; dc += uh[rr][o] * dsv[rr][o]; }
        b = jv ? b : -INFINITY;
        const float m = wave_max(b);
        const float e = jv ? expf(b - m) : 0.f;
        const float c = e / wave_sum(e);
        const float dot = wave_sum(c * dc);
        const float db = c * (dc - dot);
#pragma unroll
        for (int o = 0; o < DOUT; ++o) acc[rr][o] += db * uh[rr][o];
      }
    }
    if (i + 1 < i1) stage.store(smem + (cur ^ 1) * tile, C, t);
    __syncthreads();
  }
#pragma unroll
  for (int rr = 0; rr < RW; ++rr) {
    const int row = row0 + rr;
    if (row < R && jv) {
      float* dst = a.slab + ((long long)blockIdx.y * R + row) * CD + (long long)lane * DOUT;
#pragma unroll
      for (int o = 0; o < DOUT; ++o) dst[o] = acc[rr][o];
    }
  }
}

// forward finish of iteration `it` (s^t already summed over chunks into s_hist[it] by slab_sum_kernel):
// v = squash(s), V (+)= v; thread <-> (row, j)
template <int DOUT>
__global__ void routing_fin_fwd_kernel(const float* __restrict__ s_hist_it, float* __restrict__ V,
                                       float* __restrict__ v_out, int R, int C, int it, int last, int g, int B) {
  const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (long long)R * C) return;
  const int row = (int)(idx / C), j = (int)(idx - (long long)row * C);
  const long long my = idx * DOUT;
  float sv[DOUT], vv[DOUT];
#pragma unroll
  for (int o = 0; o < DOUT; ++o) sv[o] = s_hist_it[my + o];
  squash_vec(sv, vv, DOUT);
#pragma unroll
  for (int o = 0; o < DOUT; ++o) V[my + o] = (it == 0 ? 0.f : V[my + o]) + vv[o];
  if (last) {
    float* vo = v_out + (out_row(row, g, B) * C + j) * DOUT;
#pragma unroll
    for (int o = 0; o < DOUT; ++o) vo[o] = vv[o];
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

// backward finish of step t (>= 1), A_t already summed over chunks into `At` by slab_sum_kernel:
// SA += A_t ; ds_all[t-1] = squash_bwd(s^{t-1}, SA)
template <int DOUT>
__global__ void routing_bwd_fin_kernel(const float* __restrict__ At, const float* __restrict__ s_hist,
                                       float* __restrict__ ds_all, float* __restrict__ SA, int R, int C, int t) {
  const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (long long)R * C) return;
  const long long plane = (long long)R * C * DOUT, my = idx * DOUT;
  float sa[DOUT], sv[DOUT], ds[DOUT];
#pragma unroll
  for (int o = 0; o < DOUT; ++o) {
    sa[o] = SA[my + o] + At[my + o];
    SA[my + o] = sa[o];
    sv[o] = s_hist[(long long)(t - 1) * plane + my + o];
  }
  squash_bwd_vec(sv, sa, ds, DOUT);
#pragma unroll
  for (int o = 0; o < DOUT; ++o) ds_all[(long long)(t - 1) * plane + my + o] = ds[o];
}

struct PhasePlan { bool phased; int row_blocks, nch, ic; };
// rows per block of the fused single-launch kernel: 4 waves x RW rows
inline PhasePlan plan_phases(int R, int N, int rw) {
  PhasePlan p;
  p.row_blocks = (R + 4 * rw - 1) / (4 * rw);
  p.phased = p.row_blocks < 128;
  int nch = (512 + p.row_blocks - 1) / p.row_blocks;
  if (nch > (N + 1) / 2) nch = (N + 1) / 2;
  if (nch < 1) nch = 1;
  p.ic = (N + nch - 1) / nch;
  p.nch = (N + p.ic - 1) / p.ic;
  return p;
}

constexpr int C1_BLOCKS = 256;             // one persistent block per CU
bool fast_c1(int N, int C, int Din, int Dout) { return C == 1 && N * Din == 4096 && Dout == 5; }

template <int DIN, int DOUT>
int launch_fwd(const cy_routing_fwd_t* a, hipStream_t s) {
  using T = WTile<DIN, DOUT>;
  constexpr int RW = (DOUT <= 16) ? 2 : 1;
  const size_t lds = (size_t)2 * a->C * T::WS * 4;
  const PhasePlan p = plan_phases(a->R, a->N, RW);
  if (!p.phased) {
    int rc = cy_allow_lds(routing_fwd_kernel<DIN, DOUT, RW>, lds);
    if (rc) return rc;
    routing_fwd_kernel<DIN, DOUT, RW><<<p.row_blocks, 256, lds, s>>>(*a);
    return 0;
  }
  if (a->ws == nullptr) return cy_set_error(CY_EINVAL, "cy_routing_fwd: this shape needs the workspace (ws) of cy_routing_fwd_ws_floats()");
  int rc = cy_allow_lds(routing_phase_kernel<DIN, DOUT, RW, 0>, lds);
  if (rc) return rc;
  const long long plane = (long long)a->R * a->C * DOUT;
  float* V = a->ws;
  float* slab = a->ws + plane;
  const int fin_blocks = (int)cy_ceil_div((long long)a->R * a->C, 128);
  for (int it = 0; it < a->n_iter; ++it) {
    PhaseArgs pa{a->u, a->W, it > 0 ? V : nullptr, nullptr, slab, a->R, a->N, a->C, it, p.ic, a->gather_g, a->gather_B};
    routing_phase_kernel<DIN, DOUT, RW, 0><<<dim3(p.row_blocks, p.nch), 256, lds, s>>>(pa);
    slab_sum_kernel<<<(unsigned)cy_ceil_div(plane, 64), 1024, 0, s>>>(slab, a->s_hist + (long long)it * plane, p.nch, plane);
    routing_fin_fwd_kernel<DOUT><<<fin_blocks, 128, 0, s>>>(a->s_hist + (long long)it * plane, V, a->v_out, a->R, a->C, it,
                                                            it == a->n_iter - 1, a->gather_g, a->gather_B);
  }
  return 0;
}
template <int DIN, int DOUT>
int launch_bwd(const cy_routing_bwd_t* a, hipStream_t s) {
  using T = WTile<DIN, DOUT>;
  const size_t lds1 = (size_t)2 * a->C * T::WS * 4;
  const PhasePlan p = plan_phases(a->R, a->N, 1);
  const long long plane = (long long)a->R * a->C * DOUT;
  int rc;
  if (!p.phased) {
    rc = cy_allow_lds(routing_bwd_rows_kernel<DIN, DOUT>, lds1);
    if (rc) return rc;
    routing_bwd_rows_kernel<DIN, DOUT><<<(a->R + 3) / 4, 256, lds1, s>>>(*a);
  } else {
    float* ds_all = a->ws;
    float* V_all = a->ws + (long long)a->n_iter * plane;
    float* SA = a->ws + 2ll * a->n_iter * plane;
    float* At = SA + plane;
    float* slab = At + plane;
    const int fin_blocks = (int)cy_ceil_div((long long)a->R * a->C, 128);
    routing_bwd_prep_kernel<DOUT><<<fin_blocks, 128, 0, s>>>(a->s_hist, a->dv, ds_all, V_all, SA, a->R, a->C, a->n_iter,
                                                             a->gather_g, a->gather_B);
    rc = cy_allow_lds(routing_phase_kernel<DIN, DOUT, 1, 1>, lds1);
    if (rc) return rc;
    for (int t = a->n_iter - 1; t >= 1; --t) {
      PhaseArgs pa{a->u, a->W, V_all + (long long)t * plane, ds_all + (long long)t * plane, slab, a->R, a->N, a->C, t, p.ic,
                   a->gather_g, a->gather_B};
      routing_phase_kernel<DIN, DOUT, 1, 1><<<dim3(p.row_blocks, p.nch), 256, lds1, s>>>(pa);
      slab_sum_kernel<<<(unsigned)cy_ceil_div(plane, 64), 1024, 0, s>>>(slab, At, p.nch, plane);
      routing_bwd_fin_kernel<DOUT><<<fin_blocks, 128, 0, s>>>(At, a->s_hist, ds_all, SA, a->R, a->C, t);
    }
  }
  const int igroups = (a->N + B2_WAVES - 1) / B2_WAVES;
  int chunks = (768 + igroups - 1) / igroups;
  if (chunks > (a->R + 15) / 16) chunks = (a->R + 15) / 16;
  if (chunks < 1) chunks = 1;
  const int rpc = (a->R + chunks - 1) / chunks;
  chunks = (a->R + rpc - 1) / rpc;
  const size_t lds2 = (size_t)B2_WAVES * a->C * T::WS * 4;
  rc = cy_allow_lds(routing_bwd_caps_kernel<DIN, DOUT>, lds2);
  if (rc) return rc;
  routing_bwd_caps_kernel<DIN, DOUT><<<dim3(igroups, chunks), 64 * B2_WAVES, lds2, s>>>(*a, rpc);
  return 0;
}

int check_shape(const char* fn, int R, int N, int C, int Din, int Dout, int n_iter, int g, int B) {
  if (R <= 0 || N <= 0 || C <= 0 || n_iter <= 0) return cy_set_error(CY_EINVAL, "%s: non-positive dimension", fn);
  if (g != 0 && (N != 512 || Din != 8 || B <= 0 || R != g * g * B))
    return cy_set_error(CY_EINVAL, "%s: cell gather needs N=512, Din=8, R=g*g*B (got N=%d Din=%d R=%d g=%d B=%d)", fn, N,
                        Din, R, g, B);
  if (fast_c1(N, C, Din, Dout)) return 0;
  if (Din != 8 || !(Dout == 5 || Dout == 16 || Dout == 21) || C > 64)
    return cy_set_error(CY_EINVAL, "%s: unsupported capsule shape C=%d Din=%d Dout=%d (built: Din=8, Dout in {5,16,21}, C<=64)",
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
  } else if (a->Dout == 5) rc = launch_fwd<8, 5>(a, s);
  else if (a->Dout == 16) rc = launch_fwd<8, 16>(a, s);
  else rc = launch_fwd<8, 21>(a, s);
  if (rc) return rc;
  CY_LAUNCH_CHECK("cy_routing_fwd");
  return 0;
}

extern "C" long long cy_routing_fwd_ws_floats(const cy_routing_fwd_t* a) {
  if (!a || fast_c1(a->N, a->C, a->Din, a->Dout)) return 0;
  const PhasePlan p = plan_phases(a->R, a->N, a->Dout <= 16 ? 2 : 1);
  if (!p.phased) return 0;
  return (1ll + p.nch) * a->R * a->C * a->Dout;
}

extern "C" long long cy_routing_bwd_ws_floats(const cy_routing_bwd_t* a) {
  if (!a) return 0;
  if (fast_c1(a->N, a->C, a->Din, a->Dout)) return (long long)C1_BLOCKS * 4096 * 5;
  const PhasePlan p = plan_phases(a->R, a->N, 1);
  const long long plane = (long long)a->R * a->C * a->Dout;
  return 2ll * a->n_iter * plane + (p.phased ? (2ll + p.nch) * plane : 0);
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
  if (a->Dout == 5) rc = launch_bwd<8, 5>(a, s);
  else if (a->Dout == 16) rc = launch_bwd<8, 16>(a, s);
  else rc = launch_bwd<8, 21>(a, s);
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
