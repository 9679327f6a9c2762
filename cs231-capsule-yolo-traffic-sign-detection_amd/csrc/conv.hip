// Convolution as implicit GEMM on the fp32 matrix cores (v_mfma_f32_32x32x2_f32), gfx950.
//
// Replaces nn.Conv2d forward / dgrad / wgrad of the reference (models.py:90, 60-62, 98-110,
// 132-223, 347-363).  Activations are NHWC so that the GEMM K dimension (tap, cin) is
// contiguous in cin; one kernel serves forward and input-gradient (see capsyolo_hip.h).
//
// Tiling (forward / dgrad):  block = 256 threads = 4 waves (2 x 2), block tile 128 pixels x BN
// channels (BN = 64*NTW), K step 32.  Each wave owns 2 x NTW accumulator tiles of 32x32.
// LDS image of both operands: [kq = k/4 (8)][row][4 floats] with a 16-byte pad per kq, so that
//   * the global->LDS write of a float4 (4 consecutive cin of one pixel) is one ds_write_b128,
//     conflict-free across the 8 lanes that cover one pixel's 128 bytes, and
//   * lane (i = lane&31, h = lane>>5) fetches 4 MFMA steps of its operand with ONE ds_read_b128
//     at [kq = 2*kg + h][row i]: step t multiplies k = 8kg+t (h=0) and 8kg+4+t (h=1).
// Both operands use the same k permutation, so the sum over k is unchanged.
#include "common.h"

namespace {

constexpr int BM = 128;                   // pixels per block tile
constexpr int A_KQ = BM * 4 + 4;          // floats per kq slab of the A image (padded)

struct GemmGeom {
  int Np, K, KT;
  long long M;
};

__device__ __forceinline__ f32x16 mfma32(float a, float b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
}

// ------------------------------------------------------------------------------------------------
template <int NTW, bool VEC>
__global__ __launch_bounds__(256, 2) void conv_gemm_kernel(cy_conv_gemm_t a, GemmGeom g) {
  constexpr int BN = 64 * NTW;
  constexpr int B_KQ = BN * 4 + 4;
  constexpr int A_BUF = 8 * A_KQ, B_BUF = 8 * B_KQ;
  constexpr int NBQ = BN / 32;            // float4 of B per thread per K tile
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* As = smem;
  float* Bs = smem + 2 * A_BUF;
  long long* rowoff = (long long*)(Bs + 2 * B_BUF);
  int* ktab = (int*)(rowoff + BM);        // scalar loader only: [KT*32] {c | dy<<16 | dx<<24}, -1 = padding

  const int t = threadIdx.x;
  const int lane = t & 63, wave = t >> 6;
  const int wave_m = wave & 1, wave_n = wave >> 1;
  const int li = lane & 31, lh = lane >> 5;

  const int ntiles_n = g.Np / BN;
  const long long mt = blockIdx.x / ntiles_n;
  const int nt = blockIdx.x % ntiles_n;
  const long long m0 = mt * BM;
  const int n0 = nt * BN;
  const int HoWo = a.Ho * a.Wo;

  // ---- per-block tables
  if (t < BM) {
    long long p = m0 + t, off = -1;
    if (p < g.M) {
      int b = (int)(p / HoWo);
      int r = (int)(p - (long long)b * HoWo);
      int oy = r / a.Wo, ox = r - oy * a.Wo;
      off = (((long long)b * a.Hy + (oy * a.out_stride + a.out_oy)) * a.Wy + (ox * a.out_stride + a.out_ox)) * a.N;
    }
    rowoff[t] = off;
  }
  if (!VEC) {
    for (int k = t; k < g.KT * 32; k += 256) {
      int ent = -1;
      if (k < g.K) {
        int tap = k / a.Cin, c = k - tap * a.Cin;
        int ta = tap / a.TW, tb = tap - ta * a.TW;
        int dy = a.dy0 + ta * a.dstep, dx = a.dx0 + tb * a.dstep;
        ent = (c & 0xffff) | ((dy & 0xff) << 16) | ((dx & 0xff) << 24);
      }
      ktab[k] = ent;
    }
  }

  // ---- loader state
  // VEC:    thread owns float4 (pixel (t>>3)+32q, kq = t&7), q = 0..3
  // scalar: thread owns pixel t&127, kq = 4*(t>>7) .. +3
  constexpr int NPIX = VEC ? 4 : 1;
  int iy0[NPIX], ix0[NPIX];
  long long xb[NPIX];
#pragma unroll
  for (int q = 0; q < NPIX; ++q) {
    int pl = VEC ? ((t >> 3) + 32 * q) : (t & 127);
    long long p = m0 + pl;
    iy0[q] = -(1 << 28); ix0[q] = -(1 << 28); xb[q] = 0;
    if (p < g.M) {
      int b = (int)(p / HoWo);
      int r = (int)(p - (long long)b * HoWo);
      int oy = r / a.Wo, ox = r - oy * a.Wo;
      iy0[q] = oy * a.in_stride + (VEC ? a.dy0 : 0);
      ix0[q] = ox * a.in_stride + (VEC ? a.dx0 : 0);
      xb[q] = (long long)b * a.xs_b;
    }
  }
  const int kq_ld = t & 7;
  float4 ra[4], rb[NBQ];
  int tap_a = 0, tap_b = 0, c0 = 0;       // position of the NEXT tile to load (VEC)

  auto load_tile = [&](int kt) {
    if (VEC) {
      const int dy = tap_a * a.dstep, dx = tap_b * a.dstep;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        int iy = iy0[q] + dy, ix = ix0[q] + dx;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if ((unsigned)iy < (unsigned)a.Hi && (unsigned)ix < (unsigned)a.Wi)
          v = *(const float4*)(a.X + xb[q] + (long long)iy * a.xs_y + (long long)ix * a.xs_x + c0 + kq_ld * 4);
        ra[q] = v;
      }
      c0 += 32;
      if (c0 >= a.Cin) { c0 = 0; if (++tap_b == a.TW) { tap_b = 0; ++tap_a; } }
    } else {
      const int half = t >> 7;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        float v[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          int ent = ktab[kt * 32 + (half * 4 + j) * 4 + e];
          int c = ent & 0xffff;
          int iy = iy0[0] + (int)(signed char)((ent >> 16) & 0xff);
          int ix = ix0[0] + (int)(signed char)((ent >> 24) & 0xff);
          float x = 0.f;
          if (ent != -1 && (unsigned)iy < (unsigned)a.Hi && (unsigned)ix < (unsigned)a.Wi)
            x = a.X[xb[0] + (long long)iy * a.xs_y + (long long)ix * a.xs_x + (long long)c * a.xs_c];
          v[e] = x;
        }
        ra[j] = make_float4(v[0], v[1], v[2], v[3]);
      }
    }
#pragma unroll
    for (int q = 0; q < NBQ; ++q) {
      int f = t + 256 * q;
      int kq = f / BN, n = f % BN;
      rb[q] = *(const float4*)(a.Wp + ((long long)(kt * 8 + kq) * g.Np + n0 + n) * 4);
    }
  };
  auto store_tile = [&](int buf) {
    float* Ab = As + buf * A_BUF;
    float* Bb = Bs + buf * B_BUF;
    if (VEC) {
#pragma unroll
      for (int q = 0; q < 4; ++q) *(float4*)(Ab + kq_ld * A_KQ + ((t >> 3) + 32 * q) * 4) = ra[q];
    } else {
#pragma unroll
      for (int j = 0; j < 4; ++j) *(float4*)(Ab + ((t >> 7) * 4 + j) * A_KQ + (t & 127) * 4) = ra[j];
    }
#pragma unroll
    for (int q = 0; q < NBQ; ++q) {
      int f = t + 256 * q;
      int kq = f / BN, n = f % BN;
      *(float4*)(Bb + kq * B_KQ + n * 4) = rb[q];
    }
  };

  f32x16 acc[2][NTW];
#pragma unroll
  for (int mi = 0; mi < 2; ++mi)
#pragma unroll
    for (int ni = 0; ni < NTW; ++ni)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.f;

  __syncthreads();                         // ktab / rowoff visible
  load_tile(0);
  store_tile(0);
  __syncthreads();

  for (int kt = 0; kt < g.KT; ++kt) {
    const int cur = kt & 1;
    if (kt + 1 < g.KT) load_tile(kt + 1);
    const float* Ab = As + cur * A_BUF + (wave_m * 64 + li) * 4;
    const float* Bb = Bs + cur * B_BUF + (wave_n * 32 * NTW + li) * 4;
#pragma unroll
    for (int kg = 0; kg < 4; ++kg) {
      const int kq = 2 * kg + lh;
      float4 fa[2], fb[NTW];
#pragma unroll
      for (int mi = 0; mi < 2; ++mi) fa[mi] = *(const float4*)(Ab + kq * A_KQ + mi * 128);
#pragma unroll
      for (int ni = 0; ni < NTW; ++ni) fb[ni] = *(const float4*)(Bb + kq * B_KQ + ni * 128);
#pragma unroll
      for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int ni = 0; ni < NTW; ++ni) {
          acc[mi][ni] = mfma32(fa[mi].x, fb[ni].x, acc[mi][ni]);
          acc[mi][ni] = mfma32(fa[mi].y, fb[ni].y, acc[mi][ni]);
          acc[mi][ni] = mfma32(fa[mi].z, fb[ni].z, acc[mi][ni]);
          acc[mi][ni] = mfma32(fa[mi].w, fb[ni].w, acc[mi][ni]);
        }
    }
    if (kt + 1 < g.KT) store_tile(cur ^ 1);
    __syncthreads();
  }

  // ---- epilogue: bias, activation, store, optional BatchNorm statistics
  float ssum[NTW], ssq[NTW];
#pragma unroll
  for (int ni = 0; ni < NTW; ++ni) {
    ssum[ni] = 0.f; ssq[ni] = 0.f;
    const int n = n0 + wave_n * 32 * NTW + ni * 32 + li;
    const float bv = (a.bias != nullptr && n < a.N) ? a.bias[n] : 0.f;
#pragma unroll
    for (int mi = 0; mi < 2; ++mi) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = wave_m * 64 + mi * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
        const long long off = rowoff[row];
        float v = acc[mi][ni][r] + bv;
        if (off >= 0 && n < a.N) {
          ssum[ni] += v; ssq[ni] += v * v;
          if (a.act == 1) v = fmaxf(v, 0.f);
          a.Y[off + n] = v;
        }
      }
    }
  }
  if (a.stats != nullptr) {
    float* red = smem;                     // reuse the A image: [2 wave_m][BN][2]
#pragma unroll
    for (int ni = 0; ni < NTW; ++ni) {
      float s = ssum[ni] + __shfl_xor(ssum[ni], 32, 64);
      float q = ssq[ni] + __shfl_xor(ssq[ni], 32, 64);
      if (lh == 0) {
        const int col = wave_n * 32 * NTW + ni * 32 + li;
        red[(wave_m * BN + col) * 2 + 0] = s;
        red[(wave_m * BN + col) * 2 + 1] = q;
      }
    }
    __syncthreads();
    if (t < BN && n0 + t < a.N) {
      double s = (double)red[t * 2] + (double)red[(BN + t) * 2];
      double q = (double)red[t * 2 + 1] + (double)red[(BN + t) * 2 + 1];
      atomicAdd(a.stats + 2 * (n0 + t), s);
      atomicAdd(a.stats + 2 * (n0 + t) + 1, q);
    }
  }
}

// ------------------------------------------------------------------------------------------------
// Weight packing: Wp[((kt*8 + kq)*Np + n)*4 + e] = element (k = kt*32 + kq*4 + e, n) of the GEMM B operand.
__global__ void pack_weights_kernel(const float* __restrict__ W, float* __restrict__ Wp, int Cout, int Cin, int KH,
                                    int KW, int TH, int TW, int kh0, int kw0, int kstep, int transpose, int K,
                                    int Np, long long total) {
  const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= total) return;
  const int e = (int)(idx & 3);
  const long long r = idx >> 2;
  const int n = (int)(r % Np);
  const long long kk = r / Np;                 // kt*8 + kq
  const int k = (int)(kk * 4 + e);
  const int rows_per_tap = transpose ? Cout : Cin;
  const int ncols = transpose ? Cin : Cout;
  float v = 0.f;
  if (k < K && n < ncols) {
    const int tap = k / rows_per_tap, rr = k - tap * rows_per_tap;
    const int ta = tap / TW, tb = tap - ta * TW;
    const int kh = kh0 + ta * kstep, kw = kw0 + tb * kstep;
    const int co = transpose ? rr : n, ci = transpose ? n : rr;
    v = W[(((long long)co * Cin + ci) * KH + kh) * KW + kw];
  }
  Wp[idx] = v;
}

// ------------------------------------------------------------------------------------------------
// Weight gradient: slab[split][k][n] = sum over the split's pixels of Xpatch[p][k] * dZ[p][n].
// Both operands are k(reduction)-major in NHWC memory already: lane (i,h) of MFMA step s reads
// X_lds[2s+h][i] and dZ_lds[2s+h][j] with ds_read_b32, conflict-free.
constexpr int PT = 32;                     // pixels per pipeline stage

template <int MT, int NT, int WM, int WN, bool VEC>
__global__ __launch_bounds__(256, 2) void conv_wgrad_kernel(cy_conv_wgrad_t a, int K, long long M, long long pix_per_split) {
  constexpr int BMK = 32 * MT * WM, BNN = 32 * NT * WN;
  constexpr int LDA = BMK, LDB = BNN;
  constexpr int A_BUF = PT * LDA, B_BUF = PT * LDB;
  constexpr int NA = (PT * BMK / 4 + 255) / 256;   // float4 per thread (VEC) per stage
  constexpr int NB = (PT * BNN / 4 + 255) / 256;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* As = smem;
  float* Zs = smem + 2 * A_BUF;
  int* ktab = (int*)(Zs + 2 * B_BUF);      // scalar loader: [BMK] {c | dy<<16 | dx<<24}

  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int wm = wave % WM, wn = wave / WM;
  const int li = lane & 31, lh = lane >> 5;
  const int kk0 = blockIdx.x * BMK, n0 = blockIdx.y * BNN;
  const long long p_begin = (long long)blockIdx.z * pix_per_split;
  long long p_end = p_begin + pix_per_split;
  if (p_end > M) p_end = M;
  const int HoWo = a.Ho * a.Wo;

  // A loader geometry
  constexpr int UA = BMK / 4;              // float4 units per pixel row
  int a_dy[NA], a_dx[NA], a_c[NA], a_row[NA], a_u[NA];
  bool a_ok[NA];
  if (VEC) {
#pragma unroll
    for (int j = 0; j < NA; ++j) {
      int f = t + 256 * j;
      a_row[j] = f / UA; a_u[j] = f % UA;
      int kk = kk0 + a_u[j] * 4;
      a_ok[j] = (f < PT * UA) && (kk < K);
      int tap = kk / a.Cin, c = kk - tap * a.Cin;
      int kh = tap / a.KW, kw = tap - kh * a.KW;
      a_dy[j] = kh - a.pad; a_dx[j] = kw - a.pad; a_c[j] = c;
    }
  } else {
    for (int k = t; k < BMK; k += 256) {
      int kk = kk0 + k, ent = -1;
      if (kk < K) {
        int tap = kk / a.Cin, c = kk - tap * a.Cin;
        int kh = tap / a.KW, kw = tap - kh * a.KW;
        ent = (c & 0xffff) | (((kh - a.pad) & 0xff) << 16) | (((kw - a.pad) & 0xff) << 24);
      }
      ktab[k] = ent;
    }
    __syncthreads();
  }
  constexpr int UB = BNN / 4;
  const bool zvec = ((a.N & 3) == 0) && (((uintptr_t)a.dZ & 15) == 0);
  float4 ra[VEC ? NA : (PT * BMK / 256 / 4 > 0 ? PT * BMK / 256 / 4 : 1)];
  float4 rb[NB];

  auto decompose = [&](long long p, int& b, int& oy, int& ox) {
    b = (int)(p / HoWo);
    int r = (int)(p - (long long)b * HoWo);
    oy = r / a.Wo; ox = r - oy * a.Wo;
  };
  auto load_stage = [&](long long ps) {
    if (VEC) {
#pragma unroll
      for (int j = 0; j < NA; ++j) {
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        long long p = ps + a_row[j];
        if (a_ok[j] && p < p_end) {
          int b, oy, ox; decompose(p, b, oy, ox);
          int iy = oy * a.stride + a_dy[j], ix = ox * a.stride + a_dx[j];
          if ((unsigned)iy < (unsigned)a.Hi && (unsigned)ix < (unsigned)a.Wi)
            v = *(const float4*)(a.X + (long long)b * a.xs_b + (long long)iy * a.xs_y + (long long)ix * a.xs_x + a_c[j]);
        }
        ra[j] = v;
      }
    } else {
      // scalar: thread -> pixel row (t & 31), k group (t >> 5) of BMK/8 elements (BMK = 32 -> 4 floats)
      constexpr int EPT = PT * BMK / 256;  // elements per thread (multiple of 4)
      const int row = t % PT, kbase = (t / PT) * EPT;
      long long p = ps + row;
      int b = 0, oy = 0, ox = 0;
      const bool pv = p < p_end;
      if (pv) decompose(p, b, oy, ox);
#pragma unroll
      for (int j = 0; j < EPT / 4; ++j) {
        float v[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          int ent = ktab[kbase + j * 4 + e];
          float x = 0.f;
          if (pv && ent != -1) {
            int iy = oy * a.stride + (int)(signed char)((ent >> 16) & 0xff);
            int ix = ox * a.stride + (int)(signed char)((ent >> 24) & 0xff);
            if ((unsigned)iy < (unsigned)a.Hi && (unsigned)ix < (unsigned)a.Wi)
              x = a.X[(long long)b * a.xs_b + (long long)iy * a.xs_y + (long long)ix * a.xs_x + (long long)(ent & 0xffff) * a.xs_c];
          }
          v[e] = x;
        }
        ra[j] = make_float4(v[0], v[1], v[2], v[3]);
      }
    }
#pragma unroll
    for (int j = 0; j < NB; ++j) {
      int f = t + 256 * j;
      int row = f / UB, u = f % UB;
      long long p = ps + row;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (f < PT * UB && p < p_end) {
        const float* src = a.dZ + p * a.N + n0 + u * 4;
        if (zvec && n0 + u * 4 + 3 < a.N) v = *(const float4*)src;
        else {
          float tmp[4] = {0.f, 0.f, 0.f, 0.f};
          for (int e = 0; e < 4; ++e) if (n0 + u * 4 + e < a.N) tmp[e] = src[e];
          v = make_float4(tmp[0], tmp[1], tmp[2], tmp[3]);
        }
      }
      rb[j] = v;
    }
  };
  auto store_stage = [&](int buf) {
    float* Ab = As + buf * A_BUF;
    float* Zb = Zs + buf * B_BUF;
    if (VEC) {
#pragma unroll
      for (int j = 0; j < NA; ++j) {
        int f = t + 256 * j;
        if (f < PT * UA) *(float4*)(Ab + a_row[j] * LDA + a_u[j] * 4) = ra[j];
      }
    } else {
      constexpr int EPT = PT * BMK / 256;
      const int row = t % PT, kbase = (t / PT) * EPT;
#pragma unroll
      for (int j = 0; j < EPT / 4; ++j) *(float4*)(Ab + row * LDA + kbase + j * 4) = ra[j];
    }
#pragma unroll
    for (int j = 0; j < NB; ++j) {
      int f = t + 256 * j;
      if (f < PT * UB) *(float4*)(Zb + (f / UB) * LDB + (f % UB) * 4) = rb[j];
    }
  };

  f32x16 acc[MT][NT];
#pragma unroll
  for (int mi = 0; mi < MT; ++mi)
#pragma unroll
    for (int ni = 0; ni < NT; ++ni)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.f;

  const long long nstages = (p_end > p_begin) ? (p_end - p_begin + PT - 1) / PT : 0;
  if (nstages > 0) {
    load_stage(p_begin);
    store_stage(0);
  }
  __syncthreads();
  for (long long st = 0; st < nstages; ++st) {
    const int cur = (int)(st & 1);
    if (st + 1 < nstages) load_stage(p_begin + (st + 1) * PT);
    const float* Ab = As + cur * A_BUF + wm * 32 * MT + li;
    const float* Zb = Zs + cur * B_BUF + wn * 32 * NT + li;
#pragma unroll
    for (int s = 0; s < PT / 2; ++s) {
      float fa[MT], fb[NT];
#pragma unroll
      for (int mi = 0; mi < MT; ++mi) fa[mi] = Ab[(2 * s + lh) * LDA + mi * 32];
#pragma unroll
      for (int ni = 0; ni < NT; ++ni) fb[ni] = Zb[(2 * s + lh) * LDB + ni * 32];
#pragma unroll
      for (int mi = 0; mi < MT; ++mi)
#pragma unroll
        for (int ni = 0; ni < NT; ++ni) acc[mi][ni] = mfma32(fa[mi], fb[ni], acc[mi][ni]);
    }
    if (st + 1 < nstages) store_stage(cur ^ 1);
    __syncthreads();
  }

  float* slab = a.slabs + (long long)blockIdx.z * K * a.N;
#pragma unroll
  for (int mi = 0; mi < MT; ++mi)
#pragma unroll
    for (int ni = 0; ni < NT; ++ni) {
      const int n = n0 + wn * 32 * NT + ni * 32 + li;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int kk = kk0 + wm * 32 * MT + mi * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
        if (kk < K && n < a.N) slab[(long long)kk * a.N + n] = acc[mi][ni][r];
      }
    }
}

// dW[co][ci][kh][kw] = sum_s slab[s][(kh*KW+kw)*Cin + ci][co]
__global__ void wgrad_reduce_kernel(const float* __restrict__ slabs, float* __restrict__ dW, int S, int K, int N,
                                    int Cin, int KH, int KW) {
  const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const long long total = (long long)K * N;
  if (idx >= total) return;
  const int n = (int)(idx % N);
  const int kk = (int)(idx / N);
  float s = 0.f;
  for (int i = 0; i < S; ++i) s += slabs[(long long)i * total + idx];
  const int tap = kk / Cin, ci = kk - tap * Cin;
  const int kh = tap / KW, kw = tap - kh * KW;
  dW[(((long long)n * Cin + ci) * KH + kh) * KW + kw] = s;
}

// out[n] = sum_p dZ[p][n]; grid (ceil(N/64), splits); one atomic per (block, n)
__global__ void channel_sum_kernel(const float* __restrict__ dZ, float* __restrict__ out, long long P, int N,
                                   long long rows_per_block) {
  __shared__ float red[4][64];
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int n = blockIdx.x * 64 + lane;
  const long long r0 = (long long)blockIdx.y * rows_per_block;
  long long r1 = r0 + rows_per_block;
  if (r1 > P) r1 = P;
  float s = 0.f;
  if (n < N)
    for (long long r = r0 + wave; r < r1; r += 4) s += dZ[r * N + n];
  red[wave][lane] = s;
  __syncthreads();
  if (wave == 0 && n < N) atomicAdd(out + n, red[0][lane] + red[1][lane] + red[2][lane] + red[3][lane]);
}

int pick_splits(int tiles, long long M) {
  long long s = (1536 + tiles - 1) / tiles;
  long long cap = M / 2048;
  if (cap < 1) cap = 1;
  if (s > cap) s = cap;
  if (s < 1) s = 1;
  if (s > 512) s = 512;
  return (int)s;
}

struct WgradPlan { int variant, BMK, BNN, ktiles, ntiles, S; long long pix_per_split; bool vec; };

WgradPlan plan_wgrad(const cy_conv_wgrad_t* a) {
  WgradPlan p;
  const int K = a->KH * a->KW * a->Cin;
  const long long M = (long long)a->B * a->Ho * a->Wo;
  p.vec = (a->xs_c == 1) && (a->Cin % 4 == 0) && (a->xs_x % 4 == 0) && (a->xs_y % 4 == 0) && (a->xs_b % 4 == 0) &&
          (((uintptr_t)a->X & 15) == 0);
  if (!p.vec || K <= 32) { p.variant = 2; p.BMK = 32; p.BNN = 128; p.vec = false; }
  else if (a->N > 64) { p.variant = 0; p.BMK = 128; p.BNN = 128; }
  else { p.variant = 1; p.BMK = 128; p.BNN = 64; }
  p.ktiles = (K + p.BMK - 1) / p.BMK;
  p.ntiles = (a->N + p.BNN - 1) / p.BNN;
  p.S = pick_splits(p.ktiles * p.ntiles, M);
  long long pps = (M + p.S - 1) / p.S;
  pps = (pps + PT - 1) / PT * PT;
  p.pix_per_split = pps;
  p.S = (int)((M + pps - 1) / pps);
  return p;
}

}  // namespace

// ================================================================================================ C-ABI
extern "C" long long cy_conv_packed_floats(int K, int N) {
  long long KT = (K + 31) / 32, Np = (N + 63) / 64 * 64;
  return KT * 32 * Np;
}

extern "C" int cy_conv_pack_weights(const float* W, float* Wp, int Cout, int Cin, int KH, int KW, int TH, int TW,
                                    int kh0, int kw0, int kstep, int transpose, void* stream) {
  CY_REQUIRE(W && Wp, "cy_conv_pack_weights: null pointer");
  CY_REQUIRE(TH >= 1 && TW >= 1 && kh0 + (TH - 1) * kstep < KH && kw0 + (TW - 1) * kstep < KW,
             "cy_conv_pack_weights: tap set (%d,%d,+%d x %d) / (%d,+%d x %d) outside %dx%d kernel", kh0, kstep, TH,
             TH, kw0, kstep, TW, KH, KW);
  const int K = TH * TW * (transpose ? Cout : Cin);
  const int N = transpose ? Cin : Cout;
  const int Np = (N + 63) / 64 * 64;
  const long long total = cy_conv_packed_floats(K, N);
  const int threads = 256;
  pack_weights_kernel<<<(unsigned)cy_ceil_div(total, threads), threads, 0, (hipStream_t)stream>>>(
      W, Wp, Cout, Cin, KH, KW, TH, TW, kh0, kw0, kstep, transpose, K, Np, total);
  CY_LAUNCH_CHECK("cy_conv_pack_weights");
  return 0;
}

extern "C" int cy_conv_gemm(const cy_conv_gemm_t* a, void* stream) {
  CY_REQUIRE(a && a->X && a->Wp && a->Y, "cy_conv_gemm: null pointer");
  CY_REQUIRE(a->B > 0 && a->Ho > 0 && a->Wo > 0 && a->N > 0 && a->Cin > 0 && a->TH > 0 && a->TW > 0,
             "cy_conv_gemm: non-positive dimension");
  CY_REQUIRE(a->TH * a->TW <= 4096, "cy_conv_gemm: too many taps");
  GemmGeom g;
  g.K = a->TH * a->TW * a->Cin;
  g.KT = (g.K + 31) / 32;
  g.Np = (a->N + 63) / 64 * 64;
  g.M = (long long)a->B * a->Ho * a->Wo;
  const bool vec = (a->xs_c == 1) && (a->Cin % 32 == 0) && (a->xs_x % 4 == 0) && (a->xs_y % 4 == 0) &&
                   (a->xs_b % 4 == 0) && (((uintptr_t)a->X & 15) == 0);
  if (!vec) {
    CY_REQUIRE(g.KT * 32 <= 2048, "cy_conv_gemm: scalar loader supports K <= 2048 (got %d)", g.K);
    const int maxd = 127;
    CY_REQUIRE(abs(a->dy0) + a->TH * abs(a->dstep) <= maxd && abs(a->dx0) + a->TW * abs(a->dstep) <= maxd,
               "cy_conv_gemm: tap offsets out of range");
    CY_REQUIRE(a->Cin <= 65535, "cy_conv_gemm: scalar loader Cin too large");
  }
  const int ntw = (g.Np % 128 == 0) ? 2 : 1;
  const int BN = 64 * ntw;
  const long long mtiles = cy_ceil_div(g.M, BM);
  const long long nblocks = mtiles * (g.Np / BN);
  CY_REQUIRE(nblocks < (1ll << 31), "cy_conv_gemm: grid too large");
  size_t lds = (size_t)(2 * 8 * A_KQ + 2 * 8 * (BN * 4 + 4)) * 4 + BM * 8 + (vec ? 0 : (size_t)g.KT * 32 * 4);
  hipStream_t s = (hipStream_t)stream;
  dim3 grid((unsigned)nblocks), block(256);
#define CY_GEMM_LAUNCH(NTW_, VEC_)                                        \
  do {                                                                    \
    int rc__ = cy_allow_lds(conv_gemm_kernel<NTW_, VEC_>, lds);           \
    if (rc__) return rc__;                                                \
    conv_gemm_kernel<NTW_, VEC_><<<grid, block, lds, s>>>(*a, g);         \
  } while (0)
  if (vec && ntw == 2) CY_GEMM_LAUNCH(2, true);
  else if (vec) CY_GEMM_LAUNCH(1, true);
  else if (ntw == 2) CY_GEMM_LAUNCH(2, false);
  else CY_GEMM_LAUNCH(1, false);
#undef CY_GEMM_LAUNCH
  CY_LAUNCH_CHECK("cy_conv_gemm");
  return 0;
}

extern "C" long long cy_conv_wgrad_ws_floats(const cy_conv_wgrad_t* a) {
  if (!a) return 0;
  WgradPlan p = plan_wgrad(a);
  return (long long)p.S * a->KH * a->KW * a->Cin * a->N;
}

extern "C" int cy_conv_wgrad(const cy_conv_wgrad_t* a, void* stream) {
  CY_REQUIRE(a && a->X && a->dZ && a->dW && a->slabs, "cy_conv_wgrad: null pointer");
  CY_REQUIRE(a->B > 0 && a->Ho > 0 && a->Wo > 0 && a->N > 0 && a->Cin > 0, "cy_conv_wgrad: non-positive dimension");
  CY_REQUIRE(a->KH <= 64 && a->KW <= 64 && a->pad <= 64, "cy_conv_wgrad: kernel too large");
  const int K = a->KH * a->KW * a->Cin;
  const long long M = (long long)a->B * a->Ho * a->Wo;
  WgradPlan p = plan_wgrad(a);
  CY_REQUIRE(p.vec || a->Cin <= 65535, "cy_conv_wgrad: Cin too large for the scalar loader");
  hipStream_t s = (hipStream_t)stream;
  dim3 grid(p.ktiles, p.ntiles, p.S), block(256);
  size_t lds = (size_t)(2 * PT * p.BMK + 2 * PT * p.BNN) * 4 + (p.vec ? 0 : p.BMK * 4);
#define CY_WGRAD_LAUNCH(...)                                                       \
  do {                                                                             \
    int rc__ = cy_allow_lds(conv_wgrad_kernel<__VA_ARGS__>, lds);                  \
    if (rc__) return rc__;                                                         \
    conv_wgrad_kernel<__VA_ARGS__><<<grid, block, lds, s>>>(*a, K, M, p.pix_per_split); \
  } while (0)
  if (p.variant == 0) CY_WGRAD_LAUNCH(2, 2, 2, 2, true);
  else if (p.variant == 1) CY_WGRAD_LAUNCH(2, 1, 2, 2, true);
  else CY_WGRAD_LAUNCH(1, 1, 1, 4, false);
#undef CY_WGRAD_LAUNCH
  CY_LAUNCH_CHECK("cy_conv_wgrad");
  const long long total = (long long)K * a->N;
  wgrad_reduce_kernel<<<(unsigned)cy_ceil_div(total, 256), 256, 0, s>>>(a->slabs, a->dW, p.S, K, a->N, a->Cin, a->KH,
                                                                       a->KW);
  CY_LAUNCH_CHECK("cy_conv_wgrad(reduce)");
  return 0;
}

extern "C" int cy_channel_sum(const float* dZ, float* out, long long P, int N, void* stream) {
  CY_REQUIRE(dZ && out && P > 0 && N > 0, "cy_channel_sum: bad arguments");
  hipStream_t s = (hipStream_t)stream;
  hipError_t e = hipMemsetAsync(out, 0, (size_t)N * 4, s);
  if (e != hipSuccess) return cy_set_error((int)e, "cy_channel_sum: memset: %s", hipGetErrorString(e));
  const int nb = (N + 63) / 64;
  long long splits = cy_ceil_div(2048, nb);
  if (splits > cy_ceil_div(P, 64)) splits = cy_ceil_div(P, 64);
  const long long rpb = cy_ceil_div(P, splits);
  splits = cy_ceil_div(P, rpb);
  channel_sum_kernel<<<dim3(nb, (unsigned)splits), 256, 0, s>>>(dZ, out, P, N, rpb);
  CY_LAUNCH_CHECK("cy_channel_sum");
  return 0;
}
