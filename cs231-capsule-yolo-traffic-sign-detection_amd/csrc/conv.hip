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
  int corder;          // 1: K tiles run taps fastest, channel blocks slowest (input pixels are re-read while they are in L2)
  // split of the reduction (round 4): a layer whose output grid is far below the 256 CUs (CapsuleNet's primary-capsule convolution:
  // 42 blocks x 648 K tiles) runs S blocks per output tile on kt_per K tiles each; every block leaves its RAW partial sums in its
  // slab ws[split][M][Np] and conv_gemm_splitk_finish adds the slabs in split order (deterministic), then bias / activation.
  int S, kt_per, tiles;
  float* ws;
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
  const int sp = g.S > 1 ? (int)(blockIdx.x / (unsigned)g.tiles) : 0;      // this block's share of the reduction
  const unsigned tile_id = g.S > 1 ? blockIdx.x % (unsigned)g.tiles : blockIdx.x;
  const int kb = sp * g.kt_per, ke = (kb + g.kt_per < g.KT) ? kb + g.kt_per : g.KT;
  const long long mt = tile_id / ntiles_n;
  const int nt = tile_id % ntiles_n;
  const long long m0 = mt * BM;
  const int n0 = nt * BN;
  const int HoWo = a.Ho * a.Wo;

  // ---- per-block tables
  if (t < BM) {
    long long p = m0 + t, off = -1;
    if (p < g.M) {
      const unsigned pp = (unsigned)p;
      const unsigned b = pp / (unsigned)HoWo, r = pp - b * (unsigned)HoWo;
      const int oy = (int)(r / (unsigned)a.Wo), ox = (int)(r - (unsigned)oy * (unsigned)a.Wo);
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
  long long xo[NPIX];                     // VEC: offset of tap (0,0), channel kq*4; scalar: batch base (floats)
#pragma unroll
  for (int q = 0; q < NPIX; ++q) {
    const int pl = VEC ? ((t >> 3) + 32 * q) : (t & 127);
    const long long p = m0 + pl;
    iy0[q] = -(1 << 28); ix0[q] = -(1 << 28); xo[q] = 0;
    if (p < g.M) {
      const unsigned pp = (unsigned)p;
      const unsigned b = pp / (unsigned)HoWo, r = pp - b * (unsigned)HoWo;
      const int oy = (int)(r / (unsigned)a.Wo), ox = (int)(r - (unsigned)oy * (unsigned)a.Wo);
      iy0[q] = oy * a.in_stride + (VEC ? a.dy0 : 0);
      ix0[q] = ox * a.in_stride + (VEC ? a.dx0 : 0);
      xo[q] = (long long)b * a.xs_b +
              (VEC ? ((long long)iy0[q] * a.xs_y + (long long)ix0[q] * a.xs_x + (t & 7) * 4) : 0ll);
    }
  }
  float4 ra0, ra1, ra2, ra3;
  f32x4 rb[NBQ];
  int tap_a = 0, tap_b = 0, c0 = 0;       // position of the NEXT tile to load (VEC)
  // B tile: float4 index f = t + 256q -> kq = f / BN, n = f % BN  (BN is a power of two)
  const int b_kq0 = t / BN, b_n = t % BN;
  const float* wsrc = a.Wp + ((long long)b_kq0 * g.Np + n0 + b_n) * 4;
  constexpr int B_KQ_STEP = 256 / BN;     // kq advance per q
  const long long w_tile = (long long)8 * g.Np * 4;
  if (kb > 0) {                            // a later share of the reduction: the cursors start at K tile kb
    if (VEC) {
      const int ntap = a.TH * a.TW, cblk = a.Cin >> 5;
      const int tap = g.corder ? kb % ntap : kb / cblk;
      c0 = 32 * (g.corder ? kb / ntap : kb % cblk);
      tap_a = tap / a.TW; tap_b = tap - tap_a * a.TW;
    }
    if (!(VEC && g.corder)) wsrc += w_tile * kb;
  }

#define CY_LOAD_A_VEC(R, Q)                                                                        \
  {                                                                                                \
    const int iy = iy0[Q] + dy, ix = ix0[Q] + dx;                                                  \
    R = make_float4(0.f, 0.f, 0.f, 0.f);                                                           \
    if ((unsigned)iy < (unsigned)a.Hi && (unsigned)ix < (unsigned)a.Wi) R = *(const float4*)(a.X + (xo[Q] + toff)); \
  }
#define CY_LOAD_A_SCALAR(R, J)                                                                     \
  {                                                                                                \
    float v_[4];                                                                                   \
    _Pragma("unroll") for (int e = 0; e < 4; ++e) {                                                \
      const int ent = ktab[ktn * 32 + ((t >> 7) * 4 + J) * 4 + e];                                 \
      const int iy = iy0[0] + (int)(signed char)((ent >> 16) & 0xff);                              \
      const int ix = ix0[0] + (int)(signed char)((ent >> 24) & 0xff);                              \
      float x_ = 0.f;                                                                              \
      if (ent != -1 && (unsigned)iy < (unsigned)a.Hi && (unsigned)ix < (unsigned)a.Wi)             \
        x_ = a.X[xo[0] + (long long)iy * a.xs_y + (long long)ix * a.xs_x + (long long)(ent & 0xffff) * a.xs_c]; \
      v_[e] = x_;                                                                                  \
    }                                                                                              \
    R = make_float4(v_[0], v_[1], v_[2], v_[3]);                                                   \
  }

  f32x16 acc[2][NTW];
#pragma unroll
  for (int mi = 0; mi < 2; ++mi)
#pragma unroll
    for (int ni = 0; ni < NTW; ++ni)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.f;

  __syncthreads();                         // ktab / rowoff visible

  // kt = kb - 1 is the prologue (load + store the first tile, nothing to compute)
  for (int kt = kb - 1; kt < ke; ++kt) {
    const int ktn = kt + 1;
    const bool more = ktn < ke;
    if (more) {
      // B first: hipcc guards the B destination registers with a conservative vmcnt wait, which is free
      // while nothing is in flight and would otherwise drain the A loads issued just before it
      const float* wt = (VEC && g.corder) ? wsrc + w_tile * ((long long)(tap_a * a.TW + tap_b) * (a.Cin >> 5) + (c0 >> 5))
                                          : wsrc;
#pragma unroll
      for (int q = 0; q < NBQ; ++q) rb[q] = *(const f32x4*)(wt + (long long)q * B_KQ_STEP * g.Np * 4);
      if (!(VEC && g.corder)) wsrc += w_tile;
      if (VEC) {
        const int dy = tap_a * a.dstep, dx = tap_b * a.dstep;
        const long long toff = (long long)dy * a.xs_y + (long long)dx * a.xs_x + c0;
        CY_LOAD_A_VEC(ra0, 0) CY_LOAD_A_VEC(ra1, 1) CY_LOAD_A_VEC(ra2, 2) CY_LOAD_A_VEC(ra3, 3)
        if (g.corder) {
          if (++tap_b == a.TW) { tap_b = 0; if (++tap_a == a.TH) { tap_a = 0; c0 += 32; } }
        } else {
          c0 += 32;
          if (c0 >= a.Cin) { c0 = 0; if (++tap_b == a.TW) { tap_b = 0; ++tap_a; } }
        }
      } else {
        CY_LOAD_A_SCALAR(ra0, 0) CY_LOAD_A_SCALAR(ra1, 1) CY_LOAD_A_SCALAR(ra2, 2) CY_LOAD_A_SCALAR(ra3, 3)
      }
    }
    if (kt >= kb) {
      const int cur = kt & 1;
      const float* Ab = As + cur * A_BUF + (wave_m * 64 + li) * 4;
      const float* Bb = Bs + cur * B_BUF + (wave_n * 32 * NTW + li) * 4;
      // fragment registers are double-buffered: group kg+1 is fetched from LDS before group kg's MFMAs issue
      f32x4 fa[2][2], fb[2][NTW];
#pragma unroll
      for (int mi = 0; mi < 2; ++mi) fa[0][mi] = *(const f32x4*)(Ab + lh * A_KQ + mi * 128);
#pragma unroll
      for (int ni = 0; ni < NTW; ++ni) fb[0][ni] = *(const f32x4*)(Bb + lh * B_KQ + ni * 128);
#pragma unroll
      for (int kg = 0; kg < 4; ++kg) {
        const int cb = kg & 1, nb = cb ^ 1;
        if (kg < 3) {
          const int kq = 2 * (kg + 1) + lh;
#pragma unroll
          for (int mi = 0; mi < 2; ++mi) fa[nb][mi] = *(const f32x4*)(Ab + kq * A_KQ + mi * 128);
#pragma unroll
          for (int ni = 0; ni < NTW; ++ni) fb[nb][ni] = *(const f32x4*)(Bb + kq * B_KQ + ni * 128);
        }
        __builtin_amdgcn_sched_barrier(0);     // keep the prefetch ahead of this group's MFMAs
        // LDS reads return in order: this group's fragments are ready once all but the 2+NTW reads just
        // issued have landed (s_waitcnt lgkmcnt(N), vmcnt/expcnt untouched)
        if (kg < 3) __builtin_amdgcn_s_waitcnt(0xC07F | ((2 + NTW) << 8));
        else __builtin_amdgcn_s_waitcnt(0xC07F);
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
#pragma unroll
          for (int ni = 0; ni < NTW; ++ni) {
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[mi][ni] = mfma32(fa[cb][mi][e], fb[cb][ni][e], acc[mi][ni]);
          }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    if (more) {
      float* Ab = As + (ktn & 1) * A_BUF;
      float* Bb = Bs + (ktn & 1) * B_BUF;
      if (VEC) {
        float* dst = Ab + (t & 7) * A_KQ + (t >> 3) * 4;
        *(float4*)(dst) = ra0;
        *(float4*)(dst + 128) = ra1;
        *(float4*)(dst + 256) = ra2;
        *(float4*)(dst + 384) = ra3;
      } else {
        float* dst = Ab + (t >> 7) * 4 * A_KQ + (t & 127) * 4;
        *(float4*)(dst) = ra0;
        *(float4*)(dst + A_KQ) = ra1;
        *(float4*)(dst + 2 * A_KQ) = ra2;
        *(float4*)(dst + 3 * A_KQ) = ra3;
      }
#pragma unroll
      for (int q = 0; q < NBQ; ++q) *(f32x4*)(Bb + (b_kq0 + q * B_KQ_STEP) * B_KQ + b_n * 4) = rb[q];
    }
    __syncthreads();
  }
#undef CY_LOAD_A_VEC
#undef CY_LOAD_A_SCALAR

  // ---- epilogue: bias, activation, store, optional BatchNorm statistics.
  // The wave's 64 x (32 NTW) tile goes through its own slice of the (now idle) operand LDS so that the global stores
  // are 16 bytes per lane and 128-256 contiguous bytes per pixel row: 16 (8) store instructions per lane instead of
  // 64 (32) four-byte ones (conv_1, whose launch is nothing but its 2.8 GB of output, ran at 1.3 TB/s before).
  constexpr int WC = 32 * NTW;             // columns of the wave tile
  float* ow = smem + wave * (64 * WC);     // [64 rows][WC]; all waves are past the K loop's last barrier
  if (g.S > 1) {
    // split reduction: the raw partial sums of this share go to its slab (the padded columns too: Np floats per row, 16-byte stores)
#pragma unroll
    for (int ni = 0; ni < NTW; ++ni)
#pragma unroll
      for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int r = 0; r < 16; ++r) ow[(mi * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh) * WC + ni * 32 + li] = acc[mi][ni][r];
    constexpr int LPR = WC / 4, RPI = 64 / LPR;
    const int c4 = lane % LPR, rsub = lane / LPR;
    float* slab = g.ws + (long long)sp * g.M * g.Np + n0 + wave_n * WC + c4 * 4;
#pragma unroll
    for (int it = 0; it < 64 / RPI; ++it) {
      const int row_l = it * RPI + rsub;
      const long long m = m0 + wave_m * 64 + row_l;
      if (m < g.M) *(f32x4*)(slab + m * g.Np) = *(const f32x4*)(ow + row_l * WC + c4 * 4);
    }
    return;
  }
  float ssum[NTW], ssq[NTW];
#pragma unroll
  for (int ni = 0; ni < NTW; ++ni) {
    ssum[ni] = 0.f; ssq[ni] = 0.f;
    const int n = n0 + wave_n * WC + ni * 32 + li;
    const float bv = (a.bias != nullptr && n < a.N) ? a.bias[n] : 0.f;
#pragma unroll
    for (int mi = 0; mi < 2; ++mi) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row_l = mi * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
        float v = acc[mi][ni][r] + bv;
        if (a.stats != nullptr && n < a.N && rowoff[wave_m * 64 + row_l] >= 0) { ssum[ni] += v; ssq[ni] += v * v; }
        if (a.act == 1) v = fmaxf(v, 0.f);
        else if (a.act == 2) v = fmaxf(v, v * a.act_slope);       // LeakyReLU, 0 <= slope <= 1 (eval forward, BatchNorm folded)
        ow[row_l * WC + ni * 32 + li] = v;
      }
    }
  }
  {
    constexpr int LPR = WC / 4, RPI = 64 / LPR;          // lanes per row, rows per store instruction
    const int c4 = lane % LPR, rsub = lane / LPR;
    const int n = n0 + wave_n * WC + c4 * 4;
    const bool vec_ok = (a.N & 3) == 0 && (((uintptr_t)a.Y & 15) == 0);
    const bool bnb = a.bn_red != nullptr && vec_ok && n + 3 < a.N;      // BatchNorm-backward sums of the producer block
    f32x4 bsc = {0.f, 0.f, 0.f, 0.f}, bsh = bsc, bmu = bsc, bis = bsc, b1 = bsc, b2 = bsc;
    if (bnb) {
      bsc = *(const f32x4*)(a.bn_scale + n); bsh = *(const f32x4*)(a.bn_shift + n);
      bmu = *(const f32x4*)(a.bn_mean + n); bis = *(const f32x4*)(a.bn_invstd + n);
    }
    f32x4 zq_[64 / RPI];                   // all of this lane's bn_z loads are issued before the first is used
    if (bnb) {
#pragma unroll
      for (int it = 0; it < 64 / RPI; ++it) {
        const long long off = rowoff[wave_m * 64 + it * RPI + rsub];
        zq_[it] = *(const f32x4*)(a.bn_z + (off >= 0 ? off : 0) + n);
      }
    }
#pragma unroll
    for (int it = 0; it < 64 / RPI; ++it) {
      const int row_l = it * RPI + rsub;
      const long long off = rowoff[wave_m * 64 + row_l];
      const f32x4 v = *(const f32x4*)(ow + row_l * WC + c4 * 4);
      if (off >= 0) {
        if (vec_ok && n + 3 < a.N) *(f32x4*)(a.Y + off + n) = v;
        else {
#pragma unroll
          for (int k = 0; k < 4; ++k) if (n + k < a.N) a.Y[off + n + k] = v[k];
        }
        if (bnb) {
          const f32x4 zq = zq_[it];
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            const float y = zq[k] * bsc[k] + bsh[k];
            const float d = y > 0.f ? v[k] : v[k] * a.bn_slope;
            b1[k] += d;
            b2[k] += d * ((zq[k] - bmu[k]) * bis[k]);
          }
        }
      }
    }
    if (a.bn_red != nullptr) {             // lanes with the same channel quad (c4): add up the rows they hold
#pragma unroll
      for (int k = 0; k < 4; ++k) {
#pragma unroll
        for (int m = LPR; m < 64; m <<= 1) { b1[k] += __shfl_xor(b1[k], m, 64); b2[k] += __shfl_xor(b2[k], m, 64); }
      }
      __syncthreads();                     // every wave is done with its `ow` slice
      float* bred = smem;                  // [2 wave_m][BN][2]
      if (rsub == 0 && bnb) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const int col = wave_n * WC + c4 * 4 + k;
          bred[(wave_m * BN + col) * 2 + 0] = b1[k];
          bred[(wave_m * BN + col) * 2 + 1] = b2[k];
        }
      }
      __syncthreads();
      if (t < BN && n0 + t + 0 < a.N && vec_ok) {
        const int nq = n0 + (t & ~3);
        if (nq + 3 < a.N) {
          double* rd = a.bn_red + (size_t)(blockIdx.x % CY_STATS_COPIES) * a.N * 2;
          atomicAdd(rd + 2 * (n0 + t), (double)bred[t * 2] + (double)bred[(BN + t) * 2]);
          atomicAdd(rd + 2 * (n0 + t) + 1, (double)bred[t * 2 + 1] + (double)bred[(BN + t) * 2 + 1]);
        }
      }
    }
  }
  __syncthreads();                         // the statistics reduction below reuses the same LDS
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
      double* st = a.stats + (size_t)(blockIdx.x % CY_STATS_COPIES) * a.N * 2;
      atomicAdd(st + 2 * (n0 + t), s);
      atomicAdd(st + 2 * (n0 + t) + 1, q);
    }
  }
}

// ------------------------------------------------------------------------------------------------
// Split reduction, second launch: Y = act(sum over the S slabs (in split order: deterministic) + bias); one thread per (pixel, 4 channels).
__global__ void conv_gemm_splitk_finish(cy_conv_gemm_t a, GemmGeom g) {
  const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const int nq = g.Np / 4;
  if (idx >= g.M * nq) return;
  const long long m = idx / nq;
  const int n = (int)(idx - m * nq) * 4;
  if (n >= a.N) return;
  const float* p = g.ws + m * g.Np + n;
  f32x4 v = *(const f32x4*)p;
  for (int s_ = 1; s_ < g.S; ++s_) v += *(const f32x4*)(p + (long long)s_ * g.M * g.Np);
  const unsigned pp = (unsigned)m, HoWo = (unsigned)(a.Ho * a.Wo);
  const unsigned b = pp / HoWo, r = pp - b * HoWo;
  const int oy = (int)(r / (unsigned)a.Wo), ox = (int)(r - (unsigned)oy * (unsigned)a.Wo);
  float* y = a.Y + (((long long)b * a.Hy + (oy * a.out_stride + a.out_oy)) * a.Wy + (ox * a.out_stride + a.out_ox)) * a.N + n;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    if (n + k >= a.N) break;
    float x = v[k] + (a.bias != nullptr ? a.bias[n + k] : 0.f);
    if (a.act == 1) x = fmaxf(x, 0.f);
    else if (a.act == 2) x = fmaxf(x, x * a.act_slope);
    y[k] = x;
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
// The reduction dimension is the pixel.  A stage is PT = 32 pixels = 8 groups of 4; both operands are staged
// in LDS as [pixel group][e][u][4 pixels] (channel c = 4u + e), so that -- exactly as in the forward kernel --
// lane (i,h) fetches 4 MFMA steps with ONE ds_read_b128: group 2kg+h, step t multiplies pixels 4(2kg)+t and
// 4(2kg+1)+t.  The 4x4 (pixel x channel) transpose happens for free in registers: a thread loads the same 4
// channels of 4 consecutive pixels and writes one float4 per channel.  MFMA row i of a 32-row sub-tile is
// channel 4*(i%8) + i/8 of it; with an e-row stride of 8 (mod 16) slots both the ds_write_b128 (8 consecutive
// lanes -> 8 consecutive slots) and the ds_read_b128 (16-lane groups) are bank-conflict free.
constexpr int PT = 32;                     // pixels per pipeline stage
__host__ __device__ constexpr int wg_row_stride(int U) { return U + ((8 - (U % 16) + 16) % 16); }

template <int MT, int NT, int WM, int WN, bool VEC>
__global__ __launch_bounds__(256, 2) void conv_wgrad_kernel(cy_conv_wgrad_t a, int K, long long M, long long pix_per_split) {
  constexpr int BMK = 32 * MT * WM, BNN = 32 * NT * WN;
  constexpr int UA = BMK / 4, UB = BNN / 4;
  constexpr int RSA = wg_row_stride(UA), RSB = wg_row_stride(UB);     // slots (16 B) per e-row
  constexpr int A_BUF = 8 * 4 * RSA * 4, B_BUF = 8 * 4 * RSB * 4;      // floats per stage buffer
  constexpr int NIA = (8 * UA + 255) / 256, NIB = (8 * UB + 255) / 256; // items (pixel group x channel quad) per thread
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* As = smem;
  float* Zs = smem + 2 * A_BUF;
  int* ktab = (int*)(Zs + 2 * B_BUF);      // scalar loader: [BMK] {c | dy<<16 | dx<<24}

  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int wm = wave % WM, wn = wave / WM;
  const int li = lane & 31, lh = lane >> 5;
  // Blocks are dealt round-robin to the 8 XCDs in linear-id order.  Give each XCD a contiguous range of virtual ids
  // instead, so that the K/BMK x N/BNN blocks of one pixel split -- whose taps re-read the same input pixels (a
  // k4/s2 layer reads every input pixel through 4 taps) -- run on one XCD and share its L2 (conv_3: 25 -> HBM GB).
  unsigned bx = blockIdx.x, by = blockIdx.y, bz = blockIdx.z;
  {
    const unsigned nxy = gridDim.x * gridDim.y, total = nxy * gridDim.z;
    if ((total & 7u) == 0) {
      const unsigned lin = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);
      const unsigned vid = (lin & 7u) * (total >> 3) + (lin >> 3);
      bz = vid / nxy;
      const unsigned rem = vid - bz * nxy;
      by = rem / gridDim.x; bx = rem - by * gridDim.x;
    }
  }
  const int kk0 = bx * BMK, n0 = by * BNN;
  const long long p_begin = (long long)bz * pix_per_split;
  long long p_end = p_begin + pix_per_split;
  if (p_end > M) p_end = M;
  const int HoWo = a.Ho * a.Wo;

  // ---- A items of this thread: (pixel group pgA, channel quad uA); tap / channel are fixed per thread
  int a_dy[NIA], a_dx[NIA], a_c[NIA], a_pg[NIA], a_u[NIA];
  bool a_ok[NIA];
#pragma unroll
  for (int j = 0; j < NIA; ++j) {
    const int it = t + 256 * j;
    a_pg[j] = it / UA; a_u[j] = it % UA;
    const int kk = kk0 + a_u[j] * 4;
    a_ok[j] = (it < 8 * UA) && (kk < K);
    a_dy[j] = a_dx[j] = a_c[j] = 0;
    if (VEC) {
      const int tap = kk / a.Cin, c = kk - tap * a.Cin;
      const int kh = tap / a.KW, kw = tap - kh * a.KW;
      a_dy[j] = kh - a.pad; a_dx[j] = kw - a.pad; a_c[j] = c;
    }
  }
  if (!VEC) {
    for (int k = t; k < BMK; k += 256) {
      const int kk = kk0 + k;
      int ent = -1;
      if (kk < K) {
        const int tap = kk / a.Cin, c = kk - tap * a.Cin;
        const int kh = tap / a.KW, kw = tap - kh * a.KW;
        ent = (c & 0xffff) | (((kh - a.pad) & 0xff) << 16) | (((kw - a.pad) & 0xff) << 24);
      }
      ktab[k] = ent;
    }
    __syncthreads();
  }
  const bool zvec = ((a.N & 3) == 0) && (((uintptr_t)a.dZ & 15) == 0);

  // pixel cursor (batch index, in-image linear index) of pixel p_begin + 4*a_pg[0]; advanced by PT per stage.
  // Items j > 0 of a thread sit 256/UA groups further (same u).
  unsigned cur_b, cur_r;
  {
    const long long p0 = p_begin + 4 * a_pg[0];
    cur_b = (unsigned)(p0 / HoWo);
    cur_r = (unsigned)(p0 - (long long)cur_b * HoWo);
  }
  f32x4 ra[NIA][4], rb[NIB][4];            // [item][pixel] = 4 channels

  auto load_stage = [&](long long ps) {
#pragma unroll
    for (int j = 0; j < NIA; ++j) {
      unsigned b = cur_b, r = cur_r + 4 * (a_pg[j] - a_pg[0]);
      while (r >= (unsigned)HoWo) { r -= (unsigned)HoWo; ++b; }
      int oy = (int)(r / (unsigned)a.Wo), ox = (int)(r - (unsigned)oy * (unsigned)a.Wo);
      const long long pbase = ps + 4 * a_pg[j];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (a_ok[j] && pbase + q < p_end) {
          if (VEC) {
            const int iy = oy * a.stride + a_dy[j], ix = ox * a.stride + a_dx[j];
            if ((unsigned)iy < (unsigned)a.Hi && (unsigned)ix < (unsigned)a.Wi)
              v = *(const f32x4*)(a.X + ((long long)b * a.xs_b + (long long)iy * a.xs_y + (long long)ix * a.xs_x + a_c[j]));
          } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              const int ent = ktab[a_u[j] * 4 + e];
              if (ent != -1) {
                const int iy = oy * a.stride + (int)(signed char)((ent >> 16) & 0xff);
                const int ix = ox * a.stride + (int)(signed char)((ent >> 24) & 0xff);
                if ((unsigned)iy < (unsigned)a.Hi && (unsigned)ix < (unsigned)a.Wi)
                  v[e] = a.X[(long long)b * a.xs_b + (long long)iy * a.xs_y + (long long)ix * a.xs_x + (long long)(ent & 0xffff) * a.xs_c];
              }
            }
          }
        }
        ra[j][q] = v;
        if (++ox == a.Wo) { ox = 0; if (++oy == a.Ho) { oy = 0; ++b; } }
      }
    }
    cur_r += PT;
    while (cur_r >= (unsigned)HoWo) { cur_r -= (unsigned)HoWo; ++cur_b; }
#pragma unroll
    for (int j = 0; j < NIB; ++j) {
      const int it = t + 256 * j;
      const int pg = it / UB, u = it % UB;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const long long p = ps + 4 * pg + q;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (it < 8 * UB && p < p_end) {
          const float* src = a.dZ + (p * a.N + n0 + u * 4);
          if (zvec && n0 + u * 4 + 3 < a.N) v = *(const f32x4*)src;
          else {
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = (n0 + u * 4 + e < a.N) ? src[e] : 0.f;
          }
        }
        rb[j][q] = v;
      }
    }
  };
  auto store_stage = [&](int buf) {
    float* Ab = As + buf * A_BUF;
    float* Zb = Zs + buf * B_BUF;
#pragma unroll
    for (int j = 0; j < NIA; ++j) {
      if (t + 256 * j < 8 * UA) {
#pragma unroll
        for (int e = 0; e < 4; ++e)
          *(f32x4*)(Ab + ((a_pg[j] * 4 + e) * RSA + a_u[j]) * 4) = f32x4{ra[j][0][e], ra[j][1][e], ra[j][2][e], ra[j][3][e]};
      }
    }
#pragma unroll
    for (int j = 0; j < NIB; ++j) {
      const int it = t + 256 * j;
      if (it < 8 * UB) {
        const int pg = it / UB, u = it % UB;
#pragma unroll
        for (int e = 0; e < 4; ++e)
          *(f32x4*)(Zb + ((pg * 4 + e) * RSB + u) * 4) = f32x4{rb[j][0][e], rb[j][1][e], rb[j][2][e], rb[j][3][e]};
      }
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
  // fragment of MFMA row/col `li` of sub-tile s, pixel group pg: slot (pg*4 + li/8)*RS + 8*s + li%8
  const int fragA = ((li >> 3) * RSA + 8 * (wm * MT) + (li & 7)) * 4;
  const int fragB = ((li >> 3) * RSB + 8 * (wn * NT) + (li & 7)) * 4;
  for (long long st = 0; st < nstages; ++st) {
    const int cur = (int)(st & 1);
    if (st + 1 < nstages) load_stage(p_begin + (st + 1) * PT);
    const float* Ab = As + cur * A_BUF + fragA;
    const float* Zb = Zs + cur * B_BUF + fragB;
    f32x4 fa[2][MT], fb[2][NT];
#pragma unroll
    for (int mi = 0; mi < MT; ++mi) fa[0][mi] = *(const f32x4*)(Ab + (lh * 4 * RSA + 8 * mi) * 4);
#pragma unroll
    for (int ni = 0; ni < NT; ++ni) fb[0][ni] = *(const f32x4*)(Zb + (lh * 4 * RSB + 8 * ni) * 4);
#pragma unroll
    for (int kg = 0; kg < 4; ++kg) {
      const int cb = kg & 1, nb = cb ^ 1;
      if (kg < 3) {
        const int pg = 2 * (kg + 1) + lh;
#pragma unroll
        for (int mi = 0; mi < MT; ++mi) fa[nb][mi] = *(const f32x4*)(Ab + (pg * 4 * RSA + 8 * mi) * 4);
#pragma unroll
        for (int ni = 0; ni < NT; ++ni) fb[nb][ni] = *(const f32x4*)(Zb + (pg * 4 * RSB + 8 * ni) * 4);
      }
      __builtin_amdgcn_sched_barrier(0);
      if (kg < 3) __builtin_amdgcn_s_waitcnt(0xC07F | ((MT + NT) << 8));
      else __builtin_amdgcn_s_waitcnt(0xC07F);
#pragma unroll
      for (int mi = 0; mi < MT; ++mi)
#pragma unroll
        for (int ni = 0; ni < NT; ++ni) {
#pragma unroll
          for (int e = 0; e < 4; ++e) acc[mi][ni] = mfma32(fa[cb][mi][e], fb[cb][ni][e], acc[mi][ni]);
        }
      __builtin_amdgcn_sched_barrier(0);
    }
    if (st + 1 < nstages) store_stage(cur ^ 1);
    __syncthreads();
  }

  float* slab = a.slabs + (long long)bz * K * a.N;
#pragma unroll
  for (int mi = 0; mi < MT; ++mi)
#pragma unroll
    for (int ni = 0; ni < NT; ++ni) {
      const int n = n0 + (wn * NT + ni) * 32 + 4 * (li & 7) + (li >> 3);
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int ri = (r & 3) + 8 * (r >> 2) + 4 * lh;
        const int kk = kk0 + (wm * MT + mi) * 32 + 4 * (ri & 7) + (ri >> 3);
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

// The same for a SMALL gradient with many slabs (DarkNet conv_4: 8192 elements x 512 slabs -- one thread per element walked the 512
// slabs alone: 119 us for 16 MB): 16 elements x 16 slab groups per block; a thread adds every 16th slab, the groups are added in
// group order through LDS (a fixed order: deterministic).
__global__ __launch_bounds__(256) void wgrad_reduce_wide_kernel(const float* __restrict__ slabs, float* __restrict__ dW, int S, int K, int N,
                                                                int Cin, int KH, int KW) {
  __shared__ float part[16][17];
  const int t = threadIdx.x, e = t & 15, g = t >> 4;
  const long long total = (long long)K * N;
  const long long idx = (long long)blockIdx.x * 16 + e;
  float s = 0.f;
  if (idx < total)
    for (int i = g; i < S; i += 16) s += slabs[(long long)i * total + idx];
  part[g][e] = s;
  __syncthreads();
  if (g == 0 && idx < total) {
    float r = part[0][e];
#pragma unroll
    for (int k = 1; k < 16; ++k) r += part[k][e];
    const int n = (int)(idx % N);
    const int kk = (int)(idx / N);
    const int tap = kk / Cin, ci = kk - tap * Cin;
    const int kh = tap / KW, kw = tap - kh * KW;
    dW[(((long long)n * Cin + ci) * KH + kh) * KW + kw] = r;
  }
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

// Pixel splits of the weight-gradient reduction.  Blocks are long-running (thousands of MFMA stages) and only
// 2 fit a CU (80 KiB of LDS each), so the grid is sized to fill whole rounds of the 512 resident slots: a grid
// of 3.02 rounds costs 4.
int pick_splits(int tiles, long long M, long long slab_floats) {
  const long long capacity = 512;
  // a layer with a tiny gradient (CapsuleNet's decoder: K N = 288 .. 1152 floats) is a chain of load latencies, not of work: down to 128
  // pixels per split (its 8192-pixel layer ran 8 blocks of 1024 pixels for 0.55 ms of a 2.8 ms step)
  // (later in round 4: DarkNet's 1x1 layers at 26 x 26 / 13 x 13 -- 8 and 32 output tiles, 10816 / 2704 pixels -- ran 80 / 64 blocks
  // for 132 / 107 us against 18 us of MFMA time: the floor is 256 pixels (8 pipeline stages) per split, and what bounds the number of
  // splits of a large gradient is the slab traffic itself: at most 16 M floats of slabs, ~25 us of writes and reads)
  long long min_px = slab_floats < 8192 ? 128 : 256;
  long long cap = M / min_px;
  const long long cap_slab = (16ll << 20) / (slab_floats > 0 ? slab_floats : 1);
  if (cap > cap_slab) cap = cap_slab;
  if (cap < 1) cap = 1;
  long long best = 1;
  double best_util = 0.0;
  for (long long rounds = 1; rounds <= 4; ++rounds) {
    long long sp = capacity * rounds / tiles;
    if (sp < 1) sp = 1;
    if (sp > cap) sp = cap;
    const long long blocks = sp * tiles;
    const double util = (double)blocks / (double)(capacity * ((blocks + capacity - 1) / capacity));
    if (util > best_util + 0.02) { best_util = util; best = sp; }
  }
  if (best > 1024) best = 1024;
  return (int)best;
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
  p.S = pick_splits(p.ktiles * p.ntiles, M, (long long)K * a->N);
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

namespace {
// tile width and split of the reduction for one launch (shared by cy_conv_gemm and its workspace query)
struct GemmPlan { int ntw, S, kt_per; long long tiles; };
GemmPlan plan_gemm(const cy_conv_gemm_t* a, const GemmGeom& g) {
  GemmPlan p;
  p.ntw = (g.Np % 128 == 0) ? 2 : 1;
  const long long mtiles = cy_ceil_div(g.M, BM);
  // a grid far below the 256 CUs (CapsuleNet's primary-capsule convolution: 21 row tiles x 1 column tile of 128, 648 K
  // steps each) gets 64-column tiles: twice the blocks
  if (p.ntw == 2 && mtiles * (g.Np / 128) < 128) p.ntw = 1;
  p.tiles = mtiles * (g.Np / (64 * p.ntw));
  p.S = 1; p.kt_per = g.KT;
  // ... and, still far below two blocks per CU with a long reduction, S blocks per tile on a share of the K tiles each (>= 8 per
  // share; not with the fused statistics epilogues, which need the finished sums)
  int dev = 0, ncu = 256;
  if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || ncu <= 0)
    ncu = 256;
  if (a->stats == nullptr && a->bn_red == nullptr && p.tiles * 4 <= 3ll * ncu && g.KT >= 32) {
    long long S = (2ll * ncu) / p.tiles;
    if (S > g.KT / 8) S = g.KT / 8;
    if (S > 1) {
      p.kt_per = (int)cy_ceil_div(g.KT, S);
      p.S = (int)cy_ceil_div(g.KT, p.kt_per);
    }
  }
  return p;
}
GemmGeom gemm_geom(const cy_conv_gemm_t* a) {
  GemmGeom g{};
  g.K = a->TH * a->TW * a->Cin;
  g.KT = (g.K + 31) / 32;
  g.Np = (a->N + 63) / 64 * 64;
  g.M = (long long)a->B * a->Ho * a->Wo;
  g.corder = 1;        // measured on conv_3 (k4 s2, Cin 256): HBM fetch 22.4 GB per launch with taps outermost, -5 % time
  g.S = 1; g.kt_per = g.KT; g.tiles = 0; g.ws = nullptr;
  return g;
}
}  // namespace

extern "C" long long cy_conv_gemm_ws_floats(const cy_conv_gemm_t* a) {
  if (!a || a->B <= 0 || a->Ho <= 0 || a->Wo <= 0 || a->N <= 0 || a->Cin <= 0 || a->TH <= 0 || a->TW <= 0) return 0;
  const GemmGeom g = gemm_geom(a);
  const GemmPlan p = plan_gemm(a, g);
  return p.S > 1 ? (long long)p.S * g.M * g.Np : 0;
}

extern "C" int cy_conv_gemm(const cy_conv_gemm_t* a, void* stream) {
  CY_REQUIRE(a && a->X && a->Wp && a->Y, "cy_conv_gemm: null pointer");
  CY_REQUIRE(a->B > 0 && a->Ho > 0 && a->Wo > 0 && a->N > 0 && a->Cin > 0 && a->TH > 0 && a->TW > 0,
             "cy_conv_gemm: non-positive dimension");
  CY_REQUIRE(a->TH * a->TW <= 4096, "cy_conv_gemm: too many taps");
  CY_REQUIRE(a->act >= 0 && a->act <= 2 && (a->act != 2 || (a->act_slope >= 0.f && a->act_slope <= 1.f)),
             "cy_conv_gemm: act=%d / act_slope=%g (LeakyReLU slope must be in [0, 1])", a->act, (double)a->act_slope);
  CY_REQUIRE(a->bn_red == nullptr || (a->bn_z && a->bn_scale && a->bn_shift && a->bn_mean && a->bn_invstd && a->N % 4 == 0 &&
                                      (((uintptr_t)a->Y | (uintptr_t)a->bn_z) & 15) == 0),
             "cy_conv_gemm: bn_red needs bn_z / scale / shift / mean / invstd, N %% 4 == 0 and 16-byte aligned Y, bn_z");
  GemmGeom g = gemm_geom(a);
  const bool vec = (a->xs_c == 1) && (a->Cin % 32 == 0) && (a->xs_x % 4 == 0) && (a->xs_y % 4 == 0) &&
                   (a->xs_b % 4 == 0) && (((uintptr_t)a->X & 15) == 0);
  if (!vec) {
    CY_REQUIRE(g.KT * 32 <= 2048, "cy_conv_gemm: scalar loader supports K <= 2048 (got %d)", g.K);
    const int maxd = 127;
    CY_REQUIRE(abs(a->dy0) + a->TH * abs(a->dstep) <= maxd && abs(a->dx0) + a->TW * abs(a->dstep) <= maxd,
               "cy_conv_gemm: tap offsets out of range");
    CY_REQUIRE(a->Cin <= 65535, "cy_conv_gemm: scalar loader Cin too large");
  }
  const GemmPlan plan = plan_gemm(a, g);
  const int ntw = plan.ntw;
  const int BN = 64 * ntw;
  // the split of the reduction needs the caller's workspace (cy_conv_gemm_ws_floats); without one the layer runs unsplit
  if (plan.S > 1 && a->ws != nullptr) {
    CY_REQUIRE(a->ws_floats >= (long long)plan.S * g.M * g.Np && (((uintptr_t)a->ws) & 15) == 0,
               "cy_conv_gemm: ws holds %lld floats, the split reduction needs %lld (cy_conv_gemm_ws_floats), 16-byte aligned",
               a->ws_floats, (long long)plan.S * g.M * g.Np);
    g.S = plan.S; g.kt_per = plan.kt_per; g.tiles = (int)plan.tiles; g.ws = a->ws;
  }
  const long long nblocks = plan.tiles * g.S;
  CY_REQUIRE(nblocks < (1ll << 31) && g.M < (1ll << 31), "cy_conv_gemm: grid too large");
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
  if (g.S > 1) {
    conv_gemm_splitk_finish<<<(unsigned)cy_ceil_div(g.M * (g.Np / 4), 256), 256, 0, s>>>(*a, g);
    CY_LAUNCH_CHECK("cy_conv_gemm (split-reduction finish)");
  }
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
  size_t lds = (size_t)2 * 8 * 4 * 4 * (wg_row_stride(p.BMK / 4) + wg_row_stride(p.BNN / 4)) * 4 + (p.vec ? 0 : p.BMK * 4);
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
  if (total <= 65536 && p.S >= 64)
    wgrad_reduce_wide_kernel<<<(unsigned)cy_ceil_div(total, 16), 256, 0, s>>>(a->slabs, a->dW, p.S, K, a->N, a->Cin, a->KH, a->KW);
  else
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
