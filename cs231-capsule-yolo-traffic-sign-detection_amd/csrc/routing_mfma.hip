// Capsule routing for C > 1 output capsules with the prediction matmul u_hat = u W on the matrix cores (gfx950).
//
// Replaces, for row tiles of 16 capsule rows, the row-stationary pass of routing_rows.hip (models.py:70-79: u_hat = u W,
// r x {c = softmax_j(b), s = sum_i c u_hat, v = squash(s), b += u_hat . v}).  u_hat is still recomputed in every
// iteration (R*N*C*Dout floats cannot be kept), but as v_mfma_f32_16x16x4_f32 tiles instead of vector FMAs fed from LDS:
//
//   tile (jt, o) of input capsule i:   U_i [16 rows x 8]  x  W_i[:, 16 jt .. 16 jt + 15, :, o]^T [8 x 16 capsules]   (2 MFMAs, K = 4 + 4)
//
// The MFMA's N dimension is the OUTPUT CAPSULE j (not the flattened (j, o) column): lane l then holds u_hat[row 4 (l >> 4) + r]
// [capsule 16 jt + (l & 15)][o] for r = 0..3 -- every output component o of a (row, capsule) pair lives in the SAME lane, so the
// logit u_hat . V_t[j] and the weighted sum are lane-local, the softmax over j is a reduction over the 16 lanes of a DPP row
// (+ the JT tiles of the lane), and nothing is transposed.  Each W float is read from LDS ONCE per 16 rows (one register per
// lane and MFMA) where the vector kernel reads it once per 4 rows: the pass is no longer LDS-fed.
// One block = 4 waves = one wave per SIMD (the lane's state -- V_t, the running sums and u_hat of 4 rows x JT capsules x OW
// components, three arrays of 4 JT OW floats -- needs the 512-register budget).  The waves split the OUTPUT COMPONENTS:
// wave w owns o = 4 q + w, q < OW = ceil(Dout / 4) (a zero-padded slot where o >= Dout), for ALL capsules of the block's 16
// rows.  A logit is a sum over o, so per input capsule the waves exchange their partial logits through LDS (768 bytes per
// wave, one barrier per input capsule -- the same barrier that guards the double-buffered W image) and each wave then runs
// the same softmax on the same bits.  W is repacked once per call into the per-lane operand order
// [i][wave][jt][lane][q][s] (mfma_pack_w_kernel): a lane's operands of one capsule tile are 2 OW consecutive floats
// (ds_read_b128 at a lane stride of 48 / 32 bytes: conflict-free), and the image of W_i is one linear LDS-DMA copy.
#include "common.h"
#include <type_traits>

namespace {

constexpr int mf_ow(int dout) { return (dout + 3) / 4; }       // component slots per wave
constexpr int mf_lf(int dout) { return 2 * mf_ow(dout); }      // operand floats per (wave, capsule tile, lane): OW slots x 2 K-steps

// W [N][C][8][Dout] -> Wp [N][4 waves][JT][64 lanes][OW][2]:  lane l of wave w, tile jt, slot q, K-step s holds
// W[i][j = 16 jt + (l & 15)][k = 2 (l >> 4) + s][o = 4 q + w]   (0 where j >= C or o >= Dout)
__global__ __launch_bounds__(256) void mfma_pack_w_kernel(const float* __restrict__ W, float* __restrict__ Wp, long long total, int C,
                                                          int dout, int JT, int OW) {
  const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= total) return;
  const int s = (int)(idx & 1);
  long long r = idx >> 1;
  const int q = (int)(r % OW); r /= OW;
  const int l = (int)(r & 63); r >>= 6;
  const int jt = (int)(r % JT); r /= JT;
  const int w = (int)(r & 3);
  const long long i = r >> 2;
  const int j = 16 * jt + (l & 15), k = 2 * (l >> 4) + s, o = 4 * q + w;
  Wp[idx] = (j < C && o < dout) ? W[((i * C + j) * 8 + k) * dout + o] : 0.f;
}

__device__ __forceinline__ f32x4 row16_max4(f32x4 v) {
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    float x = v[r];
    x = fmaxf(x, dpp_get<0xB1, 0xF>(x, x));
    x = fmaxf(x, dpp_get<0x4E, 0xF>(x, x));
    x = fmaxf(x, dpp_get<0x141, 0xF>(x, x));
    x = fmaxf(x, dpp_get<0x140, 0xF>(x, x));
    v[r] = x;
  }
  return v;
}
__device__ __forceinline__ f32x4 row16_sum4(f32x4 v) {
#pragma unroll
  for (int r = 0; r < 4; ++r) v[r] = row16_sum(v[r]);
  return v;
}

// One block: 16 capsule rows (blockIdx.x), input capsules [i0, i1) (blockIdx.y chunks of a.ic; the whole range when fused).
template <int DOUT, int JT>
__global__ __launch_bounds__(256, 1) void caps_mfma_kernel(cyi_rows_args_t a) {
  constexpr int OW = mf_ow(DOUT), LF = mf_lf(DOUT), NB = LF / 4;
  static_assert(LF % 4 == 0, "operand floats per lane must be whole 16-byte reads");
  constexpr int TILE = 4 * JT * 64 * LF;             // floats of one W_i image
  constexpr int XW = 64 * 4 * JT;                    // floats one wave exchanges per step
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* Wl = smem;                                  // [2][TILE]
  float* X = smem + 2 * TILE;                        // [2][4 waves][64 lanes][JT][4 rows]
  const int t = threadIdx.x, lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int jlo = lane & 15, rg = lane >> 4;
  const int C = a.C, N = a.N, R = a.R, g = a.g;
  const int i0 = blockIdx.y * a.ic;
  const int i1 = min(N, i0 + a.ic);
  const float invC = 1.0f / (float)C;

  // ---- A operand: lane l supplies u[row 16 tile + (l & 15)][i][k = 2 (l >> 4) + s] (one 8-byte load per step)
  const int arow_raw = blockIdx.x * 16 + jlo;
  const int arow = arow_raw < R ? arow_raw : R - 1;
  long long ubase;
  if (g) {
    const int kc = arow / a.B, b = arow - kc * a.B;
    ubase = ((long long)b * 16 * g * g + 4 * kc) * 256;
  } else {
    ubase = (long long)arow * N * 8;
  }
  const float* up = a.u + ubase + 2 * rg;
  auto uoff = [&](int i) -> long long {
    return g ? (long long)((i >> 7) * 4 * g * g + ((i >> 5) & 3)) * 256 + (i & 31) * 8 : (long long)i * 8;
  };
  // ---- D rows of this lane: row0 + r, r = 0..3; capsules 16 jt + jlo
  const int row0 = blockIdx.x * 16 + 4 * rg;
  bool jv[JT];
#pragma unroll
  for (int jt = 0; jt < JT; ++jt) jv[jt] = 16 * jt + jlo < C;

  auto stage = [&](int i, int buf) {                 // image of W_i -> LDS by LDS-DMA: TILE / 256 pieces of 1 KiB, wave w takes every 4th
    const float* src = a.Wp + (long long)i * TILE;
    float* dstb = Wl + buf * TILE;
#pragma unroll
    for (int p = 0; p < TILE / 1024; ++p) {
      const int c0 = (p * 4 + wave) * 256;           // floats
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + c0 + lane * 4),
                                       (__attribute__((address_space(3))) void*)(dstb + c0), 16, 0, 0);
    }
  };
  static_assert(TILE % 1024 == 0, "the W image must be whole 1 KiB pieces per wave round");

  f32x4 SV[JT][OW], S[JT][OW];                       // V_t and the running sums of (4 rows) x (capsule tile jt) x (slot q)
  auto zero = [](f32x4 (&P)[JT][OW]) {
#pragma unroll
    for (int jt = 0; jt < JT; ++jt)
#pragma unroll
      for (int q = 0; q < OW; ++q) P[jt][q] = f32x4{0.f, 0.f, 0.f, 0.f};
  };

  // sum of the four waves' f32x4[JT] contributions, in wave order (every wave adds the same bits in the same order)
  auto exchange = [&](const f32x4 (&mine)[JT], f32x4 (&tot)[JT], int par) {
    float* xb = X + par * 4 * XW;
#pragma unroll
    for (int jt = 0; jt < JT; ++jt) *(f32x4*)(xb + wave * XW + (lane * JT + jt) * 4) = mine[jt];
    __builtin_amdgcn_s_waitcnt(0x0F70);             // vmcnt(0): this wave's DMA pieces of the next W image and its next u have landed
    __syncthreads();
#pragma unroll
    for (int jt = 0; jt < JT; ++jt) {
      f32x4 acc = *(const f32x4*)(xb + 0 * XW + (lane * JT + jt) * 4);
#pragma unroll
      for (int w = 1; w < 4; ++w) acc += *(const f32x4*)(xb + w * XW + (lane * JT + jt) * 4);
      tot[jt] = acc;
    }
  };

  // one pass over [i0, i1): S += sum_i c_i u_hat_i
  auto run_pass = [&](auto uni_tag) {
    constexpr bool UNI = decltype(uni_tag)::value;
    stage(i0, 0);
    f32x2 ua = *(const f32x2*)(up + uoff(i0));
    f32x2 un = ua;
    if (i0 + 1 < i1) { stage(i0 + 1, 1); un = *(const f32x2*)(up + uoff(i0 + 1)); }
    __builtin_amdgcn_s_waitcnt(0x0F70);
    __syncthreads();
    for (int i = i0; i < i1; ++i) {
      const int cur = (i - i0) & 1;
      const float* wl = Wl + cur * TILE + ((wave * JT) * 64 + lane) * LF;
      // ---- u_hat tiles: all first K-steps, then all second ones (a dependent MFMA pair needs 40 cycles between its halves)
      f32x4 uh[JT][OW];
      f32x4 bq[JT][NB];
#pragma unroll
      for (int jt = 0; jt < JT; ++jt)
#pragma unroll
        for (int p = 0; p < NB; ++p) bq[jt][p] = *(const f32x4*)(wl + jt * 64 * LF + 4 * p);
#pragma unroll
      for (int jt = 0; jt < JT; ++jt)
#pragma unroll
        for (int q = 0; q < OW; ++q)
          uh[jt][q] = __builtin_amdgcn_mfma_f32_16x16x4f32(ua[0], bq[jt][(2 * q) >> 2][(2 * q) & 3], f32x4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
#pragma unroll
      for (int jt = 0; jt < JT; ++jt)
#pragma unroll
        for (int q = 0; q < OW; ++q)
          uh[jt][q] = __builtin_amdgcn_mfma_f32_16x16x4f32(ua[1], bq[jt][(2 * q + 1) >> 2][(2 * q + 1) & 3], uh[jt][q], 0, 0, 0);
      f32x4 c[JT];
      if constexpr (UNI) {
        __builtin_amdgcn_s_waitcnt(0x0F70);
        __syncthreads();                             // every wave has read W image `cur`; image cur ^ 1 has landed everywhere
#pragma unroll
        for (int jt = 0; jt < JT; ++jt) { const float cv = jv[jt] ? invC : 0.f; c[jt] = f32x4{cv, cv, cv, cv}; }
      } else {
        f32x4 pl[JT], b[JT];
#pragma unroll
        for (int jt = 0; jt < JT; ++jt) {
          f32x4 acc = uh[jt][0] * SV[jt][0];
#pragma unroll
          for (int q = 1; q < OW; ++q) acc = uh[jt][q] * SV[jt][q] + acc;
          pl[jt] = acc;
        }
        exchange(pl, b, cur);                        // (contains the step's barrier)
        f32x4 m = f32x4{-INFINITY, -INFINITY, -INFINITY, -INFINITY};
#pragma unroll
        for (int jt = 0; jt < JT; ++jt) {
#pragma unroll
          for (int r = 0; r < 4; ++r) { b[jt][r] = jv[jt] ? b[jt][r] : -INFINITY; m[r] = fmaxf(m[r], b[jt][r]); }
        }
        m = row16_max4(m);
        f32x4 z = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int jt = 0; jt < JT; ++jt) {
#pragma unroll
          for (int r = 0; r < 4; ++r) c[jt][r] = __expf(b[jt][r] - m[r]);
          z += c[jt];
        }
        z = row16_sum4(z);
#pragma unroll
        for (int r = 0; r < 4; ++r) z[r] = 1.0f / z[r];
#pragma unroll
        for (int jt = 0; jt < JT; ++jt) c[jt] *= z;
      }
      // the barrier of this step is behind us: W image `cur` is free, image cur ^ 1 and the next u are complete
      ua = un;
      if (i + 2 < i1) { stage(i + 2, cur); un = *(const f32x2*)(up + uoff(i + 2)); }
#pragma unroll
      for (int jt = 0; jt < JT; ++jt)
#pragma unroll
        for (int q = 0; q < OW; ++q) S[jt][q] = c[jt] * uh[jt][q] + S[jt][q];
    }
    __builtin_amdgcn_s_waitcnt(0x0F70);
    __syncthreads();                                 // nobody still reads the exchange buffers or a W image when the next pass starts
  };
  using UniT = std::integral_constant<bool, true>;
  using SmT = std::integral_constant<bool, false>;
  const long long CD = (long long)C * DOUT;
  const long long plane = (long long)R * CD;

  if (!a.fused) {
    // ---- one iteration's partial sums over [i0, i1) -> slab[chunk] (few rows: the input capsules are split over blocks)
    zero(S);
    if (a.it > 0) {
#pragma unroll
      for (int jt = 0; jt < JT; ++jt)
#pragma unroll
        for (int q = 0; q < OW; ++q)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int row = row0 + r, j = 16 * jt + jlo, o = 4 * q + wave;
            SV[jt][q][r] = (row < R && j < C && o < DOUT) ? a.V[((long long)row * C + j) * DOUT + o] : 0.f;
          }
      run_pass(SmT{});
    } else {
      zero(SV);
      run_pass(UniT{});
    }
    float* sl = a.slab + (long long)blockIdx.y * plane;
#pragma unroll
    for (int jt = 0; jt < JT; ++jt)
#pragma unroll
      for (int q = 0; q < OW; ++q)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int row = row0 + r, j = 16 * jt + jlo, o = 4 * q + wave;
          if (row < R && j < C && o < DOUT) sl[((long long)row * C + j) * DOUT + o] = S[jt][q][r];
        }
    return;
  }

  // ---- all iterations for this block's 16 rows in one launch
  zero(SV);
  for (int it = 0; it < a.n_iter; ++it) {
    zero(S);
    if (it == 0) run_pass(UniT{}); else run_pass(SmT{});
    const bool last = it == a.n_iter - 1;
    // squash: |s|^2 over ALL components = sum over the four waves' slots
    f32x4 n2p[JT], n2[JT];
#pragma unroll
    for (int jt = 0; jt < JT; ++jt) {
      f32x4 acc = S[jt][0] * S[jt][0];
#pragma unroll
      for (int q = 1; q < OW; ++q) acc = S[jt][q] * S[jt][q] + acc;
      n2p[jt] = acc;
    }
    exchange(n2p, n2, 0);
#pragma unroll
    for (int jt = 0; jt < JT; ++jt) {
      f32x4 f;
#pragma unroll
      for (int r = 0; r < 4; ++r) f[r] = (n2[jt][r] / (1.f + n2[jt][r])) / sqrtf(n2[jt][r]);   // no epsilon: 0 -> NaN like the reference (models.py:64-67)
      const int j = 16 * jt + jlo;
#pragma unroll
      for (int q = 0; q < OW; ++q) {
        const int o = 4 * q + wave;
        const f32x4 vv = f * S[jt][q];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int row = row0 + r;
          const bool ok = row < R && j < C && o < DOUT;
          SV[jt][q][r] += ok ? vv[r] : 0.f;          // (lanes / slots without a capsule keep V = 0: no NaN enters the MFMA side)
          if (ok) {
            a.s_hist[(long long)it * plane + ((long long)row * C + j) * DOUT + o] = S[jt][q][r];
            if (last) {
              long long orow = row;
              if (g) { const int kc = row / a.B, b = row - kc * a.B; orow = (long long)b * g * g + kc; }
              a.v_out[(orow * C + j) * DOUT + o] = vv[r];
            }
          }
        }
      }
    }
    __syncthreads();                                 // the exchange buffer is rewritten by the next pass
  }
}

template <int DOUT, int JT>
int launch_jt(const cyi_rows_args_t* a, int row_tiles, int nch, hipStream_t s) {
  constexpr int LF = mf_lf(DOUT);
  const size_t lds = (size_t)(2 * 4 * JT * 64 * LF + 2 * 4 * 64 * 4 * JT) * 4;
  int rc = cy_allow_lds(caps_mfma_kernel<DOUT, JT>, lds);
  if (rc) return rc;
  caps_mfma_kernel<DOUT, JT><<<dim3(row_tiles, a->fused ? 1 : nch), 256, lds, s>>>(*a);
  return 0;
}
template <int DOUT>
int launch_dout(const cyi_rows_args_t* a, int row_tiles, int nch, hipStream_t s) {
  const int JT = (a->C + 15) / 16;
  if (JT == 1) return launch_jt<DOUT, 1>(a, row_tiles, nch, s);
  if (JT == 2) return launch_jt<DOUT, 2>(a, row_tiles, nch, s);
  if (JT == 3) return launch_jt<DOUT, 3>(a, row_tiles, nch, s);
  return cy_set_error(CY_EINVAL, "routing mfma: C=%d needs more than 3 capsule tiles", a->C);
}

}  // namespace

bool cyi_mfma_ok(int C, int Dout) { return C > 1 && C <= 48 && (Dout == 16 || Dout == 21); }

long long cyi_mfma_wp_floats(int N, int C, int Dout) { return (long long)N * 4 * ((C + 15) / 16) * 64 * mf_lf(Dout); }

int cyi_mfma_pack_w(const float* W, float* Wp, int N, int C, int Dout, hipStream_t s) {
  const long long total = cyi_mfma_wp_floats(N, C, Dout);
  mfma_pack_w_kernel<<<(unsigned)cy_ceil_div(total, 256), 256, 0, s>>>(W, Wp, total, C, Dout, (C + 15) / 16, mf_ow(Dout));
  return 0;
}

// forward pass(es) on the MFMA kernel: fused (all iterations, grid = row tiles) or one phased iteration (grid = row tiles x nch chunks of a->ic)
int cyi_mfma_launch(const cyi_rows_args_t* a, int nch, int Dout, hipStream_t s) {
  if (a->Wp == nullptr) return cy_set_error(CY_EINVAL, "routing mfma: the packed W image is missing");
  const int row_tiles = (a->R + 15) / 16;
  if (Dout == 16) return launch_dout<16>(a, row_tiles, nch, s);
  if (Dout == 21) return launch_dout<21>(a, row_tiles, nch, s);
  return cy_set_error(CY_EINVAL, "routing mfma: Dout=%d is not built (16, 21)", Dout);
}
