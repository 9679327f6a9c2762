// Capsule routing for C > 1 output capsules: the row-stationary pass (gfx950).
//
// One pass = one routing iteration's sum over the input capsules i for a tile of rows (models.py:74-79):
//   forward  (MODE 0):  s^t_j  = sum_i c^t_ij u_hat_ij,            c^t = softmax_j(u_hat_ij . V_t[j])   (uniform for t = 0)
//   backward (MODE 1):  A_t[j] = sum_i db^t_ij u_hat_ij,           db = c (dc - sum_j c dc),  dc_ij = u_hat_ij . ds^t_j
// with u_hat_ij = u_i W_ij recomputed on the fly (R*N*C*Dout floats cannot be kept: 10 GB at the DarkCapsuleNet3 head),
// V_t = sum_{tau<t} v^tau (the logits identity b^t_ij = u_hat_ij . V_t[j], SURVEY F9: logits are never stored).
//
// The pass is fp32 vector work fed from LDS (8*Dout FMAs per (row, i, j) for the prediction alone; fp32 MFMA runs at
// the vector rate on this chip and shares its ALUs, so there is nothing to gain from it here).  Measured on this chip
// (tools/probe/valu_rate.hip, lds_fma.hip): ONE wave per SIMD issuing v_pk_fma_f32 reaches the practical fp32 peak
// (114 of ~120 TFLOP/s; unpacked v_fma_f32 needs two waves and stops at 105), a ds_read_b128 costs the CU's LDS array 4
// cycles whatever it broadcasts, and a wave does not overlap its own LDS reads with its own FMAs -- a second wave
// on the SIMD does.  Hence:
//  * a 16-lane DPP row <-> 16 output capsules j (NJ = ceil(C/16) capsules per lane), the four DPP rows of a wave
//    <-> four rows; the softmax over j is a lane-local maximum / sum over NJ values plus FOUR DPP steps, not a
//    wavefront reduction (C > 48: one j per lane, wavefront reductions);
//  * every FMA is a v_pk_fma_f32 over a PAIR of output components (o, o+1): the prediction (W pair from LDS times u_d
//    broadcast through op_sel), the logit dot products and the weighted sums.  Odd Dout is padded to an even DP with
//    a zero column, so that every row d of W_ij starts pair-aligned;
//  * the lane keeps V_t[j], the running sums and u_hat of its capsules in registers as pairs (up to 256 VGPRs: two
//    waves per SIMD, so that one wave's LDS reads hide behind the other's FMAs; 512 with one wave where the state
//    does not fit); the squash / squash-backward between two iterations is lane-local;
//  * W is repacked once per call into the LDS image layout [N][C][8*DP + 4] (rows_pack_w_kernel: zero pad column, 4
//    pad floats per capsule so that 16 capsules' 16-byte reads cover the 64 banks); a tile W_i is then ONE linear
//    LDS-DMA copy (global_load_lds_dwordx4: no staging registers, no address arithmetic in the loop) into a
//    double-buffered LDS image shared by up to 8 waves, one barrier per input capsule; the rows' u_i (8 floats) are
//    prefetched one step ahead.  The cell gather of models.py:393-398 is folded into the u address;
//  * the W_ij reads are inline-asm ds_read_b128 with counted waits, PF reads ahead of the FMAs that consume them.
// Many rows (DarkCapsuleNet3 head, R = 5408): ONE launch runs all iterations for a block's rows (fused = 1); the
// block size (waves) is chosen so that the row tiles spread over the 256 CUs.
// Few rows (CapsuleNet head, R = batch): the input capsules are split over blocks as well and one launch computes
// one iteration's partial sums (a grid-wide dependency per iteration; a kernel boundary costs ~1.7 us on this chip,
// an in-kernel grid barrier 5-7 us -- MI355X_MICROARCH.md, barrier-xcd -- so the boundary is the cheaper sync).
#include "common.h"
#include <type_traits>

namespace {

// developer knob for timing experiments (results are wrong when set): 1 no W staging after the first tile, 2 no barrier,
// 4 no u prefetch, 8 no LDS reads after the first ring
#ifndef CY_ROWS_DBG
#define CY_ROWS_DBG 0
#endif
constexpr int DBG = CY_ROWS_DBG;

__device__ __forceinline__ float row16_max(float v) {
  v = fmaxf(v, dpp_get<0xB1, 0xF>(v, v));
  v = fmaxf(v, dpp_get<0x4E, 0xF>(v, v));
  v = fmaxf(v, dpp_get<0x141, 0xF>(v, v));
  v = fmaxf(v, dpp_get<0x140, 0xF>(v, v));
  return v;
}
template <int SLOTS> __device__ __forceinline__ float grp_max(float v) {
  if constexpr (SLOTS == 16) return row16_max(v); else return wave_max(v);
}
template <int SLOTS> __device__ __forceinline__ float grp_sum(float v) {
  if constexpr (SLOTS == 16) return row16_sum(v); else return wave_sum(v);
}

template <int D> __device__ __forceinline__ void squash_v(const float (&s)[D], float (&v)[D]) {
  float n2 = 0.f;
#pragma unroll
  for (int o = 0; o < D; ++o) n2 += s[o] * s[o];
  const float f = (n2 / (1.f + n2)) / sqrtf(n2);          // no epsilon: 0 -> NaN like the reference (models.py:64-67)
#pragma unroll
  for (int o = 0; o < D; ++o) v[o] = f * s[o];
}
template <int D> __device__ __forceinline__ void squash_bwd_v(const float (&s)[D], const float (&dv)[D], float (&ds)[D]) {
  float n2 = 0.f, sd = 0.f;
#pragma unroll
  for (int o = 0; o < D; ++o) { n2 += s[o] * s[o]; sd += s[o] * dv[o]; }
  const float n = sqrtf(n2);
  const float h = n / (1.f + n2);
  const float hp = (1.f - n2) / ((1.f + n2) * (1.f + n2));
  const float k = sd * hp / n;
#pragma unroll
  for (int o = 0; o < D; ++o) ds[o] = h * dv[o] + k * s[o];
}

constexpr int rows_dp(int dout) { return dout + (dout & 1); }
constexpr int rows_ws(int dout) { return 8 * rows_dp(dout) + 4; }        // floats per capsule in the W image

// W [N][C][8][Dout] -> the LDS image layout [N][C][WS]: rows d padded to DP floats (zero), 4 pad floats per capsule
__global__ __launch_bounds__(256) void rows_pack_w_kernel(const float* __restrict__ W, float* __restrict__ Wp, long long NC,
                                                          int dout, int dp, int ws4) {
  const long long q = (long long)blockIdx.x * 256 + threadIdx.x;          // one float4 of the image
  if (q >= NC * ws4) return;
  const long long ij = q / ws4;
  const int c4 = (int)(q - ij * ws4);
  f32x4 v;
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const int f = c4 * 4 + e, d = f / dp, o = f - d * dp;
    v[e] = (d < 8 && o < dout) ? W[(ij * 8 + d) * dout + o] : 0.f;
  }
  ((f32x4*)Wp)[q] = v;
}

// WPS = waves per SIMD the kernel is built for: 2 (blocks of up to 8 waves, 256 registers) when the lane's state fits,
// else 1 (blocks of up to 4 waves, 512 registers)
template <int DOUT, int SLOTS, int NJ, int MODE, int WPS>
__global__ __launch_bounds__(256 * WPS, 1) void caps_rows_kernel(cyi_rows_args_t a) {
  constexpr int DP = rows_dp(DOUT), HP = DP / 2, DD = 8 * DP, WS = DD + 4, DD4 = DD / 4, RPW = 64 / SLOTS;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int t = threadIdx.x, lane = t & 63, nthreads = blockDim.x;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int slot = lane % SLOTS, rsub = lane / SLOTS;
  const int C = a.C, N = a.N, R = a.R, g = a.g;
  const int tile = C * WS;                          // floats per W_i image (a multiple of 4)
  const int n4 = tile >> 2;
  const int i0 = blockIdx.y * a.ic;
  const int i1 = min(N, i0 + a.ic);
  const float invC = 1.0f / (float)C;

  int jk[NJ];
  bool jv[NJ];
#pragma unroll
  for (int k = 0; k < NJ; ++k) {
    const int j = slot + SLOTS * k;
    jv[k] = j < C;
    // lanes past C read the slot's previous capsule: same bank as their neighbours expect (capsule 0 would collide with
    // the lane of slot 0, a two-way LDS conflict on every read of the last k)
    jk[k] = jv[k] ? j : (k > 0 ? j - SLOTS : 0);
  }
  const int row = (blockIdx.x * (nthreads >> 6) + wave) * RPW + rsub;
  const bool rv = row < R;
  const int rowi = rv ? row : R - 1;                // rows past the end compute on the last row and store nothing
  long long ubase, orow;
  if (g) {
    const int kc = rowi / a.B, b = rowi - kc * a.B;
    ubase = ((long long)b * 16 * g * g + 4 * kc) * 256;
    orow = (long long)b * g * g + kc;
  } else {
    ubase = (long long)rowi * N * 8;
    orow = rowi;
  }
  auto uoff = [&](int i) -> long long {
    return g ? (long long)((i >> 7) * 4 * g * g + ((i >> 5) & 3)) * 256 + (i & 31) * 8 : (long long)i * 8;
  };
  auto stage = [&](int i, int buf) {                // image of W_i -> LDS by LDS-DMA: a linear copy, 1 KiB per wave and round
    const float* src = a.Wp + (long long)i * tile;
    float* dstb = smem + buf * tile;
    const int per = (nthreads >> 6) * 64;
    int c0 = wave * 64;
    for (; c0 + 64 <= n4; c0 += per)                // whole 1 KiB pieces: no exec mask, no branch around the DMA
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + (c0 + lane) * 4),
                                       (__attribute__((address_space(3))) void*)(dstb + c0 * 4), 16, 0, 0);
    if (c0 < n4 && c0 + lane < n4)                  // the image's last, partial piece (one wave)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + (c0 + lane) * 4),
                                       (__attribute__((address_space(3))) void*)(dstb + c0 * 4), 16, 0, 0);
  };
  f32x4 raw[2];
  auto load_u = [&](int i) {
    const f32x4* p = (const f32x4*)(a.u + ubase + uoff(i));
    raw[0] = p[0];
    raw[1] = p[1];
  };

  float* cdb_t = nullptr;                            // MODE 1, fused: where this pass stores its couplings ([row][i][2][C]) or NULL
  // one pass over the block's input capsules; UNI: uniform coupling 1/C (first iteration: V = 0)
  auto run_pass = [&](auto uni_tag, const f32x2 (&V)[NJ][HP], const f32x2 (&DS)[MODE == 1 ? NJ : 1][HP], f32x2 (&ACC)[NJ][HP]) {
    constexpr bool UNI = decltype(uni_tag)::value;
    stage(i0, 0);
    load_u(i0);
    __builtin_amdgcn_s_waitcnt(0x0F70);             // vmcnt(0)
    __syncthreads();
    for (int i = i0; i < i1; ++i) {
      const int cur = (i - i0) & 1;
      float uv[8];
#pragma unroll
      for (int d = 0; d < 8; ++d) uv[d] = raw[d >> 2][d & 3];
      if (i + 1 < i1) {
        if constexpr (!(DBG & 1)) stage(i + 1, cur ^ 1);
        if constexpr (!(DBG & 4)) load_u(i + 1);
      }
      const float* tb = smem + cur * tile;
      f32x2 uh[NJ][HP];
      float b[NJ], dc[NJ];
      // u_hat of the lane's NJ capsules.  The DD4 float4 reads of a capsule's W_ij are inline asm with counted waits,
      // kept PF reads ahead of the FMAs that consume them: left to itself hipcc sinks every ds_read_b128 next to its
      // use and waits lgkmcnt(0) right behind it (an exposed LDS latency per read).
      // ONE wave per SIMD (WPS = 1): four reads per wait -- a tied wait is an issue slot, and hipcc pads a wait state (s_nop, another slot)
      // between an asm statement that defines vector registers and the next vector instruction (whatever stands in between): per
      // float4 that was 5 slots for 2 packed FMAs (DarkCapsuleNet3 head, row part of the backward: 5.5 -> 4.9 ms with pairs).  Two waves per SIMD fill each other's
      // slots: there pairs with only FOUR reads in flight are the best of the grid (reads per wait, reads in flight) = (1,3) 3.69, (1,4) 3.58,
      // (1,5) 3.58, (2,4) **3.45**, (2,6) 3.91, (2,8) 4.32 ms on the same head's forward: deeper rings cost registers at 256 per wave.
#ifndef CY_ROWS_ST2
#define CY_ROWS_ST2 2
#endif
#ifndef CY_ROWS_PF2
#define CY_ROWS_PF2 4
#endif
      constexpr int ST = WPS == 1 ? 4 : CY_ROWS_ST2;    // reads per tied wait
#ifndef CY_ROWS_PF1
#define CY_ROWS_PF1 8
#endif
      constexpr int PF = WPS == 1 ? ((DD4 < CY_ROWS_PF1) ? DD4 : CY_ROWS_PF1) : ((DD4 < CY_ROWS_PF2) ? DD4 : CY_ROWS_PF2); // reads in flight
      static_assert(DD4 % 4 == 0 && PF % ST == 0, "the W image is read in groups of four float4");
#pragma unroll
      for (int k = 0; k < NJ; ++k) {
        const unsigned wa = (unsigned)(uintptr_t)(__attribute__((address_space(3))) const float*)(tb + jk[k] * WS);
        f32x4 wq[PF];
#pragma unroll
        for (int p = 0; p < PF; ++p) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(wq[p]) : "v"(wa), "n"(16 * p));
#pragma unroll
        for (int h = 0; h < HP; ++h) uh[k][h] = f32x2{0.f, 0.f};
#pragma unroll
        for (int q = 0; q < DD4; q += ST) {
          // reads q .. min(q + PF, DD4) - 1 are in flight: wait for all but those younger than q + ST - 1
          const int younger = (q + PF <= DD4 ? PF : DD4 - q) - ST;
          f32x4& w0 = wq[q % PF];
          if constexpr (ST == 4) {
            f32x4& w1 = wq[(q + 1) % PF];
            f32x4& w2 = wq[(q + 2) % PF];
            f32x4& w3 = wq[(q + 3) % PF];
            switch (younger) {
              case 8: asm volatile("s_waitcnt lgkmcnt(8)" : "+v"(w0), "+v"(w1), "+v"(w2), "+v"(w3)); break;
              case 4: asm volatile("s_waitcnt lgkmcnt(4)" : "+v"(w0), "+v"(w1), "+v"(w2), "+v"(w3)); break;
              default: asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(w0), "+v"(w1), "+v"(w2), "+v"(w3)); break;
            }
          } else if constexpr (ST == 2) {
            f32x4& w1 = wq[(q + 1) % PF];
            switch (younger) {
              case 6: asm volatile("s_waitcnt lgkmcnt(6)" : "+v"(w0), "+v"(w1)); break;
              case 4: asm volatile("s_waitcnt lgkmcnt(4)" : "+v"(w0), "+v"(w1)); break;
              case 2: asm volatile("s_waitcnt lgkmcnt(2)" : "+v"(w0), "+v"(w1)); break;
              default: asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(w0), "+v"(w1)); break;
            }
          } else {
            switch (younger) {
              case 4: asm volatile("s_waitcnt lgkmcnt(4)" : "+v"(w0)); break;
              case 3: asm volatile("s_waitcnt lgkmcnt(3)" : "+v"(w0)); break;
              case 2: asm volatile("s_waitcnt lgkmcnt(2)" : "+v"(w0)); break;
              case 1: asm volatile("s_waitcnt lgkmcnt(1)" : "+v"(w0)); break;
              default: asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(w0)); break;
            }
          }
#pragma unroll
          for (int e = 0; e < 4 * ST; e += 2) {
            const int f = 4 * q + e, d = f / DP, h = (f % DP) / 2;          // DP is even: a pair never straddles two rows d
            const f32x4& w = wq[(q + (e >> 2)) % PF];
            uh[k][h] = f32x2{w[e & 3], w[(e & 3) + 1]} * f32x2{uv[d], uv[d]} + uh[k][h];
          }
          if (q + PF < DD4 && !(DBG & 8)) {
#pragma unroll
            for (int r = 0; r < ST; ++r)
              asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(wq[(q + r) % PF]) : "v"(wa), "n"(16 * (q + PF + r < DD4 ? q + PF + r : 0)));
          }
        }
        if constexpr (!UNI) {
          f32x2 bb = uh[k][0] * V[k][0], dd = f32x2{0.f, 0.f};
          if constexpr (MODE == 1) dd = uh[k][0] * DS[k][0];
#pragma unroll
          for (int h = 1; h < HP; ++h) {
            bb = uh[k][h] * V[k][h] + bb;
            if constexpr (MODE == 1) dd = uh[k][h] * DS[k][h] + dd;
          }
          b[k] = jv[k] ? bb[0] + bb[1] : -INFINITY;
          dc[k] = dd[0] + dd[1];
        }
      }
      float c[NJ];
      if constexpr (UNI) {
#pragma unroll
        for (int k = 0; k < NJ; ++k) c[k] = jv[k] ? invC : 0.f;
      } else {
        float m = b[0];
#pragma unroll
        for (int k = 1; k < NJ; ++k) m = fmaxf(m, b[k]);
        m = grp_max<SLOTS>(m);
        float z = 0.f;
#pragma unroll
        for (int k = 0; k < NJ; ++k) { c[k] = __expf(b[k] - m); z += c[k]; }
        z = 1.0f / grp_sum<SLOTS>(z);
#pragma unroll
        for (int k = 0; k < NJ; ++k) c[k] *= z;
      }
      if constexpr (MODE == 1) {
        float dot = 0.f;
#pragma unroll
        for (int k = 0; k < NJ; ++k) dot = c[k] * dc[k] + dot;          // lanes past C have c = 0
        dot = grp_sum<SLOTS>(dot);
        float cs[NJ];
#pragma unroll
        for (int k = 0; k < NJ; ++k) { cs[k] = c[k]; c[k] = c[k] * (dc[k] - dot); }       // db
        if (cdb_t != nullptr) {               // (uniform) hand c^t, db^t to the du / dW kernel: it then recomputes neither u_hat nor the softmax
          float* q = cdb_t + ((long long)rowi * N + i) * 2 * C;
#pragma unroll
          for (int k = 0; k < NJ; ++k)
            if (rv && jv[k]) { q[jk[k]] = cs[k]; q[C + jk[k]] = c[k]; }
        }
      }
#pragma unroll
      for (int k = 0; k < NJ; ++k)
#pragma unroll
        for (int h = 0; h < HP; ++h) ACC[k][h] = f32x2{c[k], c[k]} * uh[k][h] + ACC[k][h];
      __builtin_amdgcn_s_waitcnt(0x0F70);           // tile i+1 and u_{i+1} have landed
      if constexpr (!(DBG & 2)) __syncthreads();    // ... for every wave; and every wave is done with tile i
    }
  };
  using UniT = std::integral_constant<bool, true>;
  using SmT = std::integral_constant<bool, false>;
  const long long CD = (long long)C * DOUT;
  const long long plane = (long long)R * CD;
  // pair <-> float views of a lane's vector (compile-time indices: no instructions); the pad lane of an odd Dout stays 0
  auto get = [](const f32x2 (&P)[HP], float (&f)[DOUT]) {
#pragma unroll
    for (int o = 0; o < DOUT; ++o) f[o] = P[o >> 1][o & 1];
  };
  auto put = [](f32x2 (&P)[HP], const float (&f)[DOUT]) {
#pragma unroll
    for (int o = 0; o < DOUT; ++o) P[o >> 1][o & 1] = f[o];
    if constexpr (DOUT & 1) P[HP - 1][1] = 0.f;
  };
  auto load_vec = [&](f32x2 (&P)[HP], const float* p) {
    float f[DOUT];
#pragma unroll
    for (int o = 0; o < DOUT; ++o) f[o] = p[o];
    put(P, f);
  };
  auto store_vec = [&](const f32x2 (&P)[HP], float* p) {
#pragma unroll
    for (int o = 0; o < DOUT; ++o) p[o] = P[o >> 1][o & 1];
  };
  auto zero = [](f32x2 (&P)[NJ][HP]) {
#pragma unroll
    for (int k = 0; k < NJ; ++k)
#pragma unroll
      for (int h = 0; h < HP; ++h) P[k][h] = f32x2{0.f, 0.f};
  };

  if constexpr (MODE == 0) {
    f32x2 V[NJ][HP], S[NJ][HP], none[1][HP];
    zero(V);
    if (!a.fused) {
      // ---- one iteration's partial sums over [i0, i1) -> slab[chunk]
      if (a.it > 0) {
#pragma unroll
        for (int k = 0; k < NJ; ++k) load_vec(V[k], a.V + ((long long)rowi * C + jk[k]) * DOUT);
      }
      zero(S);
      if (a.it == 0) run_pass(UniT{}, V, none, S); else run_pass(SmT{}, V, none, S);
#pragma unroll
      for (int k = 0; k < NJ; ++k)
        if (rv && jv[k]) store_vec(S[k], a.slab + (long long)blockIdx.y * plane + ((long long)rowi * C + jk[k]) * DOUT);
      return;
    }
    // ---- all iterations for this block's rows in one launch
    for (int it = 0; it < a.n_iter; ++it) {
      zero(S);
      if (it == 0) run_pass(UniT{}, V, none, S); else run_pass(SmT{}, V, none, S);
      const bool last = it == a.n_iter - 1;
#pragma unroll
      for (int k = 0; k < NJ; ++k) {
        float sv[DOUT], vv[DOUT], vt[DOUT];
        get(S[k], sv);
        squash_v<DOUT>(sv, vv);
        get(V[k], vt);
#pragma unroll
        for (int o = 0; o < DOUT; ++o) vt[o] += vv[o];
        put(V[k], vt);
        if (rv && jv[k]) {
          float* sh = a.s_hist + (long long)it * plane + ((long long)rowi * C + jk[k]) * DOUT;
#pragma unroll
          for (int o = 0; o < DOUT; ++o) sh[o] = sv[o];
          if (last) {
            float* vo = a.v_out + (orow * C + jk[k]) * DOUT;
#pragma unroll
            for (int o = 0; o < DOUT; ++o) vo[o] = vv[o];
          }
        }
      }
    }
  } else {
    f32x2 V[NJ][HP], DS[NJ][HP], A[NJ][HP];
    if (!a.fused) {
      // ---- backward step t = a.it (>= 1): partial A_t over [i0, i1) -> slab[chunk]
#pragma unroll
      for (int k = 0; k < NJ; ++k) {
        const long long my = ((long long)rowi * C + jk[k]) * DOUT;
        load_vec(V[k], a.V + my);
        load_vec(DS[k], a.ds + my);
      }
      zero(A);
      run_pass(SmT{}, V, DS, A);
#pragma unroll
      for (int k = 0; k < NJ; ++k)
        if (rv && jv[k]) store_vec(A[k], a.slab + (long long)blockIdx.y * plane + ((long long)rowi * C + jk[k]) * DOUT);
      return;
    }
    // ---- backward over the iterations, t = T-1 .. 0: ds^t and V_t for every row (ds_all / V_all), the
    // dependence of later iterations on v^tau through V accumulates in SA
    f32x2 SA[NJ][HP];
    zero(SA);
    for (int it = a.n_iter - 1; it >= 0; --it) {
#pragma unroll
      for (int k = 0; k < NJ; ++k) {
        const long long my = ((long long)rowi * C + jk[k]) * DOUT;
        float vt[DOUT], sv[DOUT], vv[DOUT], dvv[DOUT], dsv[DOUT], sa[DOUT];
#pragma unroll
        for (int o = 0; o < DOUT; ++o) vt[o] = 0.f;
        for (int tau = 0; tau < it; ++tau) {
#pragma unroll
          for (int o = 0; o < DOUT; ++o) sv[o] = a.s_hist[(long long)tau * plane + my + o];
          squash_v<DOUT>(sv, vv);
#pragma unroll
          for (int o = 0; o < DOUT; ++o) vt[o] += vv[o];
        }
        const float* dvp = a.dv + (orow * C + jk[k]) * DOUT;
        get(SA[k], sa);
#pragma unroll
        for (int o = 0; o < DOUT; ++o) {
          sv[o] = a.s_hist[(long long)it * plane + my + o];
          dvv[o] = sa[o] + (it == a.n_iter - 1 ? dvp[o] : 0.f);
        }
        squash_bwd_v<DOUT>(sv, dvv, dsv);
        put(V[k], vt);
        put(DS[k], dsv);
        if (rv && jv[k]) {
#pragma unroll
          for (int o = 0; o < DOUT; ++o) {
            a.ds_all[(long long)it * plane + my + o] = dsv[o];
            a.V_all[(long long)it * plane + my + o] = vt[o];
          }
        }
      }
      if (it == 0) break;
      cdb_t = a.cdb != nullptr ? a.cdb + (long long)(it - 1) * R * N * 2 * C : nullptr;
      run_pass(SmT{}, V, DS, SA);                   // SA += A_t (nothing reads SA during the pass)
    }
  }
}

// registers: (3 + MODE) arrays of NJ*DP floats per lane, plus ~60 for the read ring, u, addresses and the softmax
#ifndef CY_ROWS_WPS_LIMIT
#define CY_ROWS_WPS_LIMIT 200
#endif
constexpr int pick_wps(int dout, int nj, int mode) { return (3 + mode) * nj * rows_dp(dout) <= CY_ROWS_WPS_LIMIT ? 2 : 1; }

template <int DOUT, int SLOTS, int NJ, int MODE>
int launch_cfg(const cyi_rows_args_t* a, const cyi_rows_plan_t* p, hipStream_t s) {
  constexpr int WPS = pick_wps(DOUT, NJ, MODE);
  const size_t lds = (size_t)2 * a->C * rows_ws(DOUT) * 4;
  if (p->wps != WPS) return cy_set_error(CY_EINVAL, "routing rows: plan/launch mismatch (wps %d vs %d)", p->wps, WPS);
  int rc = cy_allow_lds(caps_rows_kernel<DOUT, SLOTS, NJ, MODE, WPS>, lds);
  if (rc) return rc;
  caps_rows_kernel<DOUT, SLOTS, NJ, MODE, WPS><<<dim3(p->row_blocks, a->fused ? 1 : p->nch), 64 * p->waves, lds, s>>>(*a);
  return 0;
}
template <int DOUT, int MODE>
int launch_dout(const cyi_rows_args_t* a, const cyi_rows_plan_t* p, hipStream_t s) {
  if (p->slots == 64) return launch_cfg<DOUT, 64, 1, MODE>(a, p, s);
  if (p->nj == 1) return launch_cfg<DOUT, 16, 1, MODE>(a, p, s);
  if (p->nj == 2) return launch_cfg<DOUT, 16, 2, MODE>(a, p, s);
  if constexpr (DOUT <= 21) { if (p->nj == 3) return launch_cfg<DOUT, 16, 3, MODE>(a, p, s); }
  return cy_set_error(CY_EINVAL, "routing rows: no kernel for Dout=%d slots=%d nj=%d", DOUT, p->slots, p->nj);
}

}  // namespace

long long cyi_rows_wp_floats(int N, int C, int Dout) { return (long long)N * C * rows_ws(Dout); }

int cyi_rows_pack_w(const float* W, float* Wp, int N, int C, int Dout, hipStream_t s) {
  const int ws4 = rows_ws(Dout) / 4;
  const long long n4 = (long long)N * C * ws4;
  rows_pack_w_kernel<<<(unsigned)cy_ceil_div(n4, 256), 256, 0, s>>>(W, Wp, (long long)N * C, Dout, rows_dp(Dout), ws4);
  return 0;
}

void cyi_rows_plan(int R, int N, int C, int Dout, int mode, cyi_rows_plan_t* p) {
  int nj = (C + 15) / 16;
  p->slots = (C > 48 || (3 + 1) * nj * rows_dp(Dout) > 400) ? 64 : 16;      // one j per lane when the per-lane state would not fit
  if (p->slots == 64) nj = 1;
  p->nj = nj;
  p->wps = pick_wps(Dout, nj, mode);
  const int rpw = 64 / p->slots;                                   // rows per wave
  const int wr = (R + rpw - 1) / rpw;                              // waves' worth of rows
  const int maxw = 4 * p->wps;
  // many rows: as many waves per block as spread the row tiles over the 256 CUs (one block per CU: its W image is staged once);
  // few rows: one block holds them all (up to maxw waves) and the input capsules are split over blocks instead
  int waves = wr >= 256 ? (wr + 255) / 256 : wr;                  // (waves <= 4 of a block run on different SIMDs: a block's time does not depend on them)
  if (waves > maxw) waves = maxw;
  p->waves = waves;
  p->rows_per_block = waves * rpw;
  p->row_blocks = (R + p->rows_per_block - 1) / p->rows_per_block;
  int nch = 256 / p->row_blocks;
  if (nch > (N + 1) / 2) nch = (N + 1) / 2;
  if (nch < 1) nch = 1;
  p->ic = (N + nch - 1) / nch;
  p->nch = (N + p->ic - 1) / p->ic;
  p->phased = p->nch > 1;
  if (!p->phased) p->ic = N;
}

int cyi_rows_launch(int mode, const cyi_rows_args_t* a, const cyi_rows_plan_t* p, int Dout, hipStream_t s) {
  if (a->fused && p->phased) return cy_set_error(CY_EINVAL, "routing rows: fused launch of a phased plan");
  if (a->Wp == nullptr) return cy_set_error(CY_EINVAL, "routing rows: the packed W image is missing");
#define CY_ROWS_DISPATCH(D)                                                                       \
  case D: return mode == 0 ? launch_dout<D, 0>(a, p, s) : launch_dout<D, 1>(a, p, s);
  switch (Dout) {
    CY_ROWS_DISPATCH(5)
    CY_ROWS_DISPATCH(16)
    CY_ROWS_DISPATCH(21)
    CY_ROWS_DISPATCH(48)
    default: return cy_set_error(CY_EINVAL, "routing rows: Dout=%d is not built (5, 16, 21, 48)", Dout);
  }
#undef CY_ROWS_DISPATCH
}
