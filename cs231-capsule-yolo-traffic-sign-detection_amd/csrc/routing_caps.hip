// Capsule routing backward, second part: du and dW from the per-row quantities of the first part (gfx950).
//
//   du_hat_ij = ds^0_j / C + sum_{t>=1} ( c^t_ij ds^t_j + db^t_ij V_t[j] ),   c^t = softmax_j(u_hat_ij . V_t[j]),
//                                                                          db^t = c^t (dc^t - sum_j c^t dc^t), dc^t_ij = u_hat_ij . ds^t_j
//   du_i  = sum_j W_ij du_hat_ij                      dW_ij = sum_rows u_i (x) du_hat_ij
// (the autograd backward of models.py:70-79 with the logits identity of SURVEY F9; ds_all / V_all come from the row
// part, routing_rows.hip).  dW_ij sums over ROWS, so this part is input-capsule-stationary: a wave owns one input
// capsule i and walks the rows, lanes <-> output capsule j, dW_ij stays in registers for the whole walk.
//  * FMAs are packed over PAIRS of output components (Dout padded to an even DP in the LDS images): v_pk_fma_f32, so
//    that one wave per SIMD issues at the full vector rate.
//  * W_i (fixed for the walk) sits in LDS per wave as [j][8 * DP] (+ 4 floats of bank padding) and is streamed twice
//    per row (u_hat = u W, du = W du_hat) as float4 reads kept several reads ahead of their use.
//  * The row's vectors (ds^0, then V_t, ds^t for t = 1 .. T-1: (2T-1) * C * Dout floats, 18 KB at the DarkCapsuleNet3
//    head) are brought into a double-buffered LDS image by LDS-DMA while the previous row computes and are shared by
//    the block's G waves (G input capsules): L2 traffic for them drops by G, nothing is staged through registers.
#include "common.h"
#include <stdlib.h>

namespace {

// timing knob (results are wrong when set): 1 no row staging, 2 no du store, 4 no dW, 8 no iterations t >= 1, 16 no u prefetch
#ifndef CY_B2_DBG
#define CY_B2_DBG 0
#endif

// SAVED: the row part (routing_rows.hip, fused plans) left c^t and db^t of every (t >= 1, row, i, j) in `cdb`
// ([t - 1][row][i][2][C]): this kernel then recomputes neither u_hat (a pass over W_i) nor the logits and the softmax (two
// dot products over Dout and three wavefront reductions per iteration) -- 250 instead of 560 vector instructions per (row, i).
// tied wait for ST (2 or 4) float4 of the W ring: all but the `younger` youngest LDS reads have landed
template <int ST>
__device__ __forceinline__ void caps_wait(int younger, f32x4& a, f32x4& b, f32x4& c, f32x4& d) {
  if constexpr (ST == 4) {
    switch (younger) {
      case 12: asm volatile("s_waitcnt lgkmcnt(12)" : "+v"(a), "+v"(b), "+v"(c), "+v"(d)); break;
      case 8: asm volatile("s_waitcnt lgkmcnt(8)" : "+v"(a), "+v"(b), "+v"(c), "+v"(d)); break;
      case 4: asm volatile("s_waitcnt lgkmcnt(4)" : "+v"(a), "+v"(b), "+v"(c), "+v"(d)); break;
      default: asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a), "+v"(b), "+v"(c), "+v"(d)); break;
    }
  } else {
    switch (younger) {
      case 6: asm volatile("s_waitcnt lgkmcnt(6)" : "+v"(a), "+v"(b)); break;
      case 4: asm volatile("s_waitcnt lgkmcnt(4)" : "+v"(a), "+v"(b)); break;
      case 2: asm volatile("s_waitcnt lgkmcnt(2)" : "+v"(a), "+v"(b)); break;
      default: asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a), "+v"(b)); break;
    }
  }
}
constexpr int CDB_TMAX = 4;                         // iterations t >= 1 whose couplings a lane prefetches (n_iter <= 5)
// NTT: the number of routing iterations as a compile-time constant (3: every head of the reference's models), 0 = read from the arguments.
// With it the loops over the row's vectors and over the iterations unroll, their 64-bit offsets become loop invariants and the
// selects of the saved couplings disappear: the row loop of the DarkCapsuleNet3 head issued 495 scalar instructions per (row, i)
// next to 430 vector ones -- from ONE wave per SIMD, where every instruction of either kind takes an issue slot of ~4 cycles.
template <int DOUT, int G, bool SAVED, int NTT>
__global__ __launch_bounds__(64 * G, 1) void caps_bwd_kernel(cy_routing_bwd_t a, const float* __restrict__ cdb, int rows_per_chunk,
                                                             int nbuf) {
  constexpr int dbg = CY_B2_DBG;                    // developer knob (compile time: a uniform branch per use cost ~40 cycles each in this one-wave-per-SIMD loop)
  constexpr int DP = (DOUT + 1) & ~1, HP = DP / 2, DD = 8 * DP, WS = DD + 4, DD4 = DD / 4;
  static_assert(DD % 4 == 0, "W image must be a whole number of float4");
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int t = threadIdx.x, lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int C = a.C, N = a.N, R = a.R, NT = NTT > 0 ? NTT : a.n_iter;
  const int NV = 2 * NT - 1;                        // vectors per row: ds^0, (V_t, ds^t) for t = 1 .. T-1
  const int i = blockIdx.x * G + wave;
  const bool iv = i < N;
  const bool jv = lane < C;
  const int jl = jv ? lane : 0;
  const long long CD = (long long)C * DOUT;
  const long long plane = (long long)R * CD;
  const float* ds_all = a.ws;
  const float* V_all = a.ws + (long long)NT * plane;
  const float invC = 1.0f / (float)C;
  float* Wme = smem + wave * C * WS;                // this wave's W_i image
  float* rowbuf = smem + G * C * WS;                // [nbuf][NV][vstr]: the row's vectors as they lie in memory
  // Dout % 4 == 0 (16, 48): every capsule's Dout floats are whole 16-byte pieces; the image gives each capsule one
  // more piece (stride Dout + 4 floats) so that the lanes' ds_read_b128 fall on 16 different bank groups -- stored as
  // they lie (stride 16 floats) 32 lanes hit two banks.  Odd Dout (5, 21): the vectors are copied as they lie
  // (a stride of 21 floats is conflict-free for 4-byte reads).
  constexpr bool PADV = (DOUT % 4 == 0);
  constexpr int CS = PADV ? DOUT + 4 : DOUT;        // floats between two capsules of a vector in the image
  const int vstr = PADV ? C * CS : (((int)CD + 6) >> 2) << 2;
  const int rowstride = NV * vstr;
  const long long vall0 = (long long)NT * plane;    // V_all as an offset into the workspace

  // ---- W_i -> LDS [j][d * DP + o] (pad column zeroed), once per block
  if (iv) {
    const float* Wi = a.W + (long long)i * C * 8 * DOUT;
    for (int base = 0; base < C * DD; base += 64 * 8) {       // 8 independent loads in flight per lane
      float tmp[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const int idx = base + k * 64 + lane;
        const int j = idx / DD, rem = idx - j * DD, d = rem / DP, o = rem - d * DP;
        tmp[k] = (idx < C * DD && o < DOUT) ? Wi[(j * 8 + d) * DOUT + o] : 0.f;
      }
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const int idx = base + k * 64 + lane;
        const int j = idx / DD, rem = idx - j * DD;
        if (idx < C * DD) Wme[j * WS + rem] = tmp[k];
      }
    }
  }
  const int r0 = blockIdx.y * rows_per_chunk;
  const int r1 = min(R, r0 + rows_per_chunk);

  // row vectors -> LDS by 16-byte LDS-DMA: a vector (C * Dout floats at an arbitrary 4-byte alignment) is copied as the
  // aligned 16-byte pieces that cover it, so that its first float lands `offset & 3` floats into its image
  // (one wave-instruction moves 1 KiB; 4-byte pieces would need four times the DMA instructions, and those, ~100
  // cycles each per CU, were what bounded the first version of this kernel)
  // (roff = row * CD travels as a running uniform value: recomputed per vector and use, the 64-bit products were a quarter of the
  // row loop's scalar instructions)
  auto vec_off = [&](int v, long long roff) -> long long {  // offset of vector v of the row at `roff` in the workspace, in floats
    const int tt = (v + 1) >> 1;                    // v = 0: ds^0; v = 2t-1: V_t; v = 2t: ds^t
    return ((v & 1) ? vall0 : 0) + (long long)tt * plane + roff;
  };
  // Pieces per vector and the lane's source offset inside a vector do not depend on the row: an unpadded vector is copied as the
  // vstr / 4 pieces from its aligned start (>= what any alignment needs; the image has room for exactly these), so the lane
  // predicate, the divisions and -- for the heads of the reference's models, whose vectors fit ONE round of the block's lanes --
  // the whole loop structure leave the row loop (they were 175 of its instructions per row).
  constexpr int P5 = DOUT / 4 + 1;
  const int np = PADV ? C * P5 : (vstr >> 2);
  const bool one_round = np <= 64 * G;              // uniform
  const bool mine = t < np;
  int lsrc = 4 * t;                                 // floats from the vector's (aligned) start
  if constexpr (PADV) { const int j = t / P5, o4 = t - j * P5; lsrc = o4 < DOUT / 4 ? j * DOUT + 4 * o4 : 0; }
  auto stage_row = [&](long long roff, int buf) {
    float* dst = rowbuf + buf * rowstride;
    if (one_round) {
      if (mine) {
#pragma unroll
        for (int v = 0; v < NV; ++v) {
          const long long off = vec_off(v, roff);
          const float* srcv = a.ws + (PADV ? off : (off & ~3ll));       // (padded: 16-byte aligned, C * Dout is a multiple of 4)
          __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(srcv + lsrc),
                                           (__attribute__((address_space(3))) void*)(dst + v * vstr + 4 * (wave * 64)), 16, 0, 0);
        }
      }
      return;
    }
    if constexpr (PADV) {
      for (int v = 0; v < NV; ++v) {
        const float* srcv = a.ws + vec_off(v, roff);
        for (int base = 0; base < np; base += 64 * G) {
          const int q = base + t, j = q / P5, o4 = q - j * P5;
          if (q < np)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(o4 < DOUT / 4 ? srcv + j * DOUT + 4 * o4 : srcv),
                                             (__attribute__((address_space(3))) void*)(dst + v * vstr + 4 * (base + wave * 64)), 16, 0, 0);
        }
      }
      return;
    }
    for (int v = 0; v < NV; ++v) {
      const float* srcv = a.ws + (vec_off(v, roff) & ~3ll);
      for (int base = 0; base < np; base += 64 * G) {
        if (base + t < np)
          __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(srcv + 4 * (base + t)),
                                           (__attribute__((address_space(3))) void*)(dst + v * vstr + 4 * (base + wave * 64)), 16, 0, 0);
      }
    }
  };
  // uniform running state of the NEXT row to stage / prefetch: its row * CD, its offset in u (the cell gather's division pair is done
  // once: rows are cell-major, row = cell * B + image), its row * N * 2 * C in the saved couplings
  const int ii = iv ? i : 0;
  long long roff_n = (long long)r0 * CD;
  long long uoff_n = u_offset_g(r0, ii, N, a.gather_g, a.gather_B);
  int ub_n = a.gather_g ? r0 % a.gather_B : 0;
  const long long ustep = a.gather_g ? 16ll * a.gather_g * a.gather_g * 256 : (long long)N * 8;
  const long long uwrap = a.gather_g ? 4ll * 256 - (long long)a.gather_B * ustep : 0;
  const long long cstep = (long long)N * 2 * C;
  long long crow_n = (long long)r0 * cstep;
  auto advance_next = [&]() {
    roff_n += CD; crow_n += cstep; uoff_n += ustep;
    if (a.gather_g && ++ub_n == a.gather_B) { ub_n = 0; uoff_n += uwrap; }
  };
  if (r0 < r1) stage_row(roff_n, 0);
  f32x4 un0, un1;                                   // u of the next row (prefetched)
  auto load_u = [&](long long uoff) {
    const f32x4* p = (const f32x4*)(a.u + uoff);
    un0 = p[0];
    un1 = p[1];
  };
  if (r0 < r1) load_u(uoff_n);
  float ccn[CDB_TMAX] = {0.f, 0.f, 0.f, 0.f}, dbn[CDB_TMAX] = {0.f, 0.f, 0.f, 0.f};   // SAVED: c^t, db^t of the next row (prefetched like u)
  static_assert(CDB_TMAX == 4, "initialisers above");
  const float* cdb_i = SAVED ? cdb + (long long)ii * 2 * C + jl : nullptr;
  auto load_cdb = [&](long long crow) {
    if constexpr (SAVED) {
#pragma unroll
      for (int tt = 0; tt < CDB_TMAX; ++tt) {
        if (NTT > 0 && tt >= NTT - 1) continue;       // (compile-time iteration count: no slots past the last iteration)
        const int tq = tt < NT - 1 ? tt : NT - 2;     // (run-time count: slots past the last iteration re-read it: no branch; never used)
        const float* q = cdb_i + (long long)tq * R * cstep + crow;
        ccn[tt] = q[0];
        dbn[tt] = q[C];
      }
    }
  };
  if (r0 < r1) load_cdb(crow_n);
  long long roff_c = roff_n, uoff_c = uoff_n;       // ... and of the row being computed
  advance_next();

  f32x2 dw[8][HP];
#pragma unroll
  for (int d = 0; d < 8; ++d)
#pragma unroll
    for (int h = 0; h < HP; ++h) dw[d][h] = f32x2{0.f, 0.f};

  __builtin_amdgcn_s_waitcnt(0x0F70);               // vmcnt(0): first row landed (and the plain W stores are LDS ops)
  __syncthreads();
  const unsigned wa = (unsigned)(uintptr_t)(__attribute__((address_space(3))) const float*)(Wme + jl * WS);
  // reads of the W image per tied wait and in flight (routing_rows.hip: a tied wait and the wait state hipcc pads behind it are issue
  // slots; up to four waves a wave has 512 registers for the deeper ring, the six-wave variants spill as it is)
#ifndef CY_CAPS_ST
#define CY_CAPS_ST 4
#endif
#ifndef CY_CAPS_PF
#define CY_CAPS_PF 8
#endif
  constexpr int ST = G <= 4 ? CY_CAPS_ST : 2;
  constexpr int PF = G <= 4 ? CY_CAPS_PF : 4;
  static_assert(DD4 % ST == 0 && DD4 >= PF, "the W image is read in groups of ST float4");

  // pair h of a Dout-vector in LDS (4-byte aligned: ds_read2_b32); the pad component of an odd Dout reads as 0
  auto ldpair = [&](const float* p, int h) -> f32x2 {
    if ((DOUT & 1) && h == HP - 1) return f32x2{p[2 * h], 0.f};
    return f32x2{p[2 * h], p[2 * h + 1]};
  };
  const bool b0 = lane & 1, b1 = lane & 2, b2 = lane & 4;
  float du_prev = 0.f;
  long long du_off_prev = -1;
  for (int row = r0; row < r1; ++row) {
    const int cur = nbuf == 2 ? ((row - r0) & 1) : 0;
    // the previous row's du goes out here, not at the end of its own iteration: a store counts in vmcnt and the
    // end-of-row wait for the next row's data would otherwise sit out its whole round trip
    if (iv && lane < 8 && du_off_prev >= 0 && !(dbg & 2)) a.du[du_off_prev + lane] = du_prev;
    float uv[8] = {un0[0], un0[1], un0[2], un0[3], un1[0], un1[1], un1[2], un1[3]};
    float ccur[CDB_TMAX], dbcur[CDB_TMAX];
    if constexpr (SAVED) {
#pragma unroll
      for (int tt = 0; tt < CDB_TMAX; ++tt) { ccur[tt] = jv ? ccn[tt] : 0.f; dbcur[tt] = jv ? dbn[tt] : 0.f; }   // lanes past C: c = db = 0
    }
    if (row + 1 < r1) {
      if (nbuf == 2 && !(dbg & 1)) stage_row(roff_n, cur ^ 1);
      if (!(dbg & 16)) load_u(uoff_n);
      load_cdb(crow_n);
    }
    const float* rb = rowbuf + cur * rowstride + jl * CS;      // vector v of this lane's capsule: rb + v * vstr (+ its offset & 3 when unpadded)
    // ---- u_hat = u W_ij (pairs of output components)
    f32x2 uh[HP];
#pragma unroll
    for (int h = 0; h < HP; ++h) uh[h] = f32x2{0.f, 0.f};
    if constexpr (!SAVED) {
      f32x4 wq[PF];
#pragma unroll
      for (int p = 0; p < PF; ++p) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(wq[p]) : "v"(wa), "n"(16 * p));
#pragma unroll
      for (int q = 0; q < DD4; q += ST) {             // ST reads per tied wait
        constexpr int dummyq = 0; (void)dummyq;
        const int younger = (q + PF <= DD4 ? PF : DD4 - q) - ST;      // reads in flight behind this group
        caps_wait<ST>(younger, wq[q % PF], wq[(q + 1) % PF], wq[(q + 2 < DD4 ? q + 2 : q) % PF], wq[(q + 3 < DD4 ? q + 3 : q) % PF]);
#pragma unroll
        for (int e = 0; e < 2 * ST; ++e) {
          const int f = 4 * q + 2 * e, d = f / DP, h = (f % DP) / 2;
          const f32x4& w = wq[(q + (e >> 1)) % PF];
          uh[h] = f32x2{w[2 * (e & 1)], w[2 * (e & 1) + 1]} * uv[d] + uh[h];
        }
        if (q + PF < DD4) {
#pragma unroll
          for (int r = 0; r < ST; ++r) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(wq[(q + r) % PF]) : "v"(wa), "n"(16 * (q + PF + r)));
        }
      }
    }
    // ---- du_hat = ds^0 / C + sum_t (c^t ds^t + db^t V_t); V_t and ds^t are streamed from LDS twice (dots, then the
    // update) instead of being held: dW owns the registers
    f32x2 duh[HP];
#pragma unroll
    for (int h = 0; h < HP; ++h) duh[h] = ldpair(rb + (PADV ? 0 : (int)(vec_off(0, roff_c) & 3)), h) * (jv ? invC : 0.f);    // lanes past C read capsule 0, scaled by 0
    auto t_body = [&](int it) {
      const float* vp = rb + (2 * it - 1) * vstr + (PADV ? 0 : (int)(vec_off(2 * it - 1, roff_c) & 3));
      const float* dp = rb + (2 * it) * vstr + (PADV ? 0 : (int)(vec_off(2 * it, roff_c) & 3));
      // all pairs of V_t and ds^t with ONE wait (inline asm: hipcc would wait behind every read)
      const unsigned va = (unsigned)(uintptr_t)(__attribute__((address_space(3))) const float*)vp;
      const unsigned da = (unsigned)(uintptr_t)(__attribute__((address_space(3))) const float*)dp;
      f32x2 Vt[HP], dst[HP];
#pragma unroll
      for (int h = 0; h < HP; ++h) {
        if constexpr (PADV) {                                  // two pairs per 16-byte read
          if (h % 2 == 0) {
            f32x4 t4, d4;
            asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(t4) : "v"(va), "n"(8 * h));
            asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(d4) : "v"(da), "n"(8 * h));
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(t4), "+v"(d4));
            Vt[h] = f32x2{t4[0], t4[1]}; Vt[h + 1] = f32x2{t4[2], t4[3]};
            dst[h] = f32x2{d4[0], d4[1]}; dst[h + 1] = f32x2{d4[2], d4[3]};
          }
        } else if ((DOUT & 1) && h == HP - 1) {
          asm volatile("ds_read2_b32 %0, %1 offset0:%2 offset1:%2" : "=v"(Vt[h]) : "v"(va), "n"(2 * h));    // pad component zeroed below
          asm volatile("ds_read2_b32 %0, %1 offset0:%2 offset1:%2" : "=v"(dst[h]) : "v"(da), "n"(2 * h));
        } else {
          asm volatile("ds_read2_b32 %0, %1 offset0:%2 offset1:%3" : "=v"(Vt[h]) : "v"(va), "n"(2 * h), "n"(2 * h + 1));
          asm volatile("ds_read2_b32 %0, %1 offset0:%2 offset1:%3" : "=v"(dst[h]) : "v"(da), "n"(2 * h), "n"(2 * h + 1));
        }
      }
      if constexpr (SAVED) {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int h = 0; h < HP; ++h) {
          asm volatile("" : "+v"(Vt[h]), "+v"(dst[h]));
          if ((DOUT & 1) && h == HP - 1) { Vt[h][1] = 0.f; dst[h][1] = 0.f; }
        }
        float c = 0.f, db = 0.f;
#pragma unroll
        for (int tt = 0; tt < CDB_TMAX; ++tt)
          if (tt == it - 1) { c = ccur[tt]; db = dbcur[tt]; }     // (uniform selects: `it` is a scalar)
        // two passes: a packed fp32 operation right behind the one it depends on costs a wait state (hipcc pads with s_nop, an issue slot)
#pragma unroll
        for (int h = 0; h < HP; ++h) duh[h] = Vt[h] * db + duh[h];
#pragma unroll
        for (int h = 0; h < HP; ++h) duh[h] = dst[h] * c + duh[h];
        return;
      }
      f32x2 bb = {0.f, 0.f}, dd = {0.f, 0.f};
#pragma unroll
      for (int h = 0; h < HP; ++h) {
        // the wait is tied to the registers it releases (an asm output is "ready" for the compiler as soon as the
        // statement has been issued); 2 * (HP - 1 - h) younger reads may stay in flight
        const int younger = PADV ? 0 : 2 * (HP - 1 - h);
        if (PADV) {}
        else if (younger >= 12) asm volatile("s_waitcnt lgkmcnt(12)" : "+v"(Vt[h]), "+v"(dst[h]));
        else if (younger >= 8) asm volatile("s_waitcnt lgkmcnt(8)" : "+v"(Vt[h]), "+v"(dst[h]));
        else if (younger >= 4) asm volatile("s_waitcnt lgkmcnt(4)" : "+v"(Vt[h]), "+v"(dst[h]));
        else if (younger >= 2) asm volatile("s_waitcnt lgkmcnt(2)" : "+v"(Vt[h]), "+v"(dst[h]));
        else asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(Vt[h]), "+v"(dst[h]));
        if ((DOUT & 1) && h == HP - 1) { Vt[h][1] = 0.f; dst[h][1] = 0.f; }
        bb = uh[h] * Vt[h] + bb;
        dd = uh[h] * dst[h] + dd;
      }
      const float b = jv ? bb[0] + bb[1] : -INFINITY;
      const float dc = dd[0] + dd[1];
      const float m = wave_allmax(b);
      const float e = jv ? __expf(b - m) : 0.f;
      const float c = e * __frcp_rn(wave_allsum(e));
      const float dot = wave_allsum(c * dc);
      const float db = c * (dc - dot);
#pragma unroll
      for (int h = 0; h < HP; ++h) duh[h] = Vt[h] * db + duh[h];                    // c = db = 0 on lanes past C
#pragma unroll
      for (int h = 0; h < HP; ++h) duh[h] = dst[h] * c + duh[h];
    };
#pragma unroll
    for (int it = 1; it < ((dbg & 8) ? 1 : NT); ++it) t_body(it);
    // ---- du_i[d] = sum_j sum_o W[j][d][o] du_hat_j[o];  dW_ij[d][o] += u[d] du_hat_j[o]
    float p[8];
    {
      f32x2 acc[8];
#pragma unroll
      for (int d = 0; d < 8; ++d) acc[d] = f32x2{0.f, 0.f};
      f32x4 wq[PF];
      // read order: float4 q of the image's first half (d = 0..3), then its counterpart of the second half (d + 4), ...: the two packed
      // FMAs of one float4 feed the SAME accumulator as a rule, and back to back the second one waits a state (an s_nop issue slot each)
      static_assert(DD4 % 2 == 0, "the W image splits into two halves of whole float4");
      auto qord = [](int k) { return (k & 1) ? DD4 / 2 + (k >> 1) : (k >> 1); };
#pragma unroll
      for (int pq = 0; pq < PF; ++pq) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(wq[pq]) : "v"(wa), "n"(16 * qord(pq)));
#pragma unroll
      for (int k = 0; k < DD4; k += ST) {
        const int younger = (k + PF <= DD4 ? PF : DD4 - k) - ST;
        caps_wait<ST>(younger, wq[k % PF], wq[(k + 1) % PF], wq[(k + 2 < DD4 ? k + 2 : k) % PF], wq[(k + 3 < DD4 ? k + 3 : k) % PF]);
#pragma unroll
        for (int e = 0; e < 2; ++e)
#pragma unroll
          for (int r = 0; r < ST; ++r) {              // (r inner: neighbours feed different accumulators)
            const int f = 4 * qord(k + r) + 2 * e;
            const f32x4& w = wq[(k + r) % PF];
            acc[f / DP] = f32x2{w[2 * e], w[2 * e + 1]} * duh[(f % DP) / 2] + acc[f / DP];
          }
        if (k + PF < DD4) {
#pragma unroll
          for (int r = 0; r < ST; ++r)
            asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(wq[(k + r) % PF]) : "v"(wa), "n"(16 * qord(k + PF + r < DD4 ? k + PF + r : 0)));
        }
      }
#pragma unroll
      for (int d = 0; d < 8; ++d) p[d] = acc[d][0] + acc[d][1];
    }
    if (!(dbg & 4)) {
#pragma unroll
    for (int d = 0; d < 8; ++d)
#pragma unroll
      for (int h = 0; h < HP; ++h) dw[d][h] = duh[h] * uv[d] + dw[d][h];
    }
    // sum over the lanes (capsules j; lanes past C hold du_hat = 0) of 8 values per lane: three halving exchanges
    // (lane pairs l <-> 7-l, l <-> l^1, l <-> l^2: afterwards lane l holds component l & 7 summed over its 8-lane
    // group), then all-reduce over the eight groups -- 26 instructions instead of eight wavefront reductions
    {
      float q4[4], q2[2];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const float keep = b2 ? p[k + 4] : p[k], give = b2 ? p[k] : p[k + 4];
        q4[k] = keep + dpp_get<0x141, 0xF>(0.f, give);          // row_half_mirror
      }
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        const float keep = b0 ? q4[2 * k + 1] : q4[2 * k], give = b0 ? q4[2 * k] : q4[2 * k + 1];
        q2[k] = keep + dpp_get<0xB1, 0xF>(0.f, give);           // quad_perm [1,0,3,2]
      }
      const float keep = b1 ? q2[1] : q2[0], give = b1 ? q2[0] : q2[1];
      float tot = keep + dpp_get<0x4E, 0xF>(0.f, give);         // quad_perm [2,3,0,1]
      tot += dpp_get<0x128, 0xF>(0.f, tot);                     // row_ror:8
      tot = swap_add32(swap_add16(tot));
      du_prev = tot;
    }
    du_off_prev = uoff_c;
    roff_c = roff_n; uoff_c = uoff_n;               // the next row becomes the current one
    advance_next();

    __builtin_amdgcn_s_waitcnt(0x0F70);             // next row's vectors (LDS-DMA) and u have landed
    __syncthreads();                                // ... for every wave; and every wave is done with this row's image
    if (nbuf == 1 && row + 1 < r1 && !(dbg & 1)) {
      stage_row(roff_c, 0);
      __builtin_amdgcn_s_waitcnt(0x0F70);
      __syncthreads();
    }
  }

  if (iv && lane < 8 && du_off_prev >= 0) a.du[du_off_prev + lane] = du_prev;

  // ---- dW_i: registers -> this wave's LDS region (its W image is no longer needed) -> coalesced stores / atomics
  if (iv && jv) {
#pragma unroll
    for (int d = 0; d < 8; ++d)
#pragma unroll
      for (int h = 0; h < HP; ++h) *(f32x2*)(Wme + lane * WS + d * DP + 2 * h) = dw[d][h];
  }
  __syncthreads();
  if (iv) {
    float* dWi = a.dW + (long long)i * C * 8 * DOUT;
    const bool add = gridDim.y > 1;
    for (int idx = lane; idx < C * 8 * DOUT; idx += 64) {
      const int j = idx / (8 * DOUT), rem = idx - j * (8 * DOUT), d = rem / DOUT, o = rem - d * DOUT;
      const float val = Wme[j * WS + d * DP + o];
      if (add) atomicAdd(dWi + idx, val); else dWi[idx] = val;
    }
  }
}

template <int DOUT, int G>
int launch_g(const cy_routing_bwd_t* a, const float* cdb, int chunks, int rpc, int nbuf, size_t lds, hipStream_t s) {
  const dim3 grid((a->N + G - 1) / G, chunks);
  auto go = [&](auto kernel, const float* c) -> int {
    int rc = cy_allow_lds(kernel, lds);
    if (rc) return rc;
    kernel<<<grid, 64 * G, lds, s>>>(*a, c, rpc, nbuf);
    return 0;
  };
  const bool saved = cdb != nullptr && a->n_iter - 1 <= CDB_TMAX && G <= 4;    // (six waves share four SIMDs: 256 registers, the variant spills)
  if (a->n_iter == 3) return saved ? go(caps_bwd_kernel<DOUT, G, true, 3>, cdb) : go(caps_bwd_kernel<DOUT, G, false, 3>, nullptr);
  return saved ? go(caps_bwd_kernel<DOUT, G, true, 0>, cdb) : go(caps_bwd_kernel<DOUT, G, false, 0>, nullptr);
}

template <int DOUT>
int launch_dout(const cy_routing_bwd_t* a, const float* cdb, hipStream_t s) {
  constexpr int DP = (DOUT + 1) & ~1, WS = 8 * DP + 4;
  const size_t wbytes = (size_t)a->C * WS * 4;
  auto rbytes_of = [&](int) {
    const size_t vstr = (DOUT % 4 == 0) ? (size_t)a->C * (DOUT + 4) : (size_t)(((a->C * DOUT + 6) >> 2) << 2);
    return (size_t)(2 * a->n_iter - 1) * vstr * 4;                      // the DMA lanes past an image's end are masked off
  };
  const size_t cap = 160 * 1024;
  // G input capsules (waves) per block: the fewest rounds of blocks over the 256 CUs (one block per CU: the W images
  // fill the LDS), then the row image double-buffered, then the larger group
  int G = 0, nbuf = 1;
  long long best = -1;
  for (int g : {6, 4, 2, 1}) {
    const size_t need1 = g * wbytes + rbytes_of(g);
    if (need1 > cap) continue;
    const int nb = (g * wbytes + 2 * rbytes_of(g) <= cap) ? 2 : 1;
    const int ig = (a->N + g - 1) / g;
    int ch = 256 / ig;
    if (ch > (a->R + 15) / 16) ch = (a->R + 15) / 16;
    if (ch < 1) ch = 1;
    const long long rounds = ((long long)ig * ch + 255) / 256;
    // (six waves share four SIMDs: 256 registers per wave, the Dout >= 16 variants spill 40 .. 700 bytes of scratch per lane -- worth a
    // whole extra round of blocks: CapsuleNet head 0.341 -> 0.313 ms on four waves in 1.27 rounds)
    const long long score = rounds * 16 - (nb == 2 ? 2 : 0) - (g == 4 ? 1 : 0) + ((g == 6 && DOUT >= 16) ? 16 : 0);
    if (best < 0 || score < best) { best = score; G = g; nbuf = nb; }
  }
  if (G == 0)
    return cy_set_error(CY_EINVAL, "cy_routing_bwd: C=%d Dout=%d n_iter=%d needs %zu bytes of LDS (> 160 KiB)", a->C, DOUT,
                        a->n_iter, wbytes + rbytes_of(1));
  const size_t rbytes = rbytes_of(G);
  const size_t lds = G * wbytes + nbuf * rbytes;
  // row chunks: about one block per CU; chunks > 1 add their dW with atomics (dW zeroed by the caller)
  const int igroups = (a->N + G - 1) / G;
  int chunks = 256 / igroups;
  if (chunks > (a->R + 15) / 16) chunks = (a->R + 15) / 16;
  if (chunks < 1) chunks = 1;
  const int rpc = (a->R + chunks - 1) / chunks;
  chunks = (a->R + rpc - 1) / rpc;
  if (G == 6) return launch_g<DOUT, 6>(a, cdb, chunks, rpc, nbuf, lds, s);
  if (G == 4) return launch_g<DOUT, 4>(a, cdb, chunks, rpc, nbuf, lds, s);
  if (G == 2) return launch_g<DOUT, 2>(a, cdb, chunks, rpc, nbuf, lds, s);
  return launch_g<DOUT, 1>(a, cdb, chunks, rpc, nbuf, lds, s);
}

}  // namespace

int cyi_caps_bwd_launch(const cy_routing_bwd_t* a, const float* cdb, hipStream_t s) {
  switch (a->Dout) {
    case 5: return launch_dout<5>(a, cdb, s);
    case 16: return launch_dout<16>(a, cdb, s);
    case 21: return launch_dout<21>(a, cdb, s);
    case 48: return launch_dout<48>(a, cdb, s);
    default: return cy_set_error(CY_EINVAL, "cy_routing_bwd: Dout=%d is not built (5, 16, 21, 48)", a->Dout);
  }
}
