// Fused Winograd F(4x4, 3x3) convolution on the fp32 matrix cores (gfx950) for the 3x3 / stride 1 / pad 1 layers
// (DarkCapsuleNet conv_2 = 77 % of the model's FLOPs, models.py:349-351, and its input gradient).
//
//   Y = A^T [ sum_ci (G g G^T) (.) (B^T d B) ] A     -- 36 multiplies per 4x4 outputs: 2.25 per output against 4 for
// F(2x2, 3x3) (winograd.hip) and 9 for the direct form, i.e. 1.78x fewer MFMAs than winograd.hip for the same layer.
// Interpolation points 0, +-1, +-2 (Lavin & Gray); fp32 error of the forward 2.3e-6 relative L2 on conv_2-shaped data
// against 3.6e-7 for F(2x2, 3x3) and 6.0e-7 for the direct fp32 chain (tools/probe/wino_f43_numerics.py).
//
// One block = 4 x 8 tiles (16 x 32 output pixels) x 64 output channels, 4 waves, ONE wave per SIMD with the whole
// 512-register file.  A wave owns ALL 36 Winograd positions of the block's 32 tiles for 16 output channels:
// 36 x 2 accumulator tiles of v_mfma_f32_16x16x4_f32 (tiles are the M dimension, 16 per tile half) = 288 registers
// (positions 0..31 in AGPRs, 32..35 in arch VGPRs), so the output transform is lane-local -- no exchange between waves.
// Per chunk of 8 input channels (144 MFMAs per wave):
//   * the raw 18 x 34 input patch goes global -> registers -> LDS ([k-quad][pixel][4]),
//   * every thread transforms HALF an item (tile, channel pair): rows 0..2 or 3..5 of V = B^T d B (72 v_pk_fma_f32) and
//     writes them into the A-operand image V[pos][k pair][tile slot (swizzled)][tile half][2 k-steps],
//   * the transformed weights never touch LDS: they are packed per call in the B-operand order of each wave
//     (U[co block][chunk][wave][position pair][lane][4]) and stream global/L2 -> registers through a ring of F4_NBR (9)
//     dwordx4 loads per wave, each issued F4_NBR position pairs (72 MFMAs) before its first use.
// The grid is persistent (one block per CU, XCD-aware tile order); a block walks its tiles as one continuous stream of
// chunks (winograd_s2.hip: the loads and the input transform of the next tile's first chunks run under the MFMAs of
// the current tile's last chunks).  The loop's global loads (input patch, B-operand ring of F4_NBR position pairs) are inline asm
// with hand-counted `s_waitcnt vmcnt(N)` (hipcc's own bookkeeping drew vmcnt(0) at the loop header); tools/check_vmcnt.py replays them.
#include <type_traits>
#include "common.h"

namespace {

// developer knob for timing experiments (results are wrong when set): drop 1 the input transform's arithmetic and V stores,
// 2 the patch loads / stores, 4 the B-operand loads, 8 the accumulator drain, 16 the transform's patch reads
#ifndef CY_F4_DBG
#define CY_F4_DBG 0
#endif
constexpr int F4DBG = CY_F4_DBG;

constexpr int F4_PC = 34;                       // patch columns: 8 tiles x 4 + 2
constexpr int F4_NPIX = 18 * F4_PC;             // 612 patch pixels
constexpr int F4_RAWP = 625;                    // >= 612, = 1 (mod 16): the k-quad stride is 4 banks (mod 64)
constexpr int F4_RAW_BUF = 2 * F4_RAWP * 4;     // floats: [kq][pixel][4]
constexpr int F4_V_BUF = 36 * 256;              // floats: [pos][kg][16 tile slots][tile half][2 k-steps]
constexpr int F4_NQ = 5;                        // patch float4 items per thread (1224 over 256 threads)
#ifndef CY_F4_NBR
#define CY_F4_NBR 9
#endif
constexpr int F4_NBR = CY_F4_NBR;              // ring of B-operand loads per wave (position pairs in flight; divides 18)
static_assert(18 % F4_NBR == 0 && F4_NBR >= 1 && F4_NBR <= 18 && (F4_NBR + 3) / 4 <= 5,
              "winograd4: the ring-slot identity (k + NBR - 18) % NBR == k % NBR and the hand-counted vmcnt schedule need NBR | 18");
constexpr int F4_OG = 272;                      // floats per lane group of the drain scratch: 16 pixels x 16 channels + 16 pad
constexpr int F4_OSTEP = 4 * F4_OG;             // one drain step of a wave
constexpr int F4_BAR = 136;                     // slot of the chunk's only barrier (the MFMAs behind it read registers only)

#ifdef CY_F4_PROF
// developer instrumentation (tools/f4prof.py): s_memtime stamps of chunk 5 of every tile and of the drain, per block
__device__ unsigned long long f4_prof_buf[256 * 16];
#endif

struct Wino4Args {
  const float* X; const float* U; float* Y; const float* bias; double* stats;
  int B, H, W, Cin, Cout, Np, tbh, tbw;
  int ntiles;                                   // B * tbh * tbw * Np/64 output tiles, walked by a persistent grid
  float out_slope;                              // EPI == 2: Y = lrelu(conv + bias) (eval forward, BatchNorm folded into U / bias)
};

__device__ __forceinline__ void mfma16_a(f32x4& c, float a, float b) {
  asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+a"(c) : "v"(a), "v"(b));
}
__device__ __forceinline__ void mfma16_v(f32x4& c, float a, float b) {
  asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(c) : "v"(a), "v"(b));
}
// packed fp32 arithmetic as plain vector expressions: hipcc selects v_pk_fma_f32 / v_pk_add_f32 for them on gfx950 (with inline
// constants and neg modifiers), and -- unlike inline-asm statements -- needs no s_nop pad between two dependent ones
__device__ __forceinline__ f32x2 pkfma(f32x2 x, f32x2 y, f32x2 z) { return __builtin_elementwise_fma(x, y, z); }    // x * y + z
__device__ __forceinline__ f32x2 pkfnma(f32x2 x, f32x2 y, f32x2 z) { return __builtin_elementwise_fma(-x, y, z); }  // z - x * y
__device__ __forceinline__ f32x2 pkadd(f32x2 x, f32x2 y) { return x + y; }
__device__ __forceinline__ f32x2 pksub(f32x2 x, f32x2 y) { return x - y; }
// The B-operand ring is loaded and waited for by hand: tracked by hipcc, the loop header of the chunk loop waited vmcnt(0)
// (the join of the loop's back edge with its entry), i.e. for the B loads issued in the chunk's last slots.  The counted
// wait in front of a pair's first MFMA is the EXACT number of vector-memory operations the schedule issues between the
// pair's load and that MFMA (f4_younger_b: with a flat vmcnt(4) the in-order counter made pair 2 wait for the patch loads
// issued a few slots earlier, i.e. for HBM latency: the first third of a chunk took 2.3x its MFMA time); output stores of a
// drain in between only add younger operations.
template <int OFF> __device__ __forceinline__ void bload(f32x4& dst, const char* base, unsigned voff) {
  if constexpr (CY_F4_DBG & 4) dst = f32x4{1.f, 1.f, 1.f, 1.f};
  else asm volatile("global_load_dwordx4 %0, %1, %2 offset:%3" : "=v"(dst) : "v"(voff), "s"(base), "n"(OFF));
}
template <int N> __device__ __forceinline__ void vmwait(f32x4& x) { asm volatile("s_waitcnt vmcnt(%1)" : "+v"(x) : "n"(N)); }
// ... and so are the patch loads (a tracked load pending at the loop header draws the same vmcnt(0), which would then also
// wait for the youngest ring loads).  A patch item is stored to LDS one chunk after its load; its wait counts the operations
// younger than the load in the FIRST chunk of a block (f4_younger_r: 4 - q patch loads + the 6 ring loads of the prologue +
// what the chunk has issued by then), fewer than in any later chunk.
// The patch is read through a buffer descriptor of ONE image (base = the image, num_records = its bytes): a padding item
// gets an offset beyond the image and the hardware's range check returns zeros -- no select, no zero page, no masks; the
// chunk's channel offset travels as the instruction's scalar offset (no vector add per chunk).
typedef int i32x4_ __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void rload(f32x4& dst, i32x4_ desc, unsigned voff, unsigned soff) {
#ifndef CY_F4_RLOAD_AUX
#define CY_F4_RLOAD_AUX ""
#endif
  asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen" CY_F4_RLOAD_AUX : "=v"(dst) : "v"(voff), "s"(desc), "s"(soff));
}
__device__ __forceinline__ void rwait0(f32x4& x) { asm volatile("s_waitcnt vmcnt(0)" : "+v"(x)); }
__device__ __forceinline__ float acc_elem4(float a_elem) {    // one accumulator element, read where the statement stands
  float x;
  asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(x) : "a"(a_elem));
  return x;
}

// ---- compile-time schedule of one chunk: 144 slots; slot s issues the MFMA of position s >> 2, tile half s & 1, k-step
// (s >> 1) & 1 (consecutive MFMAs alternate between the position's two accumulators).  The A fragment of position p + 2 (both
// tile halves, both k-steps: one ds_read_b128) is fetched in slot 4p.  Side work, one piece per slot, only in slots 4p + 2 and 4p + 3:
//   1 G_B    B operand of position pair q + F4_NBR (ring slot q % F4_NBR), right behind the last MFMA of pair q
//   3 S_raw  one float4 of the patch of chunk f + 2: registers -> LDS        2 G_raw  one patch load of chunk f + 3
//   4 T_rd   one column (7 float2) of the thread's patch of chunk f + 1      5 T_col  half a column of T = B^T d
//   6 T_row  a quarter of one row of V = T B (3 packed FMAs; the last two quarters store 3 positions each)
//   7 ADV    patch cursor to f + 4
struct F4Sched { int kind[144]; int idx[144]; };
constexpr F4Sched f4_make_sched() {
  F4Sched s{};
  for (int i = 0; i < 144; ++i) { s.kind[i] = 0; s.idx[i] = 0; }
  for (int q = 0; q < 18; ++q) {
    const int sl = 8 * q + 10 > 143 ? 143 : 8 * q + 10;
    s.kind[sl] = 1; s.idx[sl] = q;
  }
  int pk[48] = {}, pi[48] = {}, n = 0;
  for (int q = 0; q < F4_NQ; ++q) { pk[n] = 3; pi[n++] = q; pk[n] = 2; pi[n++] = q; }
  pk[n] = 7; pi[n++] = 0;
  for (int c = 0; c < 6; ++c) {
    pk[n] = 4; pi[n++] = c;
    if (c > 0) { pk[n] = 5; pi[n++] = 2 * (c - 1); pk[n] = 5; pi[n++] = 2 * (c - 1) + 1; }
  }
  pk[n] = 5; pi[n++] = 10; pk[n] = 5; pi[n++] = 11;
  for (int r = 0; r < 12; ++r) { pk[n] = 6; pi[n++] = r; }
  int sl = 2;
  for (int i = 0; i < n; ++i) {
    while (sl < 144 && ((sl & 3) < 2 || s.kind[sl] != 0)) ++sl;
    s.kind[sl] = pk[i]; s.idx[sl] = pi[i];
    ++sl;
  }
  return s;
}
constexpr F4Sched F4S = f4_make_sched();
constexpr bool f4_sched_ok() {                  // every piece placed exactly once
  int cnt[8] = {};
  for (int i = 0; i < 144; ++i) cnt[F4S.kind[i]]++;
  return cnt[1] == 18 && cnt[2] == F4_NQ && cnt[3] == F4_NQ && cnt[4] == 6 && cnt[5] == 12 && cnt[6] == 12 && cnt[7] == 1;
}
static_assert(f4_sched_ok(), "winograd4: chunk schedule incomplete");
// vector-memory operations (B ring loads, patch loads) the schedule issues in slots [lo, hi)
constexpr int f4_vm_between(int lo, int hi) {
  int n = 0;
  for (int i = lo < 0 ? 0 : lo; i < hi && i < 144; ++i) n += (F4S.kind[i] == 1 || F4S.kind[i] == 2) ? 1 : 0;
  return n;
}
constexpr int f4_slot_of(int kind, int idx) {
  for (int i = 0; i < 144; ++i) if (F4S.kind[i] == kind && F4S.idx[i] == idx) return i;
  return -1;
}
// operations younger than the load of position pair q when its first MFMA (slot 8 q) issues: the pair was loaded by G_B(q - F4_NBR)
// of this chunk or G_B(q + 18 - F4_NBR) of the previous one (the prologue issues pairs 0 .. F4_NBR - 1 in the same order, behind its patch loads)
constexpr int f4_younger_b(int q) {
  return q >= F4_NBR ? f4_vm_between(f4_slot_of(1, q - F4_NBR) + 1, 8 * q)
                     : f4_vm_between(f4_slot_of(1, q + 18 - F4_NBR) + 1, 144) + f4_vm_between(0, 8 * q);
}
// operations younger than the load of patch item q when S_raw(q) stores it, first chunk of a block (see rload)
constexpr int f4_younger_r(int q) { return (F4_NQ - 1 - q) + F4_NBR + f4_vm_between(0, f4_slot_of(3, q)); }

// EPI: 0 plain (input gradient), 1 BatchNorm statistics (training forward), 2 LeakyReLU (eval forward, BatchNorm folded)
template <int EPI>
__global__ __launch_bounds__(256, 1) void wino4_conv_kernel(Wino4Args a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* Vs = smem;                             // [2][F4_V_BUF]
  float* Rs = smem + 2 * F4_V_BUF;              // [2][F4_RAW_BUF]

  const int t = threadIdx.x, lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);

  unsigned vid = blockIdx.x;
  if ((gridDim.x & 7u) == 0) vid = (blockIdx.x & 7u) * (gridDim.x >> 3) + (blockIdx.x >> 3);
  const int nblk = a.Np / 64;
  const int ntile_mine = (a.ntiles - (int)vid + (int)gridDim.x - 1) / (int)gridDim.x;   // >= 1 (grid <= ntiles)
  const int nchunk = a.Cin / 8;
  struct TilePos { int nb, b, oy0, ox0; };
  auto tile_pos = [&](int k) {                  // k-th tile of this block (uniform)
    const int id = (int)vid + k * (int)gridDim.x;
    TilePos p;
    p.nb = id % nblk;
    int rest = id / nblk;
    const int tbx = rest % a.tbw; rest /= a.tbw;
    const int tby = rest % a.tbh;
    p.b = rest / a.tbh; p.oy0 = tby * 16; p.ox0 = tbx * 32;
    return p;
  };

  // ---- patch loader: item = t + 256 q -> 16 pixels x 2 k-quads per 32 items (conflict-free b128 LDS stores, both
  // 16-byte halves of a pixel's 32 bytes in one wave-load); pix(q) = pix0 + 128 q; only the last round is partial
  // (its missing items repeat the thread's previous item: no exec masks in the loop).
  const int kq_of_thread = (t >> 4) & 1;
  const int pix0 = (t >> 5) * 16 + (t & 15);
  const int roff0 = (kq_of_thread * F4_RAWP + pix0) * 4;
  const bool rlast_ok = pix0 + 128 * (F4_NQ - 1) < F4_NPIX;
  const int roff4 = roff0 + 512 * (rlast_ok ? F4_NQ - 1 : F4_NQ - 2);
  unsigned goff[F4_NQ];                         // byte offset of the item's pixel, channel quad, from the image base; padding: 2^31
  i32x4_ xdesc = {0, 0, 0, 0};                  // uniform: buffer descriptor of the patch cursor's image
  const int img_bytes = a.H * a.W * a.Cin * 4;
  auto set_raw_tile = [&](int k) {
    const TilePos p = tile_pos(k);
    const unsigned long long xb = (unsigned long long)(uintptr_t)(a.X + (long long)p.b * a.H * a.W * a.Cin);
    xdesc = i32x4_{(int)(unsigned)xb, (int)(unsigned)((xb >> 32) & 0xffffu), img_bytes, 0x00020000};
#pragma unroll
    for (int q = 0; q < F4_NQ; ++q) {
      const int pix = pix0 + 128 * ((q < F4_NQ - 1 || rlast_ok) ? q : q - 1);
      const int pr = pix / F4_PC, pc = pix - pr * F4_PC;
      const int iy = p.oy0 - 1 + pr, ix = p.ox0 - 1 + pc;
      const bool ok = (unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W;
      goff[q] = ok ? (unsigned)(((iy * a.W + ix) * a.Cin + kq_of_thread * 4) * 4) : 0x80000000u;
    }
  };
  auto advance = [&](int& k, int& c) {          // one chunk further; stops at the very last chunk of the block's stream
    if (c + 1 < nchunk) { ++c; return false; }
    if (k + 1 < ntile_mine) { ++k; c = 0; return true; }
    return false;
  };
  int kr = 0, cr = 0;                           // patch cursor
  f32x4 graw[F4_NQ];
  auto Graw1 = [&](int q, int c, f32x4& dst) { rload(dst, xdesc, goff[q], (unsigned)c * 32u); };
  auto Sraw1 = [&](float* rb, int q, const f32x4& src) { *(f32x4*)(rb + (q < F4_NQ - 1 ? roff0 + 512 * q : roff4)) = src; };

  // ---- B operand stream: U[nb][chunk][wave][pair 18][lane][4]
  const long long u_wave = 18 * 1024;                                  // bytes per (chunk, wave)
  unsigned ulane[5];                                                   // the lane's 16 bytes of every 1 KiB pair image, per 4 KiB window
#pragma unroll
  for (int k = 0; k < 5; ++k) ulane[k] = (unsigned)lane * 16u + 4096u * k;
  auto u_ptr = [&](int k, int c) {                                     // uniform: this wave's 18 KiB of (tile k, chunk c)
    return (const char*)a.U + (((long long)tile_pos(k).nb * nchunk + c) * 4 + wave) * u_wave;
  };
  int ku = 0, cu = 0;
  f32x4 bq[F4_NBR];

  // ---- transform item: rows 3 hr .. 3 hr + 2 of V for tile (ty = wave, tx), channel pair kg
  const int hr = t & 1, kg = (t >> 1) & 3, ttx = (t >> 3) & 7, tty = t >> 6;
  const int tbase = ((kg >> 1) * F4_RAWP + (4 * tty) * F4_PC + 4 * ttx) * 4 + 2 * (kg & 1);   // patch pixel (0, 0)
  const int tbase_e = tbase + hr * F4_PC * 4;                                                   // rows hr, 2 + hr, 4 + hr
  const int tslot = ((tty & 1) * 8 + ttx) ^ ((kg << 1) | (hr << 3));
  const int vdst = (kg * 16 + tslot) * 4 + (tty >> 1) * 2;
  // outputs of the column pass: o0 = 4 e0 - 5 e1 + e2 (row 0 / 5), o1 / o2 = X +- gamma Y (rows 1, 2 / 3, 4)
  const int vd0 = vdst + (hr ? 30 : 0) * 256, vd1 = vdst + (hr ? 18 : 6) * 256, vd2 = vdst + (hr ? 24 : 12) * 256;
  const float alpha_ = hr ? -1.f : -4.f, gamma_ = hr ? 2.f : 1.f;
  const f32x2 kal = {alpha_, alpha_}, kga = {gamma_, gamma_};
  const f32x2 k4 = {4.f, 4.f}, km5 = {-5.f, -5.f}, km4 = {-4.f, -4.f}, k2 = {2.f, 2.f};
  f32x2 tt[3][6];                               // T = B^T d, the thread's three rows
  f32x2 dc[2][7];                               // one patch column: e0, e1, e2, d1, d2, d3, d4
  auto Trd = [&](const float* rb, int c) {
    f32x2* d = dc[c & 1];
    d[0] = *(const f32x2*)(rb + tbase_e + (0 * F4_PC + c) * 4);
    d[1] = *(const f32x2*)(rb + tbase_e + (2 * F4_PC + c) * 4);
    d[2] = *(const f32x2*)(rb + tbase_e + (4 * F4_PC + c) * 4);
    d[3] = *(const f32x2*)(rb + tbase + (1 * F4_PC + c) * 4);
    d[4] = *(const f32x2*)(rb + tbase + (2 * F4_PC + c) * 4);
    d[5] = *(const f32x2*)(rb + tbase + (3 * F4_PC + c) * 4);
    d[6] = *(const f32x2*)(rb + tbase + (4 * F4_PC + c) * 4);
  };
  // Every piece holds mutually INDEPENDENT packed operations: gfx950 needs a wait state between a packed fp32 operation and a
  // dependent one (hipcc pads with s_nop, an issue slot next to the MFMAs), so the dependent halves stand in later slots.
  f32x2 cx_, cy_, ci_;
  auto Tcol = [&](int c, int part) {
    const f32x2* d = dc[c & 1];
    if (part == 0) {
      cx_ = pkfma(kal, d[4], d[6]);             // X = d4 + alpha d2
      cy_ = pkfma(kal, d[3], d[5]);             // Y = d3 + alpha d1
      ci_ = pkfma(km5, d[1], d[2]);
    } else {
      tt[1][c] = pkfma(kga, cy_, cx_);
      tt[2][c] = pkfnma(kga, cy_, cx_);
      tt[0][c] = pkfma(k4, d[0], ci_);
    }
  };
  f32x2 rt_[6];                                 // first halves of a row: i0, X1, Y1, X2, Y2, i5
  auto Trow = [&](float* vb, int r, int part) { // row r (of the thread's three)
    const f32x2* x = tt[r];
    float* v = vb + (r == 0 ? vd0 : r == 1 ? vd1 : vd2);
    if (part == 0) {
      rt_[0] = pkfma(km5, x[2], x[4]);
      rt_[1] = pkfma(km4, x[2], x[4]);
      rt_[2] = pkfma(km4, x[1], x[3]);
    } else if (part == 1) {
      rt_[3] = pksub(x[4], x[2]);
      rt_[4] = pksub(x[3], x[1]);
      rt_[5] = pkfma(km5, x[3], x[5]);
    } else if (part == 2) {
      *(f32x2*)(v + 0 * 256) = pkfma(k4, x[0], rt_[0]);
      *(f32x2*)(v + 1 * 256) = pkadd(rt_[1], rt_[2]);
      *(f32x2*)(v + 2 * 256) = pksub(rt_[1], rt_[2]);
    } else {
      *(f32x2*)(v + 3 * 256) = pkfma(k2, rt_[4], rt_[3]);
      *(f32x2*)(v + 4 * 256) = pkfnma(k2, rt_[4], rt_[3]);
      *(f32x2*)(v + 5 * 256) = pkfma(k4, x[1], rt_[5]);
    }
  };
  auto Tall = [&](int buf_raw, int buf_v) {
    const float* rb = Rs + buf_raw * F4_RAW_BUF;
    float* vb = Vs + buf_v * F4_V_BUF;
#pragma unroll
    for (int c = 0; c < 6; ++c) { Trd(rb, c); Tcol(c, 0); Tcol(c, 1); }
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
      for (int part = 0; part < 4; ++part) Trow(vb, r, part);
  };

  // ---- prologue (once per block).  State at the top of stream position f (tile km, chunk cm): V[f&1] = position f,
  // raw[(f+1)&1] = patch of f+1, graw = patch of f+2, patch cursor (kr, cr) at f+3; bq = position pairs 0 .. F4_NBR - 1 of f,
  // up_cur / up_nxt = this wave's U of positions f / f+1.
  {
    f32x4 graw1[F4_NQ];
    set_raw_tile(0);
#pragma unroll
    for (int q = 0; q < F4_NQ; ++q) Graw1(q, 0, graw[q]);
#pragma unroll
    for (int q = 0; q < F4_NQ; ++q) rwait0(graw[q]);
#pragma unroll
    for (int q = 0; q < F4_NQ; ++q) Sraw1(Rs, q, graw[q]);
    if (advance(kr, cr)) set_raw_tile(kr);
#pragma unroll
    for (int q = 0; q < F4_NQ; ++q) Graw1(q, cr, graw1[q]);
    __syncthreads();
    Tall(0, 0);
#pragma unroll
    for (int q = 0; q < F4_NQ; ++q) rwait0(graw1[q]);
#pragma unroll
    for (int q = 0; q < F4_NQ; ++q) Sraw1(Rs + F4_RAW_BUF, q, graw1[q]);
    if (advance(kr, cr)) set_raw_tile(kr);
#pragma unroll
    for (int q = 0; q < F4_NQ; ++q) Graw1(q, cr, graw[q]);
    __syncthreads();
    if (advance(kr, cr)) set_raw_tile(kr);
  }
  const char* up_cur = u_ptr(0, 0);
  if (advance(ku, cu)) {}
  const char* up_nxt = u_ptr(ku, cu);
#pragma unroll
  for (int q = 0; q < F4_NBR; ++q) {
    if (F4DBG & 4) bq[q] = f32x4{1.f, 1.f, 1.f, 1.f};
    else bload<0>(bq[q], up_cur + (q & 3) * 1024, ulane[q >> 2]);
  }

  // A fragment of (position, tile half): 8 bytes at ((pos * 2 + half) * 4 + kgl) * 16 + (m ^ swizzle)
  const int kgl = lane >> 4, ml = lane & 15;
  const int fragA_lo = (kgl * 16 + (ml ^ (kgl << 1))) * 4;                  // positions 0..17
  const int fragA_hi = (kgl * 16 + (ml ^ ((kgl << 1) | 8))) * 4;            // positions 18..35

  f32x4 fa[3];                                  // A fragments (both tile halves x 2 k-steps) of positions p % 3; [0] / [1] are carried into the next chunk
  fa[0] = *(const f32x4*)(Vs + fragA_lo);
  fa[1] = *(const f32x4*)(Vs + fragA_lo + 256);
  int c_next = 0;
  for (int km = 0; km < ntile_mine; ++km) {
    f32x4 accA[32][2], accV[4][2];
#pragma unroll
    for (int p = 0; p < 32; ++p)
#pragma unroll
      for (int h = 0; h < 2; ++h) accA[p][h] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int p = 0; p < 4; ++p)
#pragma unroll
      for (int h = 0; h < 2; ++h) accV[p][h] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int cm = 0; cm < nchunk; ++cm, ++c_next) {
      const int c = c_next;
      const float* va_ = Vs + (c & 1) * F4_V_BUF;
      const float* rb_ = Rs + ((c + 1) & 1) * F4_RAW_BUF;         // T(f+1) reads ...
      float* vw_ = Vs + ((c + 1) & 1) * F4_V_BUF;                 // ... and writes
      float* rw_ = Rs + (c & 1) * F4_RAW_BUF;                     // S_raw(f+2)
#define F4_BARRIER_HERE if (s_ == F4_BAR) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
#ifdef CY_F4_PROF
      unsigned long long st_[8];
      st_[0] = __builtin_amdgcn_s_memtime();
#define F4_STAMP(s_) if ((s_) == 48) st_[1] = __builtin_amdgcn_s_memtime(); if ((s_) == 96) st_[2] = __builtin_amdgcn_s_memtime(); \
                     if ((s_) == 136) st_[3] = __builtin_amdgcn_s_memtime(); if ((s_) == 137) st_[4] = __builtin_amdgcn_s_memtime();
#else
#define F4_STAMP(s_)
#endif
#define F4SLOT(SIDX)                                                                                  \
      {                                                                                               \
        constexpr int s_ = (SIDX), p_ = s_ >> 2, w_ = s_ & 3, h_ = w_ & 1, ks_ = w_ >> 1;             \
        constexpr int q_ = p_ >> 1, br_ = q_ % F4_NBR;                                                \
        F4_STAMP(s_)                                                                                  \
        F4_BARRIER_HERE                                                                               \
        if (w_ == 0 && (p_ & 1) == 0 && !(F4DBG & 4)) vmwait<f4_younger_b(q_)>(bq[br_]);              \
        if (p_ < 32) mfma16_a(accA[p_ < 32 ? p_ : 0][h_], fa[p_ % 3][2 * h_ + ks_], bq[br_][2 * (p_ & 1) + ks_]); \
        else mfma16_v(accV[p_ >= 32 ? p_ - 32 : 0][h_], fa[p_ % 3][2 * h_ + ks_], bq[br_][2 * (p_ & 1) + ks_]);   \
        if (w_ == 0 && p_ + 2 < 36) {                                                                 \
          constexpr int np_ = p_ + 2 < 36 ? p_ + 2 : 0;                                               \
          fa[np_ % 3] = *(const f32x4*)(va_ + (np_ >= 18 ? fragA_hi : fragA_lo) + np_ * 256);         \
        } else if (w_ == 0) {                   /* behind the barrier: positions 0 / 1 of the NEXT chunk (fa[1] is free after slot 139) */ \
          constexpr int np_ = p_ >= 34 ? p_ - 34 : 0;                                                             \
          fa[np_] = *(const f32x4*)(vw_ + fragA_lo + np_ * 256);                                      \
        }                                                                                             \
        constexpr int kind0_ = F4S.kind[s_], k_ = F4S.idx[s_];                                        \
        constexpr int kind = (((F4DBG & 1) && (kind0_ == 5 || kind0_ == 6)) || ((F4DBG & 2) && (kind0_ == 2 || kind0_ == 3)) || \
                              ((F4DBG & 4) && kind0_ == 1) || ((F4DBG & 16) && kind0_ == 4)) ? 0 : kind0_;        \
        if (kind == 1) {                        /* B operand of pair k_ + 6 into the ring slot pair k_ just left */ \
          constexpr int nq_ = k_ + F4_NBR;                                                            \
          constexpr int lq_ = nq_ < 18 ? nq_ : nq_ - 18;                                              \
          bload<(lq_ & 3) * 1024>(bq[k_ % F4_NBR], nq_ < 18 ? up_cur : up_nxt, ulane[lq_ >> 2]);      \
        } else if (kind == 3) {                                                                       \
          vmwait<f4_younger_r(k_ % F4_NQ)>(graw[k_ % F4_NQ]);                                                                    \
          Sraw1(rw_, k_ % F4_NQ, graw[k_ % F4_NQ]);                                                         \
        } else if (kind == 2) {                                                                       \
          Graw1(k_ % F4_NQ, cr, graw[k_ % F4_NQ]);                                                                \
        } else if (kind == 4) {                                                                       \
          Trd(rb_, k_);                                                                               \
        } else if (kind == 5) {                                                                       \
          Tcol(k_ >> 1, k_ & 1);                                                                      \
        } else if (kind == 6) {                                                                       \
          Trow(vw_, k_ >> 2, k_ & 3);                                                                 \
        } else if (kind == 7) {                 /* graw now holds f+3; patch cursor -> f+4 */         \
          if (advance(kr, cr)) set_raw_tile(kr);                                                      \
        }                                                                                             \
        __builtin_amdgcn_sched_barrier(0);                                                            \
      }
#define F4SLOT8(B) F4SLOT((B)) F4SLOT((B) + 1) F4SLOT((B) + 2) F4SLOT((B) + 3) F4SLOT((B) + 4) F4SLOT((B) + 5) F4SLOT((B) + 6) F4SLOT((B) + 7)
      F4SLOT8(0) F4SLOT8(8) F4SLOT8(16) F4SLOT8(24) F4SLOT8(32) F4SLOT8(40) F4SLOT8(48) F4SLOT8(56) F4SLOT8(64)
      F4SLOT8(72) F4SLOT8(80) F4SLOT8(88) F4SLOT8(96) F4SLOT8(104) F4SLOT8(112) F4SLOT8(120) F4SLOT8(128) F4SLOT8(136)
#undef F4SLOT8
#undef F4SLOT
#undef F4_BARRIER_HERE
#undef F4_STAMP
#ifdef CY_F4_PROF
      st_[5] = __builtin_amdgcn_s_memtime();
#endif
      up_cur = up_nxt;                          // U stream: positions f+1 / f+2
      if (advance(ku, cu)) {}
      up_nxt = u_ptr(ku, cu);
#ifdef CY_F4_PROF
      st_[6] = __builtin_amdgcn_s_memtime();
      if (t == 0 && cm == 5 && km == 3) {
        unsigned long long* pb = f4_prof_buf + blockIdx.x * 16;
        for (int i = 0; i < 7; ++i) pb[i] = st_[i];
      }
#endif
    }
#ifdef CY_F4_PROF
    const unsigned long long dr0_ = __builtin_amdgcn_s_memtime();
#endif
    // ======== tile km is complete: drain the accumulators.  V[cl&1] (cl = the tile's last position) was consumed and
    // is free until the barrier at the end of the drain: 9 KiB of it per wave are the drain's scratch.
    const int cl = c_next - 1;
    const TilePos tp = tile_pos(km);
    float* ow = Vs + (cl & 1) * F4_V_BUF + wave * (2 * F4_OSTEP);
    const int g_ = lane >> 4, co16 = lane & 15;
    const int co = tp.nb * 64 + wave * 16 + co16;
    float bv = 0.f;                             // (one self-contained statement: a tracked load inside the tile loop made hipcc
    if (a.bias != nullptr) {                    //  wait vmcnt(0) in the chunk loop, where its register is a temporary)
      const float* bp = a.bias + (co < a.Cout ? co : 0);
      asm volatile("global_load_dword %0, %1, off\n\ts_waitcnt vmcnt(0)" : "=&v"(bv) : "v"(bp));
      if (co >= a.Cout) bv = 0.f;
    }
    const bool full = tp.oy0 + 16 <= a.H && tp.ox0 + 32 <= a.W && tp.nb * 64 + 64 <= a.Cout && (a.Cout & 3) == 0;   // uniform
    const int pxl = lane >> 2, cq = lane & 3;                     // read-back: pixel (y, x) = (pxl >> 2, pxl & 3), channel quad
    const int cbase = tp.nb * 64 + wave * 16 + cq * 4;
    const unsigned lane_off = (unsigned)(((pxl >> 2) * a.W + (pxl & 3)) * a.Cout + cq * 4);
    float* ybase = a.Y + (((long long)tp.b * a.H + tp.oy0) * a.W + tp.ox0) * a.Cout + tp.nb * 64 + wave * 16;   // uniform
    float ssum = 0.f, ssq = 0.f;
    auto drain = [&](auto full_c) {
      constexpr bool FULL = decltype(full_c)::value;
#pragma unroll
      for (int st = 0; st < 8; ++st) {
        const int h = st >> 2, r = st & 3;
        float* os = ow + (st & 1) * F4_OSTEP;
        // lane-local output transform of tile T = 16 h + 4 g + r, channel co16
        float S[6][4];
#pragma unroll
        for (int i = 0; i < 6; ++i) {
          float m[6];
#pragma unroll
          for (int j = 0; j < 6; ++j) {
            const int p = 6 * i + j;
            m[j] = p < 32 ? acc_elem4(accA[p < 32 ? p : 0][h][r]) : accV[p >= 32 ? p - 32 : 0][h][r];
          }
          const float s12 = m[1] + m[2], d12 = m[1] - m[2], s34 = m[3] + m[4], d34 = m[3] - m[4];
          S[i][0] = (m[0] + s12) + s34;
          S[i][1] = __builtin_fmaf(2.f, d34, d12);
          S[i][2] = __builtin_fmaf(4.f, s34, s12);
          S[i][3] = __builtin_fmaf(8.f, d34, d12) + m[5];
        }
        const int T = 16 * h + 4 * g_ + r;
        const int oyt = tp.oy0 + 4 * (T >> 3), oxt = tp.ox0 + 4 * (T & 7);
#pragma unroll
        for (int x = 0; x < 4; ++x) {
          const float s12 = S[1][x] + S[2][x], d12 = S[1][x] - S[2][x], s34 = S[3][x] + S[4][x], d34 = S[3][x] - S[4][x];
          float y[4];
          y[0] = (S[0][x] + s12) + s34 + bv;
          y[1] = __builtin_fmaf(2.f, d34, d12) + bv;
          y[2] = __builtin_fmaf(4.f, s34, s12) + bv;
          y[3] = __builtin_fmaf(8.f, d34, d12) + S[5][x] + bv;
#pragma unroll
          for (int yy = 0; yy < 4; ++yy) {
            float v = y[yy];
            if constexpr (EPI == 2) v = fmaxf(v, v * a.out_slope);
            os[g_ * F4_OG + (yy * 4 + x) * 16 + co16] = v;
            if constexpr (EPI == 1) {
              if (FULL || (co < a.Cout && oyt + yy < a.H && oxt + x < a.W)) { ssum += v; ssq = __builtin_fmaf(v, v, ssq); }
            }
          }
        }
        // read back 16 pixels x 16 channels per lane group j: tile 16 h + 4 j + r, 16-byte stores
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int Tj = 16 * h + 4 * j + r;
          const f32x4 v = *(const f32x4*)(os + j * F4_OG + pxl * 16 + cq * 4);
          float* yp = ybase + ((long long)(4 * (Tj >> 3)) * a.W + 4 * (Tj & 7)) * a.Cout + lane_off;
          if (FULL) {
            *(f32x4*)yp = v;
          } else {
            const int oy = tp.oy0 + 4 * (Tj >> 3) + (pxl >> 2), ox = tp.ox0 + 4 * (Tj & 7) + (pxl & 3);
            if (oy < a.H && ox < a.W) {
              if ((a.Cout & 3) == 0 && cbase + 3 < a.Cout) *(f32x4*)yp = v;
              else {
#pragma unroll
                for (int k = 0; k < 4; ++k) if (cbase + k < a.Cout) yp[k] = v[k];
              }
            }
          }
        }
      }
    };
    // Blocks inside the image (all of them at 416 x 416 and 608 x 608): two tiles (accumulator registers r, r + 1) per step as
    // float2 -- one wave per SIMD issues one vector instruction per 4 cycles, packed or not, so v_pk_* halves the drain's
    // arithmetic time (transform, bias, statistics); no range checks.
    auto drain_full = [&]() {
      const f32x2 k2 = {2.f, 2.f}, k4 = {4.f, 4.f}, k8 = {8.f, 8.f}, bv2 = {bv, bv};
      f32x2 ssum2 = {0.f, 0.f}, ssq2 = {0.f, 0.f};
#pragma unroll
      for (int sp = 0; sp < 4; ++sp) {
        const int h = sp >> 1, r0 = 2 * (sp & 1);
        f32x2 S[6][4];
#pragma unroll
        for (int i = 0; i < 6; ++i) {
          f32x2 m[6];
#pragma unroll
          for (int j = 0; j < 6; ++j) {
            const int p = 6 * i + j;
            m[j][0] = p < 32 ? acc_elem4(accA[p < 32 ? p : 0][h][r0]) : accV[p >= 32 ? p - 32 : 0][h][r0];
            m[j][1] = p < 32 ? acc_elem4(accA[p < 32 ? p : 0][h][r0 + 1]) : accV[p >= 32 ? p - 32 : 0][h][r0 + 1];
          }
          const f32x2 s12 = pkadd(m[1], m[2]), d12 = pksub(m[1], m[2]), s34 = pkadd(m[3], m[4]), d34 = pksub(m[3], m[4]);
          S[i][0] = pkadd(pkadd(m[0], s12), s34);
          S[i][1] = pkfma(k2, d34, d12);
          S[i][2] = pkfma(k4, s34, s12);
          S[i][3] = pkadd(pkfma(k8, d34, d12), m[5]);
        }
#pragma unroll
        for (int x = 0; x < 4; ++x) {
          const f32x2 s12 = pkadd(pkadd(S[1][x], S[2][x]), bv2), d12 = pkadd(pksub(S[1][x], S[2][x]), bv2);   // (the bias rides on them)
          const f32x2 s34 = pkadd(S[3][x], S[4][x]), d34 = pksub(S[3][x], S[4][x]);
          f32x2 y[4];
          y[0] = pkadd(pkadd(S[0][x], s12), s34);
          y[1] = pkfma(k2, d34, d12);
          y[2] = pkfma(k4, s34, s12);
          y[3] = pkadd(pkfma(k8, d34, d12), S[5][x]);
#pragma unroll
          for (int yy = 0; yy < 4; ++yy) {
            f32x2 v = y[yy];
            if constexpr (EPI == 2) { v[0] = fmaxf(v[0], v[0] * a.out_slope); v[1] = fmaxf(v[1], v[1] * a.out_slope); }
            ow[g_ * F4_OG + (yy * 4 + x) * 16 + co16] = v[0];
            ow[F4_OSTEP + g_ * F4_OG + (yy * 4 + x) * 16 + co16] = v[1];
            if constexpr (EPI == 1) { ssum2 = pkadd(ssum2, v); ssq2 = pkfma(v, v, ssq2); }
          }
        }
#pragma unroll
        for (int rr = 0; rr < 2; ++rr)
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const int Tj = 16 * h + 4 * j + r0 + rr;
            const f32x4 v = *(const f32x4*)(ow + rr * F4_OSTEP + j * F4_OG + pxl * 16 + cq * 4);
            float* yp = ybase + ((long long)(4 * (Tj >> 3)) * a.W + 4 * (Tj & 7)) * a.Cout + lane_off;
            *(f32x4*)yp = v;
          }
      }
      ssum = ssum2[0] + ssum2[1]; ssq = ssq2[0] + ssq2[1];
    };
    if (!(F4DBG & 8)) { if (full) drain_full(); else drain(std::false_type{}); }
    if constexpr (EPI == 1) {
      ssum += __shfl_xor(ssum, 16, 64); ssq += __shfl_xor(ssq, 16, 64);
      ssum += __shfl_xor(ssum, 32, 64); ssq += __shfl_xor(ssq, 32, 64);
      if (lane < 16 && co < a.Cout) {
        double* st = a.stats + (size_t)(blockIdx.x % CY_STATS_COPIES) * a.Cout * 2;
        atomicAdd(st + 2 * co, (double)ssum);
        atomicAdd(st + 2 * co + 1, (double)ssq);
      }
    }
#ifdef CY_F4_PROF
    const unsigned long long dr1_ = __builtin_amdgcn_s_memtime();
#endif
    __syncthreads();                            // the scratch is rewritten by the next position's transform
#ifdef CY_F4_PROF
    if (t == 0 && km == 3) {
      unsigned long long* pb = f4_prof_buf + blockIdx.x * 16;
      pb[8] = dr0_; pb[9] = dr1_; pb[10] = __builtin_amdgcn_s_memtime();
    }
#endif
  }
}

// U[nb][chunk][wave][pair q][lane l][e]: e = 2 (pos & 1) + s, pos = 2 q + (e >> 1) = 6 i + j, co = 64 nb + 16 wave + (l & 15),
// ci = 8 chunk + 2 (l >> 4) + s;  value (G g G^T)[i][j], g = W[co][ci] (transpose: the input-gradient operand, 180-degree rotated)
__global__ void wino4_pack_kernel(const float* __restrict__ W, float* __restrict__ U, int Cout_l, int Cin_l, int transpose,
                                  int Cin, int Np, int Ncols, long long total) {
  const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;   // one (ci, co) per thread
  if (idx >= total) return;
  const int co = (int)(idx % Np), ci = (int)(idx / Np);
  (void)Cout_l;
  float g[3][3];
  const bool ok = ci < Cin && co < Ncols;
#pragma unroll
  for (int kh = 0; kh < 3; ++kh)
#pragma unroll
    for (int kw = 0; kw < 3; ++kw)
      g[kh][kw] = !ok ? 0.f : transpose ? W[(((long long)ci * Cin_l + co) * 3 + (2 - kh)) * 3 + (2 - kw)]
                                        : W[(((long long)co * Cin_l + ci) * 3 + kh) * 3 + kw];
  const float Gm[6][3] = {{0.25f, 0.f, 0.f}, {-1.f / 6, -1.f / 6, -1.f / 6}, {-1.f / 6, 1.f / 6, -1.f / 6},
                          {1.f / 24, 1.f / 12, 1.f / 6}, {1.f / 24, -1.f / 12, 1.f / 6}, {0.f, 0.f, 1.f}};
  const int nb = co >> 6, wv = (co >> 4) & 3, c16 = co & 15;
  const int chunk = ci >> 3, kgp = (ci >> 1) & 3, s = ci & 1;
  const int nchunk = Cin / 8;
  float* out = U + ((((long long)nb * nchunk + chunk) * 4 + wv) * 18) * 256 + (kgp * 16 + c16) * 4;
  float tmp[6][3];
#pragma unroll
  for (int i = 0; i < 6; ++i)
#pragma unroll
    for (int q = 0; q < 3; ++q) tmp[i][q] = Gm[i][0] * g[0][q] + Gm[i][1] * g[1][q] + Gm[i][2] * g[2][q];
#pragma unroll
  for (int i = 0; i < 6; ++i)
#pragma unroll
    for (int j = 0; j < 6; ++j) {
      const int pos = 6 * i + j;
      const float u = tmp[i][0] * Gm[j][0] + tmp[i][1] * Gm[j][1] + tmp[i][2] * Gm[j][2];
      out[(pos >> 1) * 256 + 2 * (pos & 1) + s] = u;
    }
}

}  // namespace

extern "C" long long cy_wino4_packed_floats(int Cin, int N) {
  return (long long)((Cin + 7) / 8 * 8) * 36 * ((N + 63) / 64 * 64);
}

extern "C" int cy_wino4_pack_weights(const float* W, float* U, int Cout, int Cin, int transpose, void* stream) {
  CY_REQUIRE(W && U && Cout > 0 && Cin > 0, "cy_wino4_pack_weights: bad arguments");
  const int cin_g = transpose ? Cout : Cin, n_g = transpose ? Cin : Cout;
  CY_REQUIRE(cin_g % 8 == 0, "cy_wino4_pack_weights: the reduction channels (%d) must be a multiple of 8", cin_g);
  const int Np = (n_g + 63) / 64 * 64;
  const long long total = (long long)cin_g * Np;
  wino4_pack_kernel<<<(unsigned)cy_ceil_div(total, 256), 256, 0, (hipStream_t)stream>>>(W, U, Cout, Cin, transpose, cin_g, Np,
                                                                                       n_g, total);
  CY_LAUNCH_CHECK("cy_wino4_pack_weights");
  return 0;
}

extern "C" int cy_conv3x3_winograd4(const float* X, const float* U, float* Y, const float* bias, double* stats, float out_slope,
                                    int B, int H, int W, int Cin, int Cout, void* stream) {
  CY_REQUIRE(X && U && Y && B > 0 && H > 0 && W > 0 && Cout > 0, "cy_conv3x3_winograd4: bad arguments");
  CY_REQUIRE(out_slope >= 0.f && out_slope <= 1.f, "cy_conv3x3_winograd4: out_slope=%g must be in [0, 1] (1 = no activation)", (double)out_slope);
  CY_REQUIRE(out_slope == 1.f || stats == nullptr, "cy_conv3x3_winograd4: the activation epilogue is for eval-mode forwards (no statistics)");
  CY_REQUIRE(Cin % 8 == 0 && Cin >= 8, "cy_conv3x3_winograd4: Cin=%d must be a multiple of 8", Cin);
  CY_REQUIRE((((uintptr_t)X | (uintptr_t)U | (uintptr_t)Y) & 15) == 0, "cy_conv3x3_winograd4: operands must be 16-byte aligned");
  CY_REQUIRE((long long)H * W * Cin < (1ll << 29) && (long long)H * W * Cout < (1ll << 29),
             "cy_conv3x3_winograd4: image too large for 32-bit byte offsets");
  Wino4Args a;
  a.X = X; a.U = U; a.Y = Y; a.bias = bias; a.stats = stats; a.out_slope = out_slope;
  a.B = B; a.H = H; a.W = W; a.Cin = Cin; a.Cout = Cout;
  a.Np = (Cout + 63) / 64 * 64;
  a.tbh = (H + 15) / 16; a.tbw = (W + 31) / 32;
  const long long tiles = (long long)B * a.tbh * a.tbw * (a.Np / 64);
  CY_REQUIRE(tiles < (1ll << 31), "cy_conv3x3_winograd4: too many tiles");
  a.ntiles = (int)tiles;
  int dev = 0, ncu = 0;
  hipError_t he = hipGetDevice(&dev);
  if (he == hipSuccess) he = hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev);
  if (he != hipSuccess || ncu <= 0) return cy_set_error((int)he, "cy_conv3x3_winograd4: cannot query the CU count: %s", hipGetErrorString(he));
  const long long blocks = tiles < ncu ? tiles : ncu;   // persistent: one block per CU (512 registers per lane)
  const size_t lds = (size_t)(2 * F4_V_BUF + 2 * F4_RAW_BUF) * 4;
  int rc = cy_allow_lds(wino4_conv_kernel<1>, lds);
  if (rc) return rc;
  rc = cy_allow_lds(wino4_conv_kernel<0>, lds);
  if (rc) return rc;
  rc = cy_allow_lds(wino4_conv_kernel<2>, lds);
  if (rc) return rc;
  if (a.stats != nullptr) wino4_conv_kernel<1><<<(unsigned)blocks, 256, lds, (hipStream_t)stream>>>(a);
  else if (out_slope != 1.f) wino4_conv_kernel<2><<<(unsigned)blocks, 256, lds, (hipStream_t)stream>>>(a);
  else wino4_conv_kernel<0><<<(unsigned)blocks, 256, lds, (hipStream_t)stream>>>(a);
  CY_LAUNCH_CHECK("cy_conv3x3_winograd4");
  return 0;
}

#ifdef CY_F4_PROF
extern "C" int cy_wino4_read_prof(unsigned long long* host_dst) {
  return (int)hipMemcpyFromSymbol(host_dst, HIP_SYMBOL(f4_prof_buf), sizeof(unsigned long long) * 256 * 16);
}
#endif
