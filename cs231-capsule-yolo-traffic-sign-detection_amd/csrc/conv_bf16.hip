// bf16 convolution path (params.json "precision": "bf16"; BASELINE configs[4]): nn.Conv2d forward and input gradient
// of models.py:347-365 as implicit GEMMs on v_mfma_f32_32x32x16_bf16 (fp32 accumulation), activations stored as
// bf16 NHWC, BatchNorm statistics and master weights in fp32 / double exactly as on the fp32 path (gfx950).
//
//   Y[pixel][n] = sum over taps (a,b) and channels c of X[b, oy*s + dy0 + a*dstep, ox*s + dx0 + b*dstep, c] * Wp[n][(a*TW + b)*Cin + c]
//
// One description (cy_conv_gemm_t, shared with the fp32 kernel) drives the forward and, per output-parity class of the
// stride, the input gradient.  Block tile 256 x 256 (8 waves of 128 x 64), 256 x 128 or 128 x 64, K step 64, PERSISTENT blocks:
//  * both operands of a K step go global -> LDS by LDS-DMA (global_load_lds_dwordx4), one step ahead, into the other of two
//    LDS buffers; rows are 128 bytes, unpadded, with the 16-byte chunk c of row r stored at chunk c ^ ((r >> 1) & 7) (applied
//    on the DMA's per-lane source address and on the fragment reads): conflict-free ds_read_b128; a lane (r = lane & 31,
//    h = lane >> 5) fetches its MFMA fragment A[r][8h .. 8h+7] with ONE ds_read_b128, the weights are packed [n][k] so that
//    B fragments read the same way; fragment reads are inline asm, one 16-k slice ahead of their MFMAs, counted waits;
//  * one barrier per K step; the first K step of the NEXT tile is requested during the last step of the current one and
//    waited for in front of the epilogue's first store (a wait behind the stores would sit out their round trip);
//  * epilogue: bias (one load per tile), BatchNorm sums (fp32 per lane -> LDS -> one double atomic per channel and tile
//    into striped copies, as cy_conv_gemm), then the wave's tile goes through LDS 16 rows at a time so that every lane
//    stores 16 bytes (8 bf16 channels of one pixel, or 4 floats when the consumer wants fp32).
// Measured per tile at conv_2 (CY_BF16_PROF=1, cycles of wave 0): 18 K steps x 3.8k (MFMA time of the SIMD's two waves: 2.0k;
// ~1.0k of barrier skew between them, ~0.1k waiting for the DMA) + 9.4k of epilogue; the register-staged one-tile-per-block
// first version had 4.0k per step and ~27k of fixed cost per tile (launch, first-step latency, 16 bias round trips).
#include "common.h"
// Output tiles are stored nontemporal: the 0.7 - 5.7 GB of a layer's output otherwise keep evicting the input rows that the 9 (16)
// taps of the implicit GEMM re-read through L2 (conv_2 forward at 416 x 416: 3.78 -> 3.51 ms; the other shapes unchanged).
#ifndef CY_BF_NT
#define CY_BF_NT 1
#endif
#if CY_BF_NT
#define CY_BF_ST(v, p) __builtin_nontemporal_store((v), (p))
#else
#define CY_BF_ST(v, p) (*(p) = (v))
#endif

#include <type_traits>
#include <stdio.h>
#include <stdlib.h>

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned short u16;
typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));

__device__ __forceinline__ u16 f2bf(float x) {
  const __bf16 b = (__bf16)x;                       // v_cvt_pk_bf16_f32: round to nearest even, NaN stays NaN
  return *(const u16*)&b;
}
__device__ __forceinline__ float bf2f(u16 v) { return __uint_as_float((unsigned)v << 16); }

constexpr int BK = 64;
constexpr int ROWB = 128;                           // bytes of an LDS row: 64 bf16, NOT padded (LDS-DMA writes 1 KiB = 8 whole rows)

// 16 zero bytes that the LDS-DMA lanes of padding pixels / rows past M read instead of the image
__device__ __attribute__((aligned(16))) unsigned conv_bf16_zero16[4];

struct Geo {
  const u16* X; const u16* Wp; void* Y; const float* bias; double* stats;
  int B, Hi, Wi, Cin, Ho, Wo, N, TH, TW, in_stride, dy0, dx0, dstep, Hy, Wy, out_stride, out_oy, out_ox, act;
  float act_slope;                                  // act == 2: LeakyReLU slope (eval forward, BatchNorm folded into the weights)
  long long M;                                      // B * Ho * Wo
  int K;                                            // TH * TW * Cin
  int ntm, ntn;                                     // tiles along M and N
  long long* prof;                                  // developer instrumentation (CY_BF16_PROF): per block cycles of loop / epilogue / waits
  // BNF (input gradient, bf16 output): the output is dA of the producer block; its raw convolution output z (bf16, the layout of Y)
  // and BatchNorm constants come in, d = dA * lrelu'(z * scale + shift) is what is stored, and sum d, sum d * xhat go to
  // bn_red[CY_STATS_COPIES][N][2] (the fp32 kernels' cy_conv_gemm_t.bn_* contract)
  const u16* bn_z; const float* bn_scale; const float* bn_shift; const float* bn_mean; const float* bn_invstd; double* bn_red;
  float bn_slope;
  // ncls > 1: ONE launch for the output-parity classes of a strided input gradient (cy_conv_gemm_bf16_classes): tile = (m tile, n tile,
  // class) with the class fastest, so that the blocks working on the classes of one pixel tile read the same rows of X at the same
  // time (separate launches read X once per class from HBM: 4 x 1.5 GB at conv_3 / 608^2).  Classes share every field but these:
  int ncls;
  int c_dy0[4], c_dx0[4], c_oy[4], c_ox[4];
  const u16* c_wp[4];
};

// BM x BN block tile on WM x WN waves; wave tile (BM / WM) x (BN / WN) = MI x NI tiles of 32x32.
// <256, 256, 2, 4>: 8 waves of 128 x 64 -- per K step 64 KB of operands feed 8 x 32 MFMAs.
//  * PERSISTENT blocks walk the tiles; both operands of a K step go global -> LDS by LDS-DMA (global_load_lds_dwordx4: no
//    staging registers, no ds_write_b128 pass), one K step ahead, into the other of two LDS buffers; one barrier per K
//    step.  The first K step of the NEXT tile is requested before the epilogue of the current one, so that neither a
//    block launch nor the first step's memory latency stands between two tiles (measured on the register-staged,
//    one-tile-per-block version: 13 us of fixed cost per tile against 2.05 us per K step).
//  * LDS rows are 128 bytes, unpadded (an LDS-DMA instruction writes 64 lanes x 16 bytes contiguously = 8 rows); the
//    16-byte chunk c of row r lives at chunk c ^ ((r >> 1) & 7) -- applied on the DMA's per-lane SOURCE address and on
//    the fragment reads -- so that the 16 lanes of a ds_read_b128 group fall on 16 different bank groups.
//  * padding pixels and rows past M read 16 zero bytes (conv_bf16_zero16): the DMA is unconditional.
template <int BM, int BN, int WM, int WN, bool OUT_F32, bool BNF = false, bool TAPIN = false>
__global__ __launch_bounds__(64 * WM * WN, (WM * WN == 8) ? 1 : 2) void conv_bf16_kernel(Geo a) {
  static_assert(!(BNF && OUT_F32), "the fused BatchNorm-backward sums are built for the bf16 output");
  constexpr int NT = 64 * WM * WN;
  constexpr int MI = BM / (32 * WM), NI = BN / (32 * WN);                // 32x32 tiles per wave
  constexpr int A_BYTES = BM * ROWB, B_BYTES = BN * ROWB, BUF_BYTES = A_BYTES + B_BYTES;
  constexpr int RSTEP = NT / 8, NA = BM / RSTEP, NB = BN / RSTEP;        // staging: rows per DMA round, rounds per operand
  static_assert(RSTEP % 16 == 0, "the row swizzle must not depend on the staging round");
  static_assert((NT / 64) * 16 * 32 * NI * 4 + BM * 8 <= BUF_BYTES, "epilogue scratch (wave slices + row offsets) must fit one buffer");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  const int t = threadIdx.x, lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int wm = wave / WN, wn = wave % WN;
  const int li = lane & 31, lh = lane >> 5;
  const int srow = t >> 3;                                               // row of a staging round this thread's lane fills
  const int chunk_l = (t & 7) ^ ((srow >> 1) & 7);                       // logical 16-byte chunk behind its physical chunk t & 7
  const int ntiles = a.ntm * a.ntn * a.ncls;
  const int KT = a.K / BK;                          // K steps: taps x (Cin / 64)
  const int cpt = a.Cin / BK;                       // K steps per tap

  // ---- staging state of the tile whose K steps are being requested
  int iy0[NA], ix0[NA];
  long long xo[NA];
  const u16* wrow[NB];
  int tap_a = 0, tap_b = 0, cstep = 0;              // position of the NEXT K step to request
  // (M < 2^31 is checked by the launcher: pixel indices are 32-bit here, one division pair per thread and tile -- the 64-bit
  // divisions of the first version were ~1500 instructions per tile in front of the last K step's MFMAs)
  auto set_tile = [&](int tile, int slot) {
    const int cls = tile % a.ncls, t2 = tile / a.ncls;
    const unsigned m0 = (unsigned)(t2 / a.ntn) * BM;
    const int n0 = (t2 % a.ntn) * BN;
    const int dy0c = a.c_dy0[cls], dx0c = a.c_dx0[cls];
    const unsigned Mu = (unsigned)a.M, Wo = (unsigned)a.Wo, Ho = (unsigned)a.Ho;
    {
      unsigned m = m0 + srow;
      unsigned q = m / Wo, ox = m - q * Wo, b = q / Ho, oy = q - b * Ho;       // first row; the others are RSTEP pixels further
#pragma unroll
      for (int i = 0; i < NA; ++i) {
        if (m < Mu) {
          iy0[i] = (int)oy * a.in_stride + dy0c;
          ix0[i] = (int)ox * a.in_stride + dx0c;
          // element offset of tap (0, 0), channel chunk_l * 8 of this row's pixel: a K step adds a UNIFORM term to it
          xo[i] = (long long)b * a.Hi * a.Wi * a.Cin + ((long long)iy0[i] * a.Wi + ix0[i]) * a.Cin + chunk_l * 8;
        } else {
          iy0[i] = -(1 << 28); ix0[i] = 0; xo[i] = 0;   // never in range
        }
        m += RSTEP; ox += RSTEP;
        while (ox >= Wo) { ox -= Wo; if (++oy == Ho) { oy = 0; ++b; } }
      }
    }
#pragma unroll
    for (int i = 0; i < NB; ++i) wrow[i] = a.c_wp[cls] + (long long)(n0 + srow + RSTEP * i) * a.K + chunk_l * 8;
    tap_a = 0; tap_b = 0; cstep = 0;
  };
  // One K step = NA + NB LDS-DMA pieces per wave.  piece(buf, j) requests piece j of the NEXT K step of the staged tile;
  // the pieces are issued between the MFMA groups of the current step (all up front, a wave spent ~1000 cycles of every
  // step issuing them before its first MFMA), step_done() advances the tap / channel cursor.
  // the zero block's address, pinned in registers: re-materialised per use it is an s_load through the GOT, whose
  // `s_waitcnt lgkmcnt(0)` also drains the LDS read queue (scalar loads share the counter) -- twice per K step
  unsigned long long zaddr = (unsigned long long)(uintptr_t)conv_bf16_zero16;
  asm volatile("" : "+s"(zaddr));
  const u16* zsrc = (const u16*)(uintptr_t)zaddr;
  auto piece = [&](int buf, int j) {
#ifdef CY_BF_DBG
    if ((CY_BF_DBG & 1) && j < NA) return;           // timing knock-outs (results are wrong): 1 no A pieces, 2 no B pieces
    if ((CY_BF_DBG & 2) && j >= NA) return;
#endif
    unsigned char* Ab = smem_raw + buf * BUF_BYTES + wave * 8 * ROWB;      // this wave's 1 KiB piece of round 0
    if (j < NA) {
      const int iy = iy0[j] + tap_a * a.dstep, ix = ix0[j] + tap_b * a.dstep;
      const bool ok = (unsigned)iy < (unsigned)a.Hi && (unsigned)ix < (unsigned)a.Wi;
      const long long ustep = ((long long)(tap_a * a.dstep) * a.Wi + tap_b * a.dstep) * a.Cin + cstep * BK;   // uniform (scalar)
      const u16* src = ok ? a.X + (xo[j] + ustep) : zsrc;                  // (a select, no branch: the address is cheap now)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                       (__attribute__((address_space(3))) void*)(Ab + j * RSTEP * ROWB), 16, 0, 0);
    } else {
      const int kbase = (tap_a * a.TW + tap_b) * a.Cin + cstep * BK;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(wrow[j - NA] + kbase),
                                       (__attribute__((address_space(3))) void*)(Ab + A_BYTES + (j - NA) * RSTEP * ROWB), 16, 0, 0);
    }
  };
  // K order: the TAPS are the inner loop, the 64-channel chunk the outer one.  A K step reads one 128-byte line per pixel row of the tile;
  // with the channel chunks inside a tap (the first version) a line was touched again only cpt steps later, for the next tap -- 256 KB of
  // other lines per block in between, 8 MB per XCD against 4 MB of L2: conv_2's input gradient at 608^2 fetched 56 GB for 6 GB of dz
  // (FETCH_SIZE; 6.4 TB/s: the launch was bound by those re-reads).  Taps inside: the next TH x TW steps re-read the same lines shifted.
  // (only where it matters -- cpt x BM x 128 B x 32 blocks beyond the L2, Geo.tap_inner set by the launcher: conv_2's forward with 2 chunks
  // per tap and 256-row tiles was 2 % faster the old way)
  // (a TEMPLATE parameter: as a run-time flag the branch in this cursor doubled every launch's time)
  auto step_done = [&]() {
    if constexpr (TAPIN) { if (++tap_b == a.TW) { tap_b = 0; if (++tap_a == a.TH) { tap_a = 0; ++cstep; } } }
    else { if (++cstep == cpt) { cstep = 0; if (++tap_b == a.TW) { tap_b = 0; ++tap_a; } } }
  };
  auto stage = [&](int buf) {                       // a whole K step at once (prologue)
#pragma unroll
    for (int j = 0; j < NA + NB; ++j) piece(buf, j);
    step_done();
  };

  // fragment reads: lane (li, lh) reads logical chunk 2 ks + lh of row li (+ multiples of 32): physical chunk ^ ((li >> 1) & 7)
  int xoff[BK / 16];
#pragma unroll
  for (int ks = 0; ks < BK / 16; ++ks) xoff[ks] = ((2 * ks + lh) ^ ((li >> 1) & 7)) * 16;
  const int a_row = (wm * 32 * MI + li) * ROWB, b_row = A_BYTES + (wn * 32 * NI + li) * ROWB;

  // All tiles take the same time, so blocks that start together would all write their outputs together: 33 MB every ~45 us
  // at conv_2, an HBM write burst that every block then sits out (a fixed ~8 us per tile), while nothing is written during
  // the K loops.  The blocks of an XCD therefore start an eighth of a tile apart (one sleep of ~0.2 us per K step and phase).
  if (gridDim.x >= 64) {
    const int phase = (blockIdx.x >> 3) & 7;
#ifndef CY_BF_NOSTAGGER
    for (int i = 0; i < phase * (KT + 6); ++i) __builtin_amdgcn_s_sleep(8);
#endif
  }
  int cur = 0, slot = 0;
  // Consecutive tiles are consecutive pixel segments: the rows a 3 x 3 tap reads above and below a tile are the rows of the next
  // one or two tiles.  Blocks go to the XCDs round-robin, so each XCD takes a CONTIGUOUS range of tiles per round (the Winograd
  // kernels' remap): the neighbours' lines are then in the same L2 (conv_2's input gradient still fetched 21.7 GB for 6 GB of dz:
  // every line once per tap row).
  int tile = blockIdx.x;
  if ((gridDim.x & 7u) == 0) tile = (int)((blockIdx.x & 7u) * (gridDim.x >> 3) + (blockIdx.x >> 3));
  if (tile < ntiles) { set_tile(tile, 0); stage(0); }
  __builtin_amdgcn_s_waitcnt(0x0F70);               // vmcnt(0): the first tile's first K step
  for (; tile < ntiles; tile += gridDim.x) {
    const int cls = tile % a.ncls, m_tile = (tile / a.ncls) / a.ntn, n0 = ((tile / a.ncls) % a.ntn) * BN;
    const bool has_next = tile + (int)gridDim.x < ntiles;
    f32x16 acc[MI][NI];
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
      for (int ni = 0; ni < NI; ++ni)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.f;
    const bool relu = a.act != 0, has_stats = a.stats != nullptr;
    const float act_lo = a.act == 2 ? a.act_slope : 0.f;          // max(v, v * act_lo): ReLU (0) or LeakyReLU (0 <= slope <= 1)
    long long pf0 = a.prof ? (long long)__builtin_amdgcn_s_memtime() : 0, pfw = 0, pfv = 0;
    for (int kt = 0; kt < KT; ++kt) {
      const long long pfa = a.prof ? (long long)__builtin_amdgcn_s_memtime() : 0;
      // vmcnt(0): this wave's pieces of step kt have landed.  Not for kt = 0: those were waited for in front of the previous
      // tile's output stores (below) -- here the wait would sit out the round trip of the stores just issued
      if (kt) __builtin_amdgcn_s_waitcnt(0x0F70);
      const long long pfb = a.prof ? (long long)__builtin_amdgcn_s_memtime() : 0;
      __builtin_amdgcn_s_barrier();                 // ... and everybody's; every wave is done reading the other buffer
      if (a.prof) { pfw += (long long)__builtin_amdgcn_s_memtime() - pfa; pfv += pfb - pfa; }
      const bool more = kt + 1 < KT || has_next;    // (uniform) is there a K step to request?
      if (kt + 1 == KT && has_next) set_tile(tile + (int)gridDim.x, slot ^ 1);
      // Fragment reads are inline asm, one K-slice (16 k) AHEAD of the MFMAs that consume them, with counted waits: left to
      // itself hipcc (at 237 registers) reuses ONE register quad for all A fragments and waits lgkmcnt(0) behind every
      // read -- four exposed LDS round trips per slice, 60 % of the step.
      const unsigned lbase = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)smem_raw + cur * BUF_BYTES;
#ifndef CY_BF_PPG_DIV
#define CY_BF_PPG_DIV 2           // the K step's DMA pieces are issued behind the first CY_BF_PPG_DIV of its four MFMA groups (swept 1 / 2 / 4)
#endif
      constexpr int PPG = (NA + NB + CY_BF_PPG_DIV - 1) / CY_BF_PPG_DIV;                              // DMA pieces per MFMA group: all in the first two groups (a piece issued late in the step is waited for at the top of the next)
      u32x4_t fa[2][MI], fb[2][NI];
      auto issue = [&](int ks, int set) {
        const unsigned aa = lbase + a_row + xoff[ks], ba = lbase + b_row + xoff[ks];
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(fa[set][mi]) : "v"(aa), "n"(mi * 32 * ROWB));
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(fb[set][ni]) : "v"(ba), "n"(ni * 32 * ROWB));
      };
      issue(0, 0);
#pragma unroll
      for (int ks = 0; ks < BK / 16; ++ks) {
        const int set = ks & 1;
        if (ks + 1 < BK / 16) {
          issue(ks + 1, set ^ 1);
          asm volatile("s_waitcnt lgkmcnt(%0)" :: "n"(MI + NI) : "memory");       // slice ks has landed, slice ks + 1 is in flight
        } else {
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) asm volatile("" : "+v"(fa[set][mi]));     // (ties the wait to the registers it releases)
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) asm volatile("" : "+v"(fb[set][ni]));
#pragma unroll
        for (int mi = 0; mi < MI; ++mi)
#pragma unroll
          for (int ni = 0; ni < NI; ++ni)
            acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, fa[set][mi]), __builtin_bit_cast(bf16x8, fb[set][ni]),
                                                                  acc[mi][ni], 0, 0, 0);
        if (more) {
#pragma unroll
          for (int j = ks * PPG; j < (ks + 1) * PPG && j < NA + NB; ++j) piece(cur ^ 1, j);
        }
      }
      if (more) step_done();
      cur ^= 1;
    }
    const long long pf1 = a.prof ? (long long)__builtin_amdgcn_s_memtime() : 0;
    float bvv[NI];                                  // the tile's bias: ONE round trip, under the barrier (not one per slice)
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) {
      const int n = n0 + wn * 32 * NI + ni * 32 + li;
      bvv[ni] = (a.bias != nullptr && n < a.N) ? a.bias[n] : 0.f;
    }
    __builtin_amdgcn_s_waitcnt(0xC07F);             // lgkmcnt(0)
    __builtin_amdgcn_s_barrier();                   // every wave is done with the last step's buffer: it becomes the epilogue's scratch
    const long long pe0 = a.prof ? (long long)__builtin_amdgcn_s_memtime() : 0;
    long long pe1 = 0, pe2 = 0;
                                                    // (the other buffer is receiving the next tile's first K step)
    // ---- epilogue: 16 rows of the wave's tile at a time through its LDS slice (4 KB), 16-byte stores
    constexpr int WC = 32 * NI;                     // columns of the wave's tile
    float* scratch = (float*)(smem_raw + (cur ^ 1) * BUF_BYTES);
    float* ow = scratch + wave * (16 * WC);
    // output offset of every row's pixel (elements; -1: past M), in the scratch buffer behind the waves' slices
    long long* rowoff = (long long*)(scratch + (NT / 64) * (16 * WC));
    {
      const unsigned Mu = (unsigned)a.M, Wo = (unsigned)a.Wo, Ho = (unsigned)a.Ho;
      for (int r = t; r < BM; r += NT) {
        const unsigned m = (unsigned)m_tile * BM + r;
        long long off = -1;
        if (m < Mu) {
          const unsigned q = m / Wo, ox = m - q * Wo, b = q / Ho, oy = q - b * Ho;
          off = (((long long)b * a.Hy + (long long)oy * a.out_stride + a.c_oy[cls]) * a.Wy + (long long)ox * a.out_stride + a.c_ox[cls]) * a.N;
        }
        rowoff[r] = off;
      }
    }
    __syncthreads();
    float ssum[NI], ssq[NI];
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) { ssum[ni] = 0.f; ssq[ni] = 0.f; }
    constexpr int CPL = OUT_F32 ? 4 : 8;            // channels per lane and store (16 bytes)
    constexpr int LPR = WC / CPL, RPI = 64 / LPR;   // lanes per row, rows per store instruction
    const int cq = lane % LPR, rsub = lane / LPR;
    const int nst = n0 + wn * WC + cq * CPL;
    const bool full_rows = (long long)(m_tile + 1) * BM <= a.M;          // uniform: no row of the tile lies past M
    // BNF: the lane's 8 channels' constants, its running sums, and the z values of a slice requested ONE SLICE AHEAD of their use
    constexpr int NIT = 16 / RPI;                   // store instructions per slice
    float bsc[8], bsh[8], bis[8], bnm[8], b1[8], b2[8];
    constexpr int ZAH = 1;                          // slices of z in flight ahead of the one being stored (2: no faster -- the launch is HBM-bound)
    u32x4_t zq[ZAH + 1][NIT];
    auto zreq = [&](int slice, u32x4_t (&dst)[NIT]) {              // slice = mi * 2 + hf
      const int rb = (wm * MI + (slice >> 1)) * 32 + 16 * (slice & 1);
#pragma unroll
      for (int it = 0; it < NIT; ++it) {
        const long long off = rowoff[rb + it * RPI + rsub];
        // (rows past M and padded channels read the tensor's first bytes and never use them: no branch around the load)
        const u16* zp = (off >= 0 && nst < a.N) ? a.bn_z + off + nst : a.bn_z;
        dst[it] = __builtin_nontemporal_load((const u32x4_t*)zp);
      }
    };
    if constexpr (BNF) {
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int n = nst + e < a.N ? nst + e : 0;
        bsc[e] = a.bn_scale[n]; bsh[e] = a.bn_shift[n]; bis[e] = a.bn_invstd[n]; bnm[e] = -a.bn_mean[n] * bis[e];
        b1[e] = 0.f; b2[e] = 0.f;
      }
#pragma unroll
      for (int q = 0; q < ZAH && q < MI * 2; ++q) zreq(q, zq[q]);
    }
    // (the bias is read ONCE per tile, above: loaded inside these loops it put a global-memory round trip in front of each of
    // the 16 slices -- 16.7k of the tile's 90k cycles at conv_2)
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) {
#pragma unroll
      for (int hf = 0; hf < 2; ++hf) {              // accumulator registers 8 hf .. 8 hf + 7 = rows 16 hf .. 16 hf + 15 of the 32
        const int rbase = (wm * MI + mi) * 32 + 16 * hf;   // first row of this slice in the block tile
        if (has_stats) {                            // (uniform: one branch per slice, selects inside)
          float okf[8];                             // 1 for the lane's rows of this slice that exist (all of them but in the last tile)
#pragma unroll
          for (int r8 = 0; r8 < 8; ++r8)
            okf[r8] = (full_rows || rowoff[rbase + (r8 & 3) + 8 * (r8 >> 2) + 4 * lh] >= 0) ? 1.f : 0.f;
#pragma unroll
          for (int ni = 0; ni < NI; ++ni) {
            const float nok = (n0 + wn * WC + ni * 32 + li < a.N) ? 1.f : 0.f;
#pragma unroll
            for (int r8 = 0; r8 < 8; ++r8) {
              const int row_l = (r8 & 3) + 8 * (r8 >> 2) + 4 * lh;         // 0 .. 15
              float v = acc[mi][ni][8 * hf + r8] + bvv[ni];
              const float vs = v * (okf[r8] * nok);
              ssum[ni] += vs; ssq[ni] = __builtin_fmaf(vs, vs, ssq[ni]);
              if (relu) v = fmaxf(v, v * act_lo);
              ow[row_l * WC + ni * 32 + li] = v;
            }
          }
        } else {
#pragma unroll
          for (int ni = 0; ni < NI; ++ni) {
#pragma unroll
            for (int r8 = 0; r8 < 8; ++r8) {
              const int row_l = (r8 & 3) + 8 * (r8 >> 2) + 4 * lh;         // 0 .. 15
              float v = acc[mi][ni][8 * hf + r8] + bvv[ni];
              if (relu) v = fmaxf(v, v * act_lo);
              ow[row_l * WC + ni * 32 + li] = v;
            }
          }
        }
        if (mi == 0 && hf == 0) {                   // the next tile's first K step (requested a whole step ago) has landed:
          if (a.prof) pe1 = (long long)__builtin_amdgcn_s_memtime();
          __builtin_amdgcn_s_waitcnt(0x0F70);       // waited for HERE, before the first store joins the queue
          if (a.prof) pe2 = (long long)__builtin_amdgcn_s_memtime();
          __builtin_amdgcn_sched_barrier(0);
        }
        if constexpr (BNF) {                        // the next slice's z: a slice of work (and the SIMD's other wave) ahead of its use
          if (mi * 2 + hf + ZAH < MI * 2) zreq(mi * 2 + hf + ZAH, zq[(mi * 2 + hf + ZAH) % (ZAH + 1)]);
        }
#pragma unroll
        for (int it = 0; it < 16 / RPI; ++it) {
          const int row_l = it * RPI + rsub;
          const long long off = rowoff[rbase + row_l];
          if (off >= 0 && nst < a.N) {
            const float* src = ow + row_l * WC + cq * CPL;
            if constexpr (BNF) {
              const f32x4 v0 = *(const f32x4*)src, v1 = *(const f32x4*)(src + 4);
              const u32x4_t zz = zq[(mi * 2 + hf) % (ZAH + 1)][it];
              float d[8];
#pragma unroll
              for (int e = 0; e < 8; ++e) {
                const float z = (e & 1) ? __uint_as_float(zz[e >> 1] & 0xffff0000u) : __uint_as_float(zz[e >> 1] << 16);
                const float v = e < 4 ? v0[e] : v1[e - 4];
                const float y = __builtin_fmaf(z, bsc[e], bsh[e]);
                d[e] = y > 0.f ? v : v * a.bn_slope;
              }
              u32x4_t o;
#pragma unroll
              for (int e = 0; e < 4; ++e) o[e] = (unsigned)f2bf(d[2 * e]) | ((unsigned)f2bf(d[2 * e + 1]) << 16);
              // the sums are those of the STORED (bf16-rounded) gradient: what cy_bn_bwd_reduce_bf16 would read back
#pragma unroll
              for (int e = 0; e < 8; ++e) {
                const float z = (e & 1) ? __uint_as_float(zz[e >> 1] & 0xffff0000u) : __uint_as_float(zz[e >> 1] << 16);
                const float dr = (e & 1) ? __uint_as_float(o[e >> 1] & 0xffff0000u) : __uint_as_float(o[e >> 1] << 16);
                b1[e] += dr;
                b2[e] = __builtin_fmaf(dr, __builtin_fmaf(z, bis[e], bnm[e]), b2[e]);
              }
              CY_BF_ST(o, (u32x4_t*)((u16*)a.Y + off + nst));
            } else if (OUT_F32) {
              CY_BF_ST(*(const f32x4*)src, (f32x4*)((float*)a.Y + off + nst));
            } else {
              const f32x4 v0 = *(const f32x4*)src, v1 = *(const f32x4*)(src + 4);
              u32x4_t o;
              o[0] = (unsigned)f2bf(v0[0]) | ((unsigned)f2bf(v0[1]) << 16);
              o[1] = (unsigned)f2bf(v0[2]) | ((unsigned)f2bf(v0[3]) << 16);
              o[2] = (unsigned)f2bf(v1[0]) | ((unsigned)f2bf(v1[1]) << 16);
              o[3] = (unsigned)f2bf(v1[2]) | ((unsigned)f2bf(v1[3]) << 16);
              CY_BF_ST(o, (u32x4_t*)((u16*)a.Y + off + nst));
            }
          }
        }
      }
    }
    if (a.stats != nullptr) {
      __syncthreads();                              // every wave is done with its `ow` slice
      float* red = scratch;                         // [WM][BN][2]
#pragma unroll
      for (int ni = 0; ni < NI; ++ni) {
        const float s2 = ssum[ni] + __shfl_xor(ssum[ni], 32, 64);
        const float q2 = ssq[ni] + __shfl_xor(ssq[ni], 32, 64);
        if (lh == 0) {
          const int col = wn * WC + ni * 32 + li;
          red[(wm * BN + col) * 2 + 0] = s2;
          red[(wm * BN + col) * 2 + 1] = q2;
        }
      }
      __syncthreads();
      if (t < BN && n0 + t < a.N) {
        double s2 = 0.0, q2 = 0.0;
#pragma unroll
        for (int w = 0; w < WM; ++w) { s2 += (double)red[(w * BN + t) * 2]; q2 += (double)red[(w * BN + t) * 2 + 1]; }
        double* st = a.stats + (size_t)(m_tile % CY_STATS_COPIES) * a.N * 2;
        atomicAdd(st + 2 * (n0 + t), s2);
        atomicAdd(st + 2 * (n0 + t) + 1, q2);
      }
    }
    if constexpr (BNF) {
      // lanes that share cq (the same 8 channels) differ in the lane bits above log2(LPR): fold those, then the waves along M through LDS
#pragma unroll
      for (int e = 0; e < 8; ++e)
#pragma unroll
        for (int msk = LPR; msk < 64; msk <<= 1) { b1[e] += __shfl_xor(b1[e], msk, 64); b2[e] += __shfl_xor(b2[e], msk, 64); }
      __syncthreads();                              // every wave is done with its `ow` slice
      float* red = scratch;                         // [WM][BN][2]
      if (lane < LPR) {
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const int col = wn * WC + cq * CPL + e;
          red[(wm * BN + col) * 2 + 0] = b1[e];
          red[(wm * BN + col) * 2 + 1] = b2[e];
        }
      }
      __syncthreads();
      if (t < BN && n0 + t < a.N) {
        double s1 = 0.0, s2 = 0.0;
#pragma unroll
        for (int w = 0; w < WM; ++w) { s1 += (double)red[(w * BN + t) * 2]; s2 += (double)red[(w * BN + t) * 2 + 1]; }
        double* rd = a.bn_red + (size_t)(m_tile % CY_STATS_COPIES) * a.N * 2;
        atomicAdd(rd + 2 * (n0 + t), s1);
        atomicAdd(rd + 2 * (n0 + t) + 1, s2);
      }
      __syncthreads();                              // (the scratch is the next tile's operand buffer after the next K step)
    }
    slot ^= 1;
    if (a.prof && t == 0) {
      const long long pf2 = (long long)__builtin_amdgcn_s_memtime();
      long long* pp = a.prof + blockIdx.x * 8;
      pp[0] += pf1 - pf0; pp[1] += pf2 - pf1; pp[2] += pfw; pp[3] += 1; pp[4] += pe0 - pf1; pp[5] += pe1 - pe0; pp[6] += pfv; pp[7] += pf2 - pe2;
    }
  }
}

// Wp[n][k] (bf16), k = tap * rows_per_tap + rr: the GEMM B operand, n-major so that a lane's 8 consecutive k are 16 bytes
__global__ void pack_bf16_kernel(const float* __restrict__ W, u16* __restrict__ Wp, int Cout, int Cin, int KH, int KW, int TH,
                                 int TW, int kh0, int kw0, int kstep, int transpose, int K, int Np, long long total) {
  const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= total) return;
  const int k = (int)(idx % K), n = (int)(idx / K);
  const int rows_per_tap = transpose ? Cout : Cin;
  const int ncols = transpose ? Cin : Cout;
  float v = 0.f;
  if (n < ncols) {
    const int tap = k / rows_per_tap, rr = k - tap * rows_per_tap;
    const int ta = tap / TW, tb = tap - ta * TW;
    const int kh = kh0 + ta * kstep, kw = kw0 + tb * kstep;
    const int co = transpose ? rr : n, ci = transpose ? n : rr;
    v = W[(((long long)co * Cin + ci) * KH + kh) * KW + kw];
  }
  Wp[idx] = f2bf(v);
  (void)Np;
}

__global__ void cast_kernel_f2b(const float* __restrict__ in, u16* __restrict__ out, long long n8) {
  const long long stride = (long long)gridDim.x * blockDim.x;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += stride) {
    const f32x4 a = ((const f32x4*)in)[2 * i], b = ((const f32x4*)in)[2 * i + 1];
    u32x4_t o;
    o[0] = (unsigned)f2bf(a[0]) | ((unsigned)f2bf(a[1]) << 16);
    o[1] = (unsigned)f2bf(a[2]) | ((unsigned)f2bf(a[3]) << 16);
    o[2] = (unsigned)f2bf(b[0]) | ((unsigned)f2bf(b[1]) << 16);
    o[3] = (unsigned)f2bf(b[2]) | ((unsigned)f2bf(b[3]) << 16);
    ((u32x4_t*)out)[i] = o;
  }
}
__global__ void cast_kernel_b2f(const u16* __restrict__ in, float* __restrict__ out, long long n8) {
  const long long stride = (long long)gridDim.x * blockDim.x;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += stride) {
    const u32x4_t v = ((const u32x4_t*)in)[i];
    f32x4 a, b;
    a[0] = __uint_as_float(v[0] << 16); a[1] = __uint_as_float(v[0] & 0xffff0000u);
    a[2] = __uint_as_float(v[1] << 16); a[3] = __uint_as_float(v[1] & 0xffff0000u);
    b[0] = __uint_as_float(v[2] << 16); b[1] = __uint_as_float(v[2] & 0xffff0000u);
    b[2] = __uint_as_float(v[3] << 16); b[3] = __uint_as_float(v[3] & 0xffff0000u);
    ((f32x4*)out)[2 * i] = a;
    ((f32x4*)out)[2 * i + 1] = b;
  }
}

template <int BM, int BN, int WM, int WN, bool OUT_F32, bool BNF = false, bool TAPIN = false>
int launch(Geo g, hipStream_t s) {
  if constexpr (!TAPIN) {
    // taps inside a channel chunk where a tile's lines would not survive in L2 between two taps (see step_done)
    if ((g.Cin / BK) * BM > 768 && g.TH * g.TW > 1) return launch<BM, BN, WM, WN, OUT_F32, BNF, true>(g, s);
  }
  const size_t lds = (size_t)2 * (BM + BN) * ROWB;
  int rc = cy_allow_lds(conv_bf16_kernel<BM, BN, WM, WN, OUT_F32, BNF, TAPIN>, lds);
  if (rc) return rc;
  g.ntm = (int)cy_ceil_div(g.M, BM);
  g.ntn = (int)cy_ceil_div(g.N, BN);

  const long long ntiles = (long long)g.ntm * g.ntn * g.ncls;
  const int resident = 256 * ((WM * WN == 8) ? 1 : 2);               // persistent blocks: what fits the chip at once
  const unsigned nblk = (unsigned)(ntiles < resident ? ntiles : resident);
  g.prof = nullptr;
  static const bool prof_on = getenv("CY_BF16_PROF") != nullptr;      // developer instrumentation: synchronous, prints per launch
  if (prof_on) {
    (void)hipMalloc(&g.prof, nblk * 64);
    (void)hipMemsetAsync(g.prof, 0, nblk * 64, s);
  }
  conv_bf16_kernel<BM, BN, WM, WN, OUT_F32, BNF, TAPIN><<<nblk, 64 * WM * WN, lds, s>>>(g);
  if (prof_on) {
    (void)hipStreamSynchronize(s);
    long long* h = (long long*)malloc(nblk * 64);
    (void)hipMemcpy(h, g.prof, nblk * 64, hipMemcpyDeviceToHost);
    double v[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (unsigned i = 0; i < nblk; ++i) for (int k = 0; k < 8; ++k) v[k] += h[8 * i + k];
    const double nt = v[3];
    fprintf(stderr, "conv_bf16<%d,%d> K steps %d: per tile loop %.0f ticks (of which top-of-step wait+barrier %.0f), epilogue %.0f = barrier %.0f + first slice %.0f + ... + rest %.0f; of the top-of-step time, vmcnt wait %.0f; tiles %.0f\n",
            BM, BN, g.K / BK, v[0] / nt, v[2] / nt, v[1] / nt, v[4] / nt, v[5] / nt, v[7] / nt, v[6] / nt, nt);
    free(h); (void)hipFree(g.prof);
  }
  return 0;
}
template <bool OUT_F32, bool BNF = false>
int launch_n(const Geo& g, hipStream_t s) {
  if (g.N % 256 == 0) return launch<256, 256, 2, 4, OUT_F32, BNF>(g, s);
  if (g.N % 128 == 0) return launch<512, 128, 4, 2, OUT_F32, BNF>(g, s);   // the same 128 x 64 wave tiles (256 x 128 had 16 MFMAs per wave and K step: 0.30)
  if (g.M >= 512 * 256) return launch<512, 64, 8, 1, OUT_F32, BNF>(g, s);     // N = 64 (conv_3): 8 waves of 64 x 64 instead of 4 of 32 x 64
  return launch<128, 64, 4, 1, OUT_F32, BNF>(g, s);
}

}  // namespace

extern "C" long long cy_conv_bf16_packed_elems(int K, int N) { return (long long)K * (cy_ceil_div(N, 64) * 64); }

extern "C" int cy_conv_bf16_pack_weights(const float* W, void* Wp, int Cout, int Cin, int KH, int KW, int TH, int TW, int kh0,
                                         int kw0, int kstep, int transpose, void* stream) {
  CY_REQUIRE(W && Wp && Cout > 0 && Cin > 0 && TH > 0 && TW > 0, "cy_conv_bf16_pack_weights: bad arguments");
  CY_REQUIRE(kh0 + (TH - 1) * kstep < KH && kw0 + (TW - 1) * kstep < KW, "cy_conv_bf16_pack_weights: taps outside the kernel");
  const int K = TH * TW * (transpose ? Cout : Cin);
  const int Np = (int)(cy_ceil_div(transpose ? Cin : Cout, 64) * 64);
  const long long total = (long long)K * Np;
  pack_bf16_kernel<<<(unsigned)cy_ceil_div(total, 256), 256, 0, (hipStream_t)stream>>>(W, (u16*)Wp, Cout, Cin, KH, KW, TH, TW, kh0,
                                                                                     kw0, kstep, transpose, K, Np, total);
  CY_LAUNCH_CHECK("cy_conv_bf16_pack_weights");
  return 0;
}

// ncls descriptors that differ only in Wp, dy0, dx0, out_oy, out_ox: the output-parity classes of a strided input gradient in ONE launch
extern "C" int cy_conv_gemm_bf16_classes(const cy_conv_gemm_t* a, int ncls, int out_f32, void* stream) {
  CY_REQUIRE(a && ncls >= 1 && ncls <= 4, "cy_conv_gemm_bf16_classes: 1 to 4 class descriptors");
  CY_REQUIRE(a->X && a->Wp && a->Y, "cy_conv_gemm_bf16: null pointer");
  CY_REQUIRE(a->Cin % 64 == 0 && a->N % 64 == 0, "cy_conv_gemm_bf16: Cin=%d and N=%d must be multiples of 64", a->Cin, a->N);
  CY_REQUIRE(a->xs_c == 1 && a->xs_x == a->Cin && a->xs_y == (long long)a->Wi * a->Cin &&
             a->xs_b == (long long)a->Hi * a->Wi * a->Cin, "cy_conv_gemm_bf16: X must be plain NHWC");
  CY_REQUIRE(a->bn_red == nullptr || (!out_f32 && a->bn_z && a->bn_scale && a->bn_shift && a->bn_mean && a->bn_invstd && (((uintptr_t)a->bn_z) & 15) == 0),
             "cy_conv_gemm_bf16: the fused BatchNorm-backward sums need the bf16 output, bn_z (bf16, 16-byte aligned) and scale / shift / mean / invstd");
  CY_REQUIRE((((uintptr_t)a->X | (uintptr_t)a->Wp | (uintptr_t)a->Y) & 15) == 0, "cy_conv_gemm_bf16: pointers must be 16-byte aligned");
  Geo g;
  g.X = (const u16*)a->X; g.Wp = (const u16*)a->Wp; g.Y = (void*)a->Y; g.bias = a->bias; g.stats = a->stats;
  g.B = a->B; g.Hi = a->Hi; g.Wi = a->Wi; g.Cin = a->Cin; g.Ho = a->Ho; g.Wo = a->Wo; g.N = a->N; g.TH = a->TH; g.TW = a->TW;
  g.in_stride = a->in_stride; g.dy0 = a->dy0; g.dx0 = a->dx0; g.dstep = a->dstep; g.Hy = a->Hy; g.Wy = a->Wy;
  g.out_stride = a->out_stride; g.out_oy = a->out_oy; g.out_ox = a->out_ox; g.act = a->act; g.act_slope = a->act_slope;
  CY_REQUIRE(a->act >= 0 && a->act <= 2 && (a->act != 2 || (a->act_slope >= 0.f && a->act_slope <= 1.f)),
             "cy_conv_gemm_bf16: act=%d / act_slope=%g (LeakyReLU slope must be in [0, 1])", a->act, (double)a->act_slope);
  g.ncls = ncls;
  for (int c = 0; c < 4; ++c) {
    const cy_conv_gemm_t* q = a + (c < ncls ? c : 0);
    if (c < ncls && c > 0) {
      CY_REQUIRE(q->X == a->X && q->Y == a->Y && q->bias == a->bias && q->stats == a->stats && q->B == a->B && q->Hi == a->Hi &&
                 q->Wi == a->Wi && q->Cin == a->Cin && q->Ho == a->Ho && q->Wo == a->Wo && q->N == a->N && q->TH == a->TH &&
                 q->TW == a->TW && q->in_stride == a->in_stride && q->dstep == a->dstep && q->Hy == a->Hy && q->Wy == a->Wy &&
                 q->out_stride == a->out_stride && q->act == a->act && q->act_slope == a->act_slope && q->bn_red == a->bn_red &&
                 q->bn_z == a->bn_z && q->bn_scale == a->bn_scale && q->bn_shift == a->bn_shift && q->bn_mean == a->bn_mean &&
                 q->bn_invstd == a->bn_invstd && q->bn_slope == a->bn_slope && q->Wp && (((uintptr_t)q->Wp) & 15) == 0,
                 "cy_conv_gemm_bf16_classes: class %d differs from class 0 in more than Wp / dy0 / dx0 / out_oy / out_ox", c);
    }
    g.c_dy0[c] = q->dy0; g.c_dx0[c] = q->dx0; g.c_oy[c] = q->out_oy; g.c_ox[c] = q->out_ox; g.c_wp[c] = (const u16*)q->Wp;
  }
  g.M = (long long)a->B * a->Ho * a->Wo;
  CY_REQUIRE(g.M < (1ll << 31) - 512, "cy_conv_gemm_bf16: more than 2^31 output pixels");
  CY_REQUIRE(g.M * ncls * ((a->N + 63) / 64) < (1ll << 31), "cy_conv_gemm_bf16: too many tiles");
  g.K = a->TH * a->TW * a->Cin;
  hipStream_t s = (hipStream_t)stream;
  int rc;
  g.bn_z = (const u16*)a->bn_z; g.bn_scale = a->bn_scale; g.bn_shift = a->bn_shift; g.bn_mean = a->bn_mean; g.bn_invstd = a->bn_invstd;
  g.bn_red = a->bn_red; g.bn_slope = a->bn_slope;
  rc = out_f32 ? launch_n<true>(g, s) : (a->bn_red != nullptr ? launch_n<false, true>(g, s) : launch_n<false>(g, s));
  if (rc) return rc;
  CY_LAUNCH_CHECK("cy_conv_gemm_bf16");
  return 0;
}

extern "C" int cy_conv_gemm_bf16(const cy_conv_gemm_t* a, int out_f32, void* stream) {
  return cy_conv_gemm_bf16_classes(a, 1, out_f32, stream);
}

extern "C" int cy_cast_f32_bf16(const float* in, void* out, long long n, void* stream) {
  CY_REQUIRE(in && out && n > 0 && n % 8 == 0, "cy_cast_f32_bf16: n must be a positive multiple of 8");
  long long blocks = cy_ceil_div(n / 8, 256);
  if (blocks > 65536) blocks = 65536;
  cast_kernel_f2b<<<(unsigned)blocks, 256, 0, (hipStream_t)stream>>>(in, (u16*)out, n / 8);
  CY_LAUNCH_CHECK("cy_cast_f32_bf16");
  return 0;
}
extern "C" int cy_cast_bf16_f32(const void* in, float* out, long long n, void* stream) {
  CY_REQUIRE(in && out && n > 0 && n % 8 == 0, "cy_cast_bf16_f32: n must be a positive multiple of 8");
  long long blocks = cy_ceil_div(n / 8, 256);
  if (blocks > 65536) blocks = 65536;
  cast_kernel_b2f<<<(unsigned)blocks, 256, 0, (hipStream_t)stream>>>((const u16*)in, out, n / 8);
  CY_LAUNCH_CHECK("cy_cast_bf16_f32");
  return 0;
}

// ================================================================================================ BatchNorm / activation, bf16 tensors
// The bf16 counterparts of cy_affine_act / cy_bn_bwd_reduce / cy_bn_bwd_apply (bn.hip): raw conv outputs z and the
// activations are bf16 NHWC, the per-channel quantities (scale, shift, mean, invstd, the double sums) stay fp32 / double.
// A thread owns 8 consecutive channels (16 bytes of bf16); N divides 2048, so the channels of a thread never change
// along its grid-stride walk and their parameters live in registers.
namespace {

__device__ __forceinline__ void unpack8(const u32x4_t v, float (&f)[8]) {
#pragma unroll
  for (int k = 0; k < 4; ++k) { f[2 * k] = __uint_as_float(v[k] << 16); f[2 * k + 1] = __uint_as_float(v[k] & 0xffff0000u); }
}
__device__ __forceinline__ u32x4_t pack8(const float (&f)[8]) {
  u32x4_t o;
#pragma unroll
  for (int k = 0; k < 4; ++k) o[k] = (unsigned)f2bf(f[2 * k]) | ((unsigned)f2bf(f[2 * k + 1]) << 16);
  return o;
}
// 8 values of a gradient tensor that is bf16 or fp32
template <bool F32> __device__ __forceinline__ void load8(const void* p, long long i8, float (&f)[8]) {
  if (F32) {
    const f32x4 a = ((const f32x4*)p)[2 * i8], b = ((const f32x4*)p)[2 * i8 + 1];
    f[0] = a[0]; f[1] = a[1]; f[2] = a[2]; f[3] = a[3]; f[4] = b[0]; f[5] = b[1]; f[6] = b[2]; f[7] = b[3];
  } else {
    unpack8(((const u32x4_t*)p)[i8], f);
  }
}
template <bool F32> __device__ __forceinline__ void store8(void* p, long long i8, const float (&f)[8]) {
  if (F32) {
    ((f32x4*)p)[2 * i8] = f32x4{f[0], f[1], f[2], f[3]};
    ((f32x4*)p)[2 * i8 + 1] = f32x4{f[4], f[5], f[6], f[7]};
  } else {
    ((u32x4_t*)p)[i8] = pack8(f);
  }
}

template <bool OUT_F32>
__global__ __launch_bounds__(256) void affine_act_bf16_kernel(const u16* __restrict__ Z, void* __restrict__ A,
                                                              const float* __restrict__ scale, const float* __restrict__ shift,
                                                              float slope, long long n8, int N) {
  const long long stride = (long long)gridDim.x * 256;
  long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  const int c = (int)((i * 8) % N);
  float sc[8], sh[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) { sc[k] = scale[c + k]; sh[k] = shift[c + k]; }
  // four items (64 bytes) in flight per thread: with one, a CU had 32 KB outstanding against the ~72 KB that cover an HBM round trip
  for (; i + 3 * stride < n8; i += 4 * stride) {
    u32x4_t v[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) v[u] = ((const u32x4_t*)Z)[i + u * stride];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      float z[8];
      unpack8(v[u], z);
#pragma unroll
      for (int k = 0; k < 8; ++k) { const float y = z[k] * sc[k] + sh[k]; z[k] = y > 0.f ? y : y * slope; }
      store8<OUT_F32>(A, i + u * stride, z);
    }
  }
  for (; i < n8; i += stride) {
    float z[8];
    unpack8(((const u32x4_t*)Z)[i], z);
#pragma unroll
    for (int k = 0; k < 8; ++k) { const float y = z[k] * sc[k] + sh[k]; z[k] = y > 0.f ? y : y * slope; }
    store8<OUT_F32>(A, i, z);
  }
}

template <bool DA_F32>
__global__ __launch_bounds__(256) void bn_bwd_reduce_bf16_kernel(const u16* __restrict__ Z, const void* __restrict__ dA,
                                                                 const float* __restrict__ scale, const float* __restrict__ shift,
                                                                 const float* __restrict__ mean, const float* __restrict__ invstd,
                                                                 float slope, double* __restrict__ red, long long P, int N,
                                                                 long long rows_per_block) {
  __shared__ float sm[256 * 16];
  const int t = threadIdx.x;
  const int G = N / 8;                              // channel groups; G divides 256
  const int cg = t % G, rsub = t / G, rpar = 256 / G;
  const int c = cg * 8;
  float sc[8], sh[8], mu[8], is[8], s1[8], s2[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) { sc[k] = scale[c + k]; sh[k] = shift[c + k]; mu[k] = mean[c + k]; is[k] = invstd[c + k]; s1[k] = 0.f; s2[k] = 0.f; }
  const long long r0 = (long long)blockIdx.x * rows_per_block;
  long long r1 = r0 + rows_per_block;
  if (r1 > P) r1 = P;
  for (long long r = r0 + rsub; r < r1; r += rpar) {
    float z[8], g[8];
    unpack8(((const u32x4_t*)Z)[r * G + cg], z);
    load8<DA_F32>(dA, r * G + cg, g);
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const float y = z[k] * sc[k] + sh[k];
      const float d = y > 0.f ? g[k] : g[k] * slope;
      s1[k] += d;
      s2[k] += d * ((z[k] - mu[k]) * is[k]);
    }
  }
#pragma unroll
  for (int k = 0; k < 8; ++k) { sm[t * 16 + k] = s1[k]; sm[t * 16 + 8 + k] = s2[k]; }
  __syncthreads();
  if (t < G) {
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      float a = 0.f, b = 0.f;
      for (int j = 0; j < rpar; ++j) { a += sm[(j * G + t) * 16 + k]; b += sm[(j * G + t) * 16 + 8 + k]; }
      atomicAdd(red + 2 * (c + k), (double)a);
      atomicAdd(red + 2 * (c + k) + 1, (double)b);
    }
  }
}

__device__ __forceinline__ void bn_apply_consts(float sc, float mu, float is, double r0, double r1, double inv_count, float& ka, float& kb,
                                                float& kc) {
  const float m1 = (float)(r0 * inv_count), m2 = (float)(r1 * inv_count);
  ka = sc;
  kb = -sc * is * m2;
  kc = sc * (mu * is * m2 - m1);
}

template <bool DA_F32>
__global__ __launch_bounds__(256) void bn_bwd_apply_bf16_kernel(const u16* __restrict__ Z, const void* __restrict__ dA,
                                                                u16* __restrict__ dZ, const float* __restrict__ scale,
                                                                const float* __restrict__ shift, const float* __restrict__ mean,
                                                                const float* __restrict__ invstd, float slope,
                                                                const double* __restrict__ red, double inv_count, long long n8,
                                                                int N) {
  const long long stride = (long long)gridDim.x * 256;
  long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  const int c = (int)((i * 8) % N);
  // dz = sc (d - m1 - (z - mu) is m2) = d ka + z kb + kc: the three constants are formed ONCE per channel by bn_apply_consts (the same
  // function in the fused weight gradient, cy_conv_wgrad_bf16_bn: the two paths agree bit for bit)
  float sc[8], sh[8], ka[8], kb[8], kc[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    sc[k] = scale[c + k]; sh[k] = shift[c + k];
    bn_apply_consts(sc[k], mean[c + k], invstd[c + k], red[2 * (c + k)], red[2 * (c + k) + 1], inv_count, ka[k], kb[k], kc[k]);
  }
  for (; i < n8; i += stride) {
    float z[8], g[8], o[8];
    unpack8(((const u32x4_t*)Z)[i], z);
    load8<DA_F32>(dA, i, g);
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const float y = z[k] * sc[k] + sh[k];
      const float d = y > 0.f ? g[k] : g[k] * slope;
      o[k] = __builtin_fmaf(d, ka[k], __builtin_fmaf(z[k], kb[k], kc[k]));
    }
    ((u32x4_t*)dZ)[i] = pack8(o);
  }
}

__global__ void bn_param_grad2_kernel(const double* __restrict__ red, float* dgamma, float* dbeta, int N) {
  const int n = blockIdx.x * blockDim.x + threadIdx.x;
  if (n >= N) return;
  dbeta[n] = (float)red[2 * n];
  dgamma[n] = (float)red[2 * n + 1];
}

inline bool bf16_shape_ok(long long P, int N) { return P > 0 && N >= 8 && N <= 2048 && (2048 % N) == 0; }
inline unsigned bf16_grid(long long n8) {
  long long b = cy_ceil_div(n8, 256 * 4);
  if (b > 65536) b = 65536;
  if (b < 1) b = 1;
  return (unsigned)b;
}

}  // namespace

extern "C" int cy_affine_act_bf16(const void* Z, void* A, const float* scale, const float* shift, float slope, long long P, int N,
                                  int out_f32, void* stream) {
  CY_REQUIRE(Z && A && scale && shift && bf16_shape_ok(P, N), "cy_affine_act_bf16: bad arguments (N=%d must divide 2048)", N);
  const long long n8 = P * N / 8;
  if (out_f32) affine_act_bf16_kernel<true><<<bf16_grid(n8), 256, 0, (hipStream_t)stream>>>((const u16*)Z, A, scale, shift, slope, n8, N);
  else affine_act_bf16_kernel<false><<<bf16_grid(n8), 256, 0, (hipStream_t)stream>>>((const u16*)Z, A, scale, shift, slope, n8, N);
  CY_LAUNCH_CHECK("cy_affine_act_bf16");
  return 0;
}

extern "C" int cy_bn_bwd_reduce_bf16(const void* Z, const void* dA, int da_f32, const float* scale, const float* shift,
                                     const float* mean, const float* invstd, float slope, double* red, long long P, int N,
                                     void* stream) {
  CY_REQUIRE(Z && dA && scale && shift && mean && invstd && red && bf16_shape_ok(P, N), "cy_bn_bwd_reduce_bf16: bad arguments");
  hipStream_t s = (hipStream_t)stream;
  hipError_t e = hipMemsetAsync(red, 0, (size_t)N * 2 * sizeof(double), s);
  if (e != hipSuccess) return cy_set_error((int)e, "cy_bn_bwd_reduce_bf16: memset: %s", hipGetErrorString(e));
  const int rpar = 256 / (N / 8);
  long long rows_per_block = cy_ceil_div(P, 2048);
  if (rows_per_block < 4 * rpar) rows_per_block = 4 * rpar;
  rows_per_block = cy_ceil_div(rows_per_block, rpar) * rpar;
  const unsigned blocks = (unsigned)cy_ceil_div(P, rows_per_block);
  if (da_f32) bn_bwd_reduce_bf16_kernel<true><<<blocks, 256, 0, s>>>((const u16*)Z, dA, scale, shift, mean, invstd, slope, red, P, N, rows_per_block);
  else bn_bwd_reduce_bf16_kernel<false><<<blocks, 256, 0, s>>>((const u16*)Z, dA, scale, shift, mean, invstd, slope, red, P, N, rows_per_block);
  CY_LAUNCH_CHECK("cy_bn_bwd_reduce_bf16");
  return 0;
}

extern "C" int cy_bn_bwd_apply_bf16(const void* Z, const void* dA, int da_f32, void* dZ, const float* scale, const float* shift,
                                    const float* mean, const float* invstd, float slope, const double* red, float* dgamma,
                                    float* dbeta, long long P, int N, void* stream) {
  CY_REQUIRE(Z && dA && dZ && scale && shift && mean && invstd && red && bf16_shape_ok(P, N), "cy_bn_bwd_apply_bf16: bad arguments");
  hipStream_t s = (hipStream_t)stream;
  const long long n8 = P * N / 8;
  if (da_f32) bn_bwd_apply_bf16_kernel<true><<<bf16_grid(n8), 256, 0, s>>>((const u16*)Z, dA, (u16*)dZ, scale, shift, mean, invstd, slope, red, 1.0 / (double)P, n8, N);
  else bn_bwd_apply_bf16_kernel<false><<<bf16_grid(n8), 256, 0, s>>>((const u16*)Z, dA, (u16*)dZ, scale, shift, mean, invstd, slope, red, 1.0 / (double)P, n8, N);
  CY_LAUNCH_CHECK("cy_bn_bwd_apply_bf16");
  if (dgamma && dbeta) {
    bn_param_grad2_kernel<<<(N + 255) / 256, 256, 0, s>>>(red, dgamma, dbeta, N);
    CY_LAUNCH_CHECK("cy_bn_bwd_apply_bf16(param)");
  }
  return 0;
}

// ================================================================================================ weight gradient, bf16
// dW[co][ci][kh][kw] = sum over (b, oy, ox) of dZ[b][oy][ox][co] * X[b][oy*s - 1 + kh][ox*s - 1 + kw][ci]   (pad 1)
// for the two layer types of the backbone (3x3 / stride 1 and 4x4 / stride 2; models.py:350-363).  The pixels are the
// MFMA k dimension, and both operands are stored pixel-major (NHWC), so the fragments -- 8 consecutive PIXELS of one
// channel per lane -- come out of LDS through gfx950's transposing read (ds_read_b64_tr_b16: a 4 pixel x 16 channel
// block per 16 lanes, delivered channel-major); nothing is transposed in memory or in registers.
//  * Block = CO_T output channels x CI_T input channels x ALL taps, 4 waves, each 32 output channels x CI_W input
//    channels (CI_W = 64 for 3x3, 32 for 4x4): 9 x 2 or 16 x 1 accumulator tiles of 32x32 stay in registers.
//  * Work unit (chunk) = 2 output rows x 32 pixels: their dZ rows (64 x CO_T) and the KH + stride input rows they
//    touch ((31 s + KH) x CI_T each, zero outside the image) are staged in LDS once and serve every tap; rows are
//    padded so that the four pixel rows of a transposing read fall on different bank groups.  The staging loads are
//    inline-asm buffer loads through per-image descriptors, two chunks ahead, waited for by hand (see the kernel).
//  * The pixel range is split over blocks; partial sums go to a slab per split and are added in a fixed order.
namespace {

typedef short s16x4 __attribute__((ext_vector_type(4)));

struct WgArgs {
  const u16* X; const u16* dZ; float* slabs;
  int B, Hi, Wi, Cin, Ho, Wo, Cout;
  int nsplit, ntiles_ci;                            // grid: blockIdx.x = (s_hi * ntiles + tile) * 8 + s_lo
  long long nchunk;
  // BNF (cy_conv_wgrad_bf16_bn): dZ holds the PREMASKED gradient d = dA * lrelu'(y) (what the consumer's input-gradient epilogue stored);
  // dz = sc (d - m1 - (z - mu) is m2) -- in cy_bn_bwd_apply_bf16's form d ka + z kb + kc, so that the two paths agree bit for bit -- is formed
  // between the load and the LDS store and written to dZout for the input-gradient kernel: the elementwise pass (18 GB at conv_2,
  // 608 x 608) is gone.  bnc = [3][Cout]: ka, kb, kc of dz = d ka + z kb + kc.
  const u16* Z; u16* dZout; const float* bnc;
};

// developer knob for timing experiments (results are wrong when set): 1 one MFMA per chunk, 2 every load out of range (no memory traffic),
// fused variant: 4 no z DMA, 8 dz stores dropped, 16 no BatchNorm arithmetic
#ifndef CY_WG_DBG
#define CY_WG_DBG 0
#endif
#ifndef CY_WG_ST_AUX
#define CY_WG_ST_AUX ""              // cache policy of the fused variant's dz store (" nt": swept)
#endif
#ifndef CY_WG_RING
#define CY_WG_RING 3               // B fragments in flight ahead of their MFMAs (swept 2 / 3 / 4)
#endif
typedef int wg_i32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ wg_i32x4 wg_desc(const void* p, unsigned bytes) {
  const unsigned long long b = (unsigned long long)(uintptr_t)p;
  return wg_i32x4{(int)(unsigned)b, (int)(unsigned)((b >> 32) & 0xffffu), (int)bytes, 0x00020000};
}
// loads hipcc does not count (see wgrad_bf16_kernel); waited for by hand
__device__ __forceinline__ void wg_load(u32x4_t& dst, wg_i32x4 desc, unsigned voff, unsigned soff) {
  asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen" : "=v"(dst) : "v"(voff), "s"(desc), "s"(soff));
}

// (the s_nop: a vector-memory store of more than 64 bits reads its data registers a cycle behind its issue, and a vector instruction
// that overwrites them right away needs a wait state in between -- hipcc pads it for its own stores and cannot for an asm statement:
// without it the first dword of now and then a stored piece was the NEXT item's LDS address, computed into the same register)
__device__ __forceinline__ void wg_store(const u32x4_t& src, wg_i32x4 desc, unsigned voff, unsigned soff) {
  asm volatile("buffer_store_dwordx4 %0, %1, %2, %3 offen" CY_WG_ST_AUX "\n\ts_nop 1" : : "v"(src), "v"(voff), "s"(desc), "s"(soff) : "memory");
}

__device__ __forceinline__ bf16x8 tr_frag(const u16* p0, const u16* p1) {
  const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)p0);
  const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)p1);
  typedef short s16x8 __attribute__((ext_vector_type(8)));
  const s16x8 v = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
  return *(const bf16x8*)&v;
}

// NW waves per block: 4 (one per SIMD, each 32 output channels x CI_W input channels x all taps), or 8 for the 3x3 layers --
// the same block tile cut into 32 x 32-channel wave tiles (9 accumulator tiles = 144 registers instead of 288), so that two
// waves share a SIMD and one's LDS round trips (every MFMA needs a fresh transposed B fragment) run under the other's MFMAs.
// 4x4 layers (16 taps, one 32-channel column per wave: 256 accumulator registers): the 8 waves are two TAP-ROW groups of
// the 4-wave layout (TG = 2: kernel rows 0-1 / 2-3), 128 accumulator registers each.
template <int KH, int STRIDE, int WCO, int NW = 4, int TG = 1, bool BNF = false>
__global__ __launch_bounds__(64 * NW, 1) void wgrad_bf16_kernel(WgArgs a) {
  constexpr int NTHR = 64 * NW;
  constexpr int WCI = NW / (WCO * TG), CO_T = 32 * WCO, CI_T = (KH == 3) ? 64 : 32 * (4 / WCO), CI_W = CI_T / WCI, NT = CI_W / 32;
  constexpr int KHG = KH / TG;                      // kernel rows per tap group
  static_assert(CI_W % 32 == 0 && NT >= 1 && KH % TG == 0 && WCI >= 1, "wave tile");
  // work unit: PR = 2 output rows x PW = 32 pixels (64 pixels = four 16-pixel k slices per barrier; the XR = KH + STRIDE input rows they
  // touch are staged once: a third fewer input bytes than two one-row units, and half the barriers per MFMA)
  constexpr int PW = 32, PR = 2, PIX = PW * PR, PXW = (PW - 1) * STRIDE + KH, XR = (PR - 1) * STRIDE + KH, TAPS = KH * KH;
  constexpr int DZB = CO_T * 2 + 64;                                         // bytes per dZ pixel row in LDS
  constexpr int XB = (STRIDE == 1) ? CI_T * 2 + 64 : (CI_T == 32 ? 96 : 160); // bytes per X pixel in LDS
  constexpr int DZ_IMG = PIX * DZB, X_IMG = XR * PXW * XB;
  constexpr int DZ_CH = PIX * (CO_T / 8), X_CH = XR * PXW * (CI_T / 8);       // 16-byte pieces per chunk
  constexpr int NDZ = (DZ_CH + NTHR - 1) / NTHR, NX = (X_CH + NTHR - 1) / NTHR;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  unsigned char* dzimg = smem_raw;                  // [2][DZ_IMG]
  unsigned char* ximg = smem_raw + 2 * DZ_IMG;      // [2][X_IMG]
  float* bncs = (float*)(smem_raw + 2 * DZ_IMG + 2 * X_IMG);   // BNF: [3][CO_T]
  // BNF: z of the staged chunk comes in by LDS-DMA, every lane's 16 bytes at its item index (it reads back what it requested: no
  // layout, no barrier): in registers it cost a second register file of staging and left ONE set -- no prefetch beyond the chunk
  unsigned char* zlds = smem_raw + 2 * DZ_IMG + 2 * X_IMG + 3 * CO_T * 4;   // [NSET][NDZ * NTHR * 16]
  const int t = threadIdx.x, lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int tg = wave / (WCO * WCI), wv = wave % (WCO * WCI);
  const int wco = wv / WCI, wci = wv % WCI;
  const int ntiles = (a.Cout / CO_T) * a.ntiles_ci;
  const int s_lo = blockIdx.x & 7, rest = blockIdx.x >> 3;
  const int tile = rest % ntiles, split = (rest / ntiles) * 8 + s_lo;
  if (split >= a.nsplit) return;                    // (whole block: no barrier has been reached yet)
  const int co0 = (tile / a.ntiles_ci) * CO_T, ci0 = (tile % a.ntiles_ci) * CI_T;
  const int CW = (a.Wo + PW - 1) / PW;
  const long long c_lo = a.nchunk * split / a.nsplit, c_hi = a.nchunk * (split + 1) / a.nsplit;

  f32x16 acc[TAPS / TG][NT];
#pragma unroll
  for (int k = 0; k < TAPS / TG; ++k)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[k][nt][r] = 0.f;

  // TWO register sets for the staged chunk: the loads of chunk c + 2 are issued while chunk c is multiplied and chunk c + 1 waits in the
  // other set.  The loads are inline asm through buffer descriptors of the image, waited for by hand: tracked by hipcc, the exec-masked
  // loads drew `s_waitcnt vmcnt(0)` in front of every LDS store and `__syncthreads()` drains the vector-memory queue as well, so a
  // chunk's 24 KB per CU were requested and awaited inside the same chunk: 24 KB in flight against ~2 us of loaded HBM latency is the
  // 1.3 us per chunk the launch took whatever else the loop did (conv_2 at 608 x 608: 36 GB through L2 in 7.5 ms).
  //  * padding and items that do not exist get the offset 2^31, beyond every descriptor's num_records: zeros, no access, no exec mask
  //    (the range check sees the vector offset only, so top / bottom rows are flagged like left / right columns);
  //  * a block whose chunk range has ended keeps loading PHANTOM chunks (num_records = 0): the loop issues the same NL loads in every
  //    iteration, and the wait in front of the LDS stores is the constant vmcnt(NL).
  constexpr int NL = (BNF ? 2 : 1) * NDZ + NX;     // vector-memory operations per staged chunk (BNF: + the z DMAs)
  constexpr int NSET = NDZ + NX <= 5 ? 2 : 1;            // (conv_3's 64 x 64-channel tile stages 8 pieces per thread: two sets spill)
  u32x4_t rdz[NSET][NDZ], rx[NSET][NX];
  unsigned dzv[NDZ], xv[NX];
  int dzpx[NDZ], xjj[NX];                           // (packed: the column in bits 0..7, the row above)
#pragma unroll
  for (int i = 0; i < NDZ; ++i) {
    const int c = t + NTHR * i;
    const int px = c / (CO_T / 8), q = c % (CO_T / 8);
    dzv[i] = c < DZ_CH ? (unsigned)((((px / PW) * a.Wo + (px % PW)) * a.Cout + co0 + q * 8) * 2) : 0x80000000u;
    dzpx[i] = (px % PW) | ((px / PW) << 8);          // column, and bit 8: the chunk's second output row
  }
#pragma unroll
  for (int i = 0; i < NX; ++i) {
    const int c = t + NTHR * i;
    const int q = c % (CI_T / 8), pj = c / (CI_T / 8), j = pj % PXW, r = pj / PXW;
    xv[i] = c < X_CH ? (unsigned)(((r * a.Wi + j) * a.Cin + ci0 + q * 8) * 2) : 0x80000000u;
    xjj[i] = j | (r << 8) | (r == 0 ? 1 << 16 : 0) | (j == 0 ? 2 << 16 : 0);   // patch column, row from bit 8, first-row / first-column flags from bit 16
  }
  const unsigned xshift = (unsigned)((a.Wi + 1) * a.Cin * 2);      // the X descriptor's base stands one row and one pixel in front of the image
  const unsigned ximg_b = (unsigned)(a.Hi * a.Wi * a.Cin * 2), zimg_b = (unsigned)(a.Ho * a.Wo * a.Cout * 2);
  // chunk cursor (segment of 32 pixels, output row, image) of the NEXT chunk to load: it RUNS -- derived from the chunk index per
  // chunk, the two 64-bit division pairs were ~240 of the ~290 scalar instructions a wave issued per chunk next to its 18 MFMAs
  const int RH = (a.Ho + PR - 1) / PR;             // row pairs per image
  int ccw = (int)(c_lo % CW), coy, cb_;
  {
    const long long rr = c_lo / CW;
    coy = (int)(rr % RH) * PR; cb_ = (int)(rr / RH);
  }
  long long left = c_hi - c_lo;                     // real chunks not yet requested
  const bool writer = BNF && (tile % a.ntiles_ci) == 0;          // the input-channel tiles of a dz tile all form it; the first one writes it
  unsigned dzbad[NSET] = {};                        // BNF: items of the staged chunk that lie outside the image (their dz is zero, not kc)
  wg_i32x4 ozd[NSET]; unsigned ozs[NSET] = {};
  (void)writer;
  auto load_chunk = [&](auto set_c) {
    constexpr int S = decltype(set_c)::value;
    const int cw = ccw, oy = coy, b = cb_;
    if (++ccw == CW) { ccw = 0; coy += PR; if (coy >= a.Ho) { coy = 0; ++cb_; } }
    const bool real = left > 0;
    --left;
    const int ox0 = cw * PW;
    const wg_i32x4 dzd = wg_desc(a.dZ + (long long)b * a.Ho * a.Wo * a.Cout, real ? zimg_b : 0u);
    wg_i32x4 zzd = dzd;
    if constexpr (BNF) zzd = wg_desc(a.Z + (long long)b * a.Ho * a.Wo * a.Cout, real ? zimg_b : 0u);
    const wg_i32x4 xd = wg_desc((const char*)(a.X + (long long)b * a.Hi * a.Wi * a.Cin) - xshift, real ? ximg_b + xshift : 0u);
    const unsigned zso = (unsigned)__builtin_amdgcn_readfirstlane(((oy * a.Wo + ox0) * a.Cout) * 2);
    const unsigned xso = (unsigned)__builtin_amdgcn_readfirstlane(((oy * STRIDE * a.Wi + ox0 * STRIDE) * a.Cin) * 2);
    // first pixel / output row / patch column / patch row beyond the image (row limits from bit 8 on, as in dzpx / xjj: one compare
    // of the packed (row, column) against (row limit, 255) and one of the column)
    const int zlim = a.Wo - ox0, zrlim = a.Ho - oy, xlim = a.Wi - (ox0 * STRIDE - 1), xrlim = a.Hi - (oy * STRIDE - 1);
    const unsigned bt = (oy == 0 ? 1u : 0u) | (cw == 0 ? 2u : 0u);
#pragma unroll
    for (int i = 0; i < NDZ; ++i) {
      const bool bad = (CY_WG_DBG & 2) || (dzpx[i] & 255) >= zlim || (dzpx[i] >> 8) >= zrlim;
      wg_load(rdz[S][i], dzd, bad ? 0x80000000u : dzv[i], zso);
      if constexpr (BNF && !(CY_WG_DBG & 4)) {
        const unsigned m0v = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)(zlds + (S * NDZ + i) * NTHR * 16) + wave * 1024);
        unsigned keep;                              // (M0 is compiler-reserved: saved and restored inside the statement that uses it)
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %4 offen lds\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(bad ? 0x80000000u : dzv[i]), "s"(zzd), "s"(m0v), "s"(zso) : "memory");
      }
      if constexpr (BNF) dzbad[S] = (dzbad[S] & ~(1u << i)) | ((bad ? 1u : 0u) << i);
    }
    if constexpr (BNF) {                            // where this chunk's dz goes (stored when the set is); a phantom chunk or a block that shares the tile: nowhere
      ozd[S] = wg_desc(a.dZout + (long long)b * a.Ho * a.Wo * a.Cout, (real && writer) ? zimg_b : 0u);
      ozs[S] = zso;
    }
#pragma unroll
    for (int i = 0; i < NX; ++i) {
      const bool bad = (CY_WG_DBG & 2) || (((unsigned)xjj[i] >> 16) & bt) != 0u || (xjj[i] & 255) >= xlim || ((xjj[i] >> 8) & 255) >= xrlim;
      wg_load(rx[S][i], xd, bad ? 0x80000000u : xv[i], xso);
    }
  };
  auto wait_set = [&](auto set_c, auto n_c) {       // all but the n youngest vector-memory operations have landed; ties the set's registers
    constexpr int S = decltype(set_c)::value, N = decltype(n_c)::value;
#pragma unroll
    for (int i = 0; i < NDZ; ++i) { u32x4_t& r = rdz[S][i]; asm volatile("s_waitcnt vmcnt(%1)" : "+v"(r) : "n"(N)); }
    asm volatile("" ::: "memory");                  // (BNF: the z DMAs of the set have landed with its loads: they are older than its x loads)
#pragma unroll
    for (int i = 0; i < NX; ++i) { u32x4_t& r = rx[S][i]; asm volatile("s_waitcnt vmcnt(%1)" : "+v"(r) : "n"(N)); }
  };
  auto store_chunk = [&](auto set_c, int buf) {
    constexpr int S = decltype(set_c)::value;
#pragma unroll
    for (int i = 0; i < NDZ; ++i) {
      const int c = t + NTHR * i;
      const int px = c / (CO_T / 8), q = c % (CO_T / 8);
      if constexpr (BNF) {
        float d[8], z[8], o[8];
        unpack8(rdz[S][i], d);
        unpack8(*(const u32x4_t*)(zlds + ((S * NDZ + i) * NTHR + t) * 16), z);
        const float* kc = bncs + q * 8;
#pragma unroll
        for (int k = 0; k < 8; ++k) o[k] = (CY_WG_DBG & 16) ? d[k] + z[k] : __builtin_fmaf(d[k], kc[k], __builtin_fmaf(z[k], kc[CO_T + k], kc[2 * CO_T + k]));
        u32x4_t ov = pack8(o);
        if ((dzbad[S] >> i) & 1u) ov = u32x4_t{0u, 0u, 0u, 0u};       // outside the image: no pixel, no contribution
        if (c < DZ_CH) *(u32x4_t*)(dzimg + buf * DZ_IMG + px * DZB + q * 16) = ov;
        wg_store(ov, ozd[S], ((CY_WG_DBG & 8) || ((dzbad[S] >> i) & 1u)) ? 0x80000000u : dzv[i], ozs[S]);
      } else {
        if (c < DZ_CH) *(u32x4_t*)(dzimg + buf * DZ_IMG + px * DZB + q * 16) = rdz[S][i];
      }
    }
#pragma unroll
    for (int i = 0; i < NX; ++i) {
      const int c = t + NTHR * i;
      const int q = c % (CI_T / 8), pj = c / (CI_T / 8);
      if (c < X_CH) *(u32x4_t*)(ximg + buf * X_IMG + pj * XB + q * 16) = rx[S][i];
    }
  };

  // transposing-read lane roles: group g = lane >> 4 -> (channel block cb = g & 1, k half h = g >> 1); lane i of the
  // group supplies the address of pixel row q = i >> 2, channels 4 (i & 3) .. +3 of its block
  const int g4 = lane >> 4, li = lane & 15, q4 = li >> 2, p4 = li & 3;
  const int cb = g4 & 1, kh8 = 8 * (g4 >> 1);
  const int a_off = (kh8 + q4) * DZB + (wco * 32 + cb * 16 + 4 * p4) * 2;                 // + ks * 16 * DZB (+ 4 * DZB)
  const int b_off = ((kh8 + q4) * STRIDE) * XB + (wci * CI_W + cb * 16 + 4 * p4) * 2;     // + tap / tile / ks terms

  if constexpr (BNF) {
    for (int i = t; i < 3 * CO_T; i += NTHR) bncs[i] = a.bnc[(i / CO_T) * a.Cout + co0 + i % CO_T];
    __syncthreads();                                // (no load is in flight yet)
  }
  using S0 = std::integral_constant<int, 0>;
  using S1 = std::integral_constant<int, 1>;
  using N0 = std::integral_constant<int, 0>;
  load_chunk(S0{});
  wait_set(S0{}, N0{});
  if constexpr (BNF) asm volatile("s_barrier" ::: "memory");           // (its z came by LDS-DMA: wait, barrier, then read)
  store_chunk(S0{}, 0);
  if constexpr (NSET == 2) load_chunk(S0{});        // chunk c_lo + 1 (or a phantom) waits in set 0
  if constexpr (BNF && NSET == 2) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(NX) : "memory");   // its z DMAs: retired in front of the barrier
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");     // (not __syncthreads(): its fence would wait for the loads in flight)
  // chunk cid lies in LDS buffer P = (cid - c_lo) & 1; set P holds chunk cid + 1, set 1 - P receives chunk cid + 2
  auto body = [&](auto par_c, long long cid) {
    constexpr int P = decltype(par_c)::value;
    load_chunk(std::integral_constant<int, NSET == 2 ? 1 - P : 0>{});
    // Fragment reads as inline asm, the B fragments through a ring of three kept AHEAD of their MFMAs with counted waits: every MFMA needs
    // a fresh transposed B fragment, and left to hipcc each read stood right in front of its MFMA with its own wait (15 waits for 18
    // MFMAs; the ring holds three): an LDS round trip per MFMA and wave, which two waves per SIMD only half cover (mfma_busy 0.39).
    constexpr int NF = KHG * KH * NT, NKS = PIX / 16, NTOT = NKS * NF, RING = CY_WG_RING, ASL = 2;      // (2 ASL + 2 RING + 2 <= 15: lgkmcnt is a 4-bit counter)
    static_assert(2 * ASL + 2 * RING + 2 <= 15 && NF > RING && NKS % ASL == 0, "LDS reads in flight");
    const unsigned dza = (unsigned)(uintptr_t)(__attribute__((address_space(3))) const unsigned char*)(dzimg + P * DZ_IMG + a_off);
    const unsigned xba = (unsigned)(uintptr_t)(__attribute__((address_space(3))) const unsigned char*)(ximg + P * X_IMG + b_off + tg * KHG * PXW * XB);
    // the A fragments of TWO k slices at a time (all four up front were 16 registers): slice ks + 2 is requested into slice ks's pair
    // behind that slice's last MFMA -- younger than the B fragments in flight then, so the counted waits in front of those only wait
    // for more than they need, and NF MFMAs older than its own first use
    s16x4 alo[ASL], ahi[ASL], blo[RING], bhi[RING];
#pragma unroll
    for (int ks = 0; ks < ASL; ++ks) {
      asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(alo[ks]) : "v"(dza), "n"(ks * 16 * DZB));
      asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(ahi[ks]) : "v"(dza), "n"((ks * 16 + 4) * DZB));
    }
    auto boff = [](int i) constexpr -> int {
      const int ks = i / NF, f = i % NF, r = f / (KH * NT), sx = (f / NT) % KH, nt = f % NT;
      // k slice ks = 16 pixels: output row ks / 2 of the pair (STRIDE patch rows further down), columns 16 (ks % 2) ..
      return ((r + (ks / 2) * STRIDE) * PXW + sx + (ks % 2) * 16 * STRIDE) * XB + nt * 64;
    };
#pragma unroll
    for (int i = 0; i < RING; ++i) {
      asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(blo[i]) : "v"(xba), "n"(boff(i)));
      asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(bhi[i]) : "v"(xba), "n"(boff(i) + 4 * STRIDE * XB));
    }
#pragma unroll
    for (int i = 0; i < NTOT; ++i) {
      constexpr int dummy = 0; (void)dummy;
      const int younger = 2 * ((NTOT - 1 - i) < (RING - 1) ? (NTOT - 1 - i) : (RING - 1));
      s16x4& lo = blo[i % RING];
      s16x4& hi = bhi[i % RING];
      // (the A fragments were requested in front of every B fragment: landed whenever a B fragment has)
      // (tied only where a k slice starts: tied in every wait, hipcc copied them into fresh registers in front of every MFMA)
      if (i % NF == 0) {
        s16x4& a0 = alo[(i / NF) % ASL];
        s16x4& a1 = ahi[(i / NF) % ASL];
        asm volatile("" : "+v"(a0), "+v"(a1));
      }
      switch (younger) {
        case 6: asm volatile("s_waitcnt lgkmcnt(6)" : "+v"(lo), "+v"(hi)); break;
        case 4: asm volatile("s_waitcnt lgkmcnt(4)" : "+v"(lo), "+v"(hi)); break;
        case 2: asm volatile("s_waitcnt lgkmcnt(2)" : "+v"(lo), "+v"(hi)); break;
        default: asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(lo), "+v"(hi)); break;
      }
      typedef short s16x8 __attribute__((ext_vector_type(8)));
      const s16x8 av = __builtin_shufflevector(alo[(i / NF) % ASL], ahi[(i / NF) % ASL], 0, 1, 2, 3, 4, 5, 6, 7);
      const s16x8 bv = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
      const int f = i % NF;
#if CY_WG_DBG & 1
      if (i == 0)
#endif
      acc[f / NT][f % NT] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*(const bf16x8*)&av, *(const bf16x8*)&bv, acc[f / NT][f % NT], 0, 0, 0);
      if (i % NF == NF - 1 && i / NF + ASL < NKS) {     // the slice's last MFMA is issued: its A pair takes slice ks + ASL
        constexpr int dummy2 = 0; (void)dummy2;
        s16x4& a0 = alo[(i / NF) % ASL];
        s16x4& a1 = ahi[(i / NF) % ASL];
        asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(a0) : "v"(dza), "n"((i / NF + ASL < NKS ? i / NF + ASL : 0) * 16 * DZB));
        asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(a1) : "v"(dza), "n"(((i / NF + ASL < NKS ? i / NF + ASL : 0) * 16 + 4) * DZB));
      }
      if (i + RING < NTOT) {
        asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(lo) : "v"(xba), "n"(boff(i + RING < NTOT ? i + RING : 0)));
        asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(hi) : "v"(xba), "n"(boff(i + RING < NTOT ? i + RING : 0) + 4 * STRIDE * XB));
      }
    }
    // set P (requested an iteration ago): only this iteration's NL loads are younger (one set: the loads of this iteration)
    using SetC = std::integral_constant<int, NSET == 2 ? P : 0>;
    wait_set(SetC{}, std::integral_constant<int, NSET == 2 ? NL : 0>{});
    // BNF: LDS-DMA data is ordered for a ds_read only by the issuing wave's vmcnt wait FOLLOWED BY A BARRIER (the same lane reading its
    // own piece right behind the wait returned stale dwords now and then -- invisible while the stale bytes were a previous launch's
    // identical values).  One set: a barrier of its own behind the wait above.  Two sets: the z DMAs of the chunk requested at the top of
    // THIS iteration are retired in front of this iteration's end barrier (the NX loads and NDZ stores behind them stay in flight) and
    // read an iteration later.
    if constexpr (BNF && NSET == 1) asm volatile("s_barrier" ::: "memory");
    if (cid + 1 < c_hi) store_chunk(SetC{}, 1 - P);
    if constexpr (BNF && NSET == 2) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(NX + NDZ) : "memory");
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
  };
  for (long long cid = c_lo; cid < c_hi; cid += 2) {
    body(S0{}, cid);
    if (cid + 1 < c_hi) body(S1{}, cid + 1);
  }
  // (the loop's shape is fragile: with `else break;` behind the second body, or written with two breaks, hipcc spills 500 .. 840 bytes per
  // lane INCLUDING asm load destinations right behind their loads -- tools/check_vmcnt.py holds the kernel to zero scratch)
  // phantom loads may still be in flight and hipcc does not know: their registers must not be reused before they have landed
  wait_set(S0{}, N0{});
  if constexpr (NSET == 2) wait_set(S1{}, N0{});
  // slab[split][co][ci][kh][kw]
  float* slab = a.slabs + (long long)split * a.Cout * a.Cin * TAPS;
  const int lcol = lane & 31, lh = lane >> 5;
#pragma unroll
  for (int k = 0; k < TAPS / TG; ++k)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int co = co0 + wco * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
        const int ci = ci0 + wci * CI_W + nt * 32 + lcol;
        slab[((long long)co * a.Cin + ci) * TAPS + tg * (TAPS / TG) + k] = acc[k][nt][r];
      }
}

__global__ __launch_bounds__(256) void wgrad_bf16_reduce_kernel(const float* __restrict__ slabs, float* __restrict__ dW, int nsplit,
                                                               long long n) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  int k = 0;
  for (; k + 3 < nsplit; k += 4) {
    s0 += slabs[(long long)k * n + i]; s1 += slabs[(long long)(k + 1) * n + i];
    s2 += slabs[(long long)(k + 2) * n + i]; s3 += slabs[(long long)(k + 3) * n + i];
  }
  for (; k < nsplit; ++k) s0 += slabs[(long long)k * n + i];
  dW[i] = (s0 + s1) + (s2 + s3);
}

struct WgPlan { int kh, stride, wco, ci_t, co_t, ntiles_ci, ntiles, nsplit; long long nchunk; size_t lds; };
inline int wg_plan(int B, int Ho, int Wo, int Cin, int Cout, int KH, int stride, WgPlan* p) {
  if (!((KH == 3 && stride == 1) || (KH == 4 && stride == 2))) return 1;
  p->kh = KH; p->stride = stride;
  p->wco = (Cout % 128 == 0) ? 4 : 2;
  const int ci_w = KH == 3 ? 64 : 32;
  p->ci_t = ci_w * (4 / p->wco); p->co_t = 32 * p->wco;
  if (Cout % p->co_t || Cin % p->ci_t) return 1;
  if (KH == 3 && p->wco != 4) return 1;             // instantiated: <3,1,4>, <4,2,4>, <4,2,2>
  p->ntiles_ci = Cin / p->ci_t;
  p->ntiles = (Cout / p->co_t) * p->ntiles_ci;
  p->nchunk = (long long)B * ((Ho + 1) / 2) * ((Wo + 31) / 32);        // chunks of 2 output rows x 32 pixels
  // ONE round of blocks (a multiple of 8: XCD grouping): every block ends with its whole accumulator tile going to a slab (295 KB at
  // conv_2) that the reduce kernel reads back -- swept 256 / 512 / 768 / 1024 blocks: conv_4 / conv_5 0.29 / 0.38 / 0.46 / 0.56 ms,
  // conv_3 1.97 / 2.03 / 2.09 / 2.17, conv_2 6.08 / 6.14 / 6.16 / 6.23
  int ns = (256 / p->ntiles) & ~7;
#ifdef CY_WG_ENV
  if (const char* e = getenv("CY_WG_NS")) ns = (atoi(e) / p->ntiles) & ~7;   // developer sweep of the block count (builds with -DCY_WG_ENV only)
#endif
  if (ns < 8) ns = 8;
  while (ns > 8 && p->nchunk / ns < 8) ns -= 8;
  p->nsplit = ns;
  const int PXW = 31 * stride + KH;
  const int dzb = p->co_t * 2 + 64, xb = stride == 1 ? p->ci_t * 2 + 64 : (p->ci_t == 32 ? 96 : 160);
  p->lds = (size_t)2 * (64 * dzb + (KH + stride) * PXW * xb);
  return 0;
}

}  // namespace

extern "C" long long cy_conv_wgrad_bf16_ws_floats(int B, int Ho, int Wo, int Cin, int Cout, int KH, int stride);
namespace {
// bnc[3][N]: ka, kb, kc of dz = d ka + z kb + kc   (bn_apply_consts: cy_bn_bwd_apply_bf16's constants, slope 1)
__global__ void wgrad_bn_consts_kernel(const float* __restrict__ scale, const float* __restrict__ mean, const float* __restrict__ invstd,
                                       const double* __restrict__ red, double inv_count, float* __restrict__ bnc, int N) {
  const int n = blockIdx.x * blockDim.x + threadIdx.x;
  if (n >= N) return;
  bn_apply_consts(scale[n], mean[n], invstd[n], red[2 * n], red[2 * n + 1], inv_count, bnc[n], bnc[N + n], bnc[2 * N + n]);
}
}  // namespace

extern "C" long long cy_conv_wgrad_bf16_bn_ws_floats(int B, int Ho, int Wo, int Cin, int Cout, int KH, int stride) {
  const long long n = cy_conv_wgrad_bf16_ws_floats(B, Ho, Wo, Cin, Cout, KH, stride);
  return n < 0 ? n : n + 3ll * Cout;
}

extern "C" long long cy_conv_wgrad_bf16_ws_floats(int B, int Ho, int Wo, int Cin, int Cout, int KH, int stride) {
  WgPlan p;
  if (wg_plan(B, Ho, Wo, Cin, Cout, KH, stride, &p)) return -1;
  return (long long)p.nsplit * Cout * Cin * KH * KH;
}

static int wgrad_bf16_impl(const char* who, const void* X, const void* dZ, const void* Z, void* dZout, const float* bnc, float* dW, float* ws,
                           int B, int Hi, int Wi, int Cin, int Ho, int Wo, int Cout, int KH, int stride, hipStream_t s) {
  WgPlan p;
  CY_REQUIRE(wg_plan(B, Ho, Wo, Cin, Cout, KH, stride, &p) == 0,
             "%s: built for 3x3/s1 (Cout %% 128, Cin %% 64) and 4x4/s2 (Cout %% 64, Cin %% 32..64) with pad 1; got k=%d s=%d Cin=%d Cout=%d",
             who, KH, stride, Cin, Cout);
  CY_REQUIRE(Ho == (Hi + 2 - KH) / stride + 1 && Wo == (Wi + 2 - KH) / stride + 1, "%s: output size does not match pad 1", who);
  CY_REQUIRE((long long)(Hi + 2) * (Wi + 2) * Cin * 2 < (1ll << 31) && (long long)Ho * Wo * Cout * 2 < (1ll << 31),
             "%s: an image must stay below 2 GiB (32-bit buffer offsets; the offset 2^31 marks padding)", who);
  WgArgs a{(const u16*)X, (const u16*)dZ, ws, B, Hi, Wi, Cin, Ho, Wo, Cout, p.nsplit, p.ntiles_ci, p.nchunk, (const u16*)Z, (u16*)dZout, bnc};
  const unsigned grid = (unsigned)(p.ntiles * p.nsplit);
  const bool bnf = Z != nullptr;
  // BNF: + the constants and the z staging (one 16-byte piece per dz item and register set: wgrad_bf16_kernel's NDZ, NX, NSET)
  const int ndz = (64 * (p.co_t / 8) + 511) / 512, nx = ((KH + stride) * (31 * stride + KH) * (p.ci_t / 8) + 511) / 512;
  const size_t lds = p.lds + (bnf ? (size_t)3 * p.co_t * 4 + (size_t)(ndz + nx <= 5 ? 2 : 1) * ndz * 512 * 16 : 0);
  int rc;
#define CY_WG_LAUNCH(...)                                                                              \
  do {                                                                                               \
    if (bnf) { rc = cy_allow_lds(wgrad_bf16_kernel<__VA_ARGS__, true>, lds); if (rc) return rc;       \
               wgrad_bf16_kernel<__VA_ARGS__, true><<<grid, 512, lds, s>>>(a); }                      \
    else { rc = cy_allow_lds(wgrad_bf16_kernel<__VA_ARGS__, false>, lds); if (rc) return rc;          \
           wgrad_bf16_kernel<__VA_ARGS__, false><<<grid, 512, lds, s>>>(a); }                         \
  } while (0)
  if (KH == 3) CY_WG_LAUNCH(3, 1, 4, 8, 1);
  else if (p.wco == 4) CY_WG_LAUNCH(4, 2, 4, 8, 2);
  else CY_WG_LAUNCH(4, 2, 2, 8, 2);
#undef CY_WG_LAUNCH
  CY_LAUNCH_CHECK(who);
  const long long n = (long long)Cout * Cin * KH * KH;
  wgrad_bf16_reduce_kernel<<<(unsigned)cy_ceil_div(n, 256), 256, 0, s>>>(ws, dW, p.nsplit, n);
  CY_LAUNCH_CHECK(who);
  return 0;
}

extern "C" int cy_conv_wgrad_bf16(const void* X, const void* dZ, float* dW, float* ws, int B, int Hi, int Wi, int Cin, int Ho,
                                  int Wo, int Cout, int KH, int stride, void* stream) {
  CY_REQUIRE(X && dZ && dW && ws, "cy_conv_wgrad_bf16: null pointer");
  return wgrad_bf16_impl("cy_conv_wgrad_bf16", X, dZ, nullptr, nullptr, nullptr, dW, ws, B, Hi, Wi, Cin, Ho, Wo, Cout, KH, stride,
                         (hipStream_t)stream);
}

// The weight gradient with the block's BatchNorm backward (pass 2) on the way in: D = the premasked gradient dA * lrelu'(y) (bf16, what the
// consumer's fused input-gradient epilogue stored), Z the block's raw convolution output; dz = scale (d - m1 - xhat m2) is written to dZ
// (for the input-gradient kernel) and dW = its weight gradient; dgamma / dbeta from the sums `red` [N][2].  Replaces cy_bn_bwd_apply_bf16
// (slope 1) + cy_conv_wgrad_bf16: dz comes out bit-identical.  ws: cy_conv_wgrad_bf16_bn_ws_floats().
extern "C" int cy_conv_wgrad_bf16_bn(const void* X, const void* D, const void* Z, void* dZ, float* dW, float* ws, const float* scale,
                                     const float* mean, const float* invstd, const double* red, float* dgamma, float* dbeta, int B,
                                     int Hi, int Wi, int Cin, int Ho, int Wo, int Cout, int KH, int stride, void* stream) {
  CY_REQUIRE(X && D && Z && dZ && dW && ws && scale && mean && invstd && red, "cy_conv_wgrad_bf16_bn: null pointer");
  CY_REQUIRE(dZ != D && dZ != Z, "cy_conv_wgrad_bf16_bn: dZ must not alias its inputs (several blocks read a tile that one of them writes)");
  const long long base = cy_conv_wgrad_bf16_ws_floats(B, Ho, Wo, Cin, Cout, KH, stride);
  CY_REQUIRE(base >= 0, "cy_conv_wgrad_bf16_bn: unsupported layer k=%d s=%d Cin=%d Cout=%d", KH, stride, Cin, Cout);
  hipStream_t s = (hipStream_t)stream;
  float* bnc = ws + base;
  wgrad_bn_consts_kernel<<<(Cout + 255) / 256, 256, 0, s>>>(scale, mean, invstd, red, 1.0 / ((double)B * Ho * Wo), bnc, Cout);
  CY_LAUNCH_CHECK("cy_conv_wgrad_bf16_bn(constants)");
  int rc = wgrad_bf16_impl("cy_conv_wgrad_bf16_bn", X, D, Z, dZ, bnc, dW, ws, B, Hi, Wi, Cin, Ho, Wo, Cout, KH, stride, s);
  if (rc) return rc;
  if (dgamma && dbeta) {
    bn_param_grad2_kernel<<<(Cout + 255) / 256, 256, 0, s>>>(red, dgamma, dbeta, Cout);
    CY_LAUNCH_CHECK("cy_conv_wgrad_bf16_bn(param)");
  }
  return 0;
}
