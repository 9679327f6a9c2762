// Fused Winograd F(2x2, 3x3) convolution on the fp32 matrix cores (gfx950), for the 3x3 / stride 1 / pad 1
// layers (DarkCapsuleNet conv_2 = 77 % of the model's FLOPs, models.py:351; its input gradient; DarkNet's 3x3s).
//
// fp32 MFMA runs at the vector rate (157 TFLOP/s) and the direct implicit GEMM already sits at ~80 % of it, so
// the only way to be materially faster in exact fp32 is to do fewer multiplies:
//   Y = A^T [ sum_ci (G g G^T) (.) (B^T d B) ] A      -- 16 multiplies per 2x2 outputs instead of 36 (2.25x).
// Everything is fused in one kernel so that neither the transformed input (4x the activation) nor the 16
// partial products ever touch HBM:
//   block = 8x8 tiles (16x16 output pixels) x 64 output channels, 4 waves (2 x 2), ONE wave per SIMD with the
//   whole 512-register file: each wave holds all 16 Winograd positions of its 32 tiles x 32 channels
//   (16 accumulator tiles of mfma_f32_32x32x2f32 = 256 registers), so the output transform is lane-local.
//   Per chunk of 8 input channels: raw 18x18 input patch -> LDS; every thread transforms part of a tile
//   (B^T d B, adds only) into the 16 A images V[xi][k/4][tile][4]; the pre-transformed weights U[xi][k/4][co][4]
//   are copied to LDS; each wave then issues 16 positions x 4 MFMAs.  The transform of chunk c+1 and the
//   global loads of chunk c+2 are issued between the MFMAs of chunk c (three-stage software pipeline, two
//   barriers per chunk).
#include "common.h"

// Nontemporal stores of the (streamed once, 5.7 GB) output keep it from pushing the transformed weights, which every tile
// re-reads, out of the XCD's 4 MB L2: conv_2 forward fetches 4.9 instead of 8.5 GB -- and runs 13.62 instead of 13.28 ms (the
// drain of a wave that is alone on its SIMD waits longer on them), so they stay OFF here; the stride-2 kernels
// (winograd_s2.hip), where the same switch is free, use them.
#ifndef CY_NT
#define CY_NT 0
#endif
#if CY_NT
#define CY_NT_STORE(v, p) __builtin_nontemporal_store((v), (p))
#else
#define CY_NT_STORE(v, p) (*(p) = (v))
#endif


namespace {

constexpr int WT = 64;                      // tiles per block (8 x 8)
constexpr int WN = 64;                      // output channels per block
constexpr int KC = 8;                       // input channels per chunk
constexpr int SLAB = WT * 4 + 4;            // floats per (xi, kq) slab of V or U (16 B pad; 16 floats of pad, conflict-free V stores, measured no faster)
constexpr int VU_BUF = 32 * SLAB;           // 16 positions x 2 k-quads
constexpr int RAWP = 337;                   // 18*18 = 324 pixels, padded: the k-quad stride is 4 banks (mod 64)
constexpr int RAW_BUF = 2 * RAWP * 4;       // [kq][pixel][4]

// zeros that padding items of a patch are loaded from (the chunk offset, < 4 Cin bytes, is added to every item's pointer)
constexpr int WINO_MAX_CIN = 2048;
__device__ __attribute__((aligned(256))) float wino_zero_pad[WINO_MAX_CIN];

struct WinoArgs {
  const float* X; const float* U; float* Y; const float* bias; double* stats;
  int B, H, W, Cin, Cout, Np, tbh, tbw;
  int ntiles;                               // B * tbh * tbw * Np/64 output tiles, walked by a persistent grid
  float out_slope;                          // EPI == 2: Y = lrelu(conv + bias) with this slope (eval mode, BatchNorm folded into U / bias)
  // split of the reduction (round 4; grid.y = shares): a launch whose tiles fill at most half the CUs (DarkNet's 13 x 13 input
  // gradients: 128 tiles) runs grid.y blocks per tile, share y on input channels y * Cin .. (y + 1) * Cin - 1 of the Cs channels a pixel
  // has, into its own slab Y + y * yslab; wino_split_sum adds the slabs in share order
  int Cs;
  long long yslab;
};

__device__ __forceinline__ f32x16 mfma_zero() {
  f32x16 c;
  asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %1, 0" : "=a"(c) : "v"(0.f));
  return c;
}
__device__ __forceinline__ f32x16 mfma32(float a, float b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
}

// two fp32 adds in one VALU instruction (the compiler splits most float2 adds into two v_add_f32)
__device__ __forceinline__ f32x2 pk_add(f32x2 x, f32x2 y) {
  f32x2 r;
  asm("v_pk_add_f32 %0, %1, %2" : "=v"(r) : "v"(x), "v"(y));
  return r;
}
__device__ __forceinline__ f32x2 pk_fma(f32x2 x, f32x2 y, f32x2 z) {
  f32x2 r;
  asm("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(r) : "v"(x), "v"(y), "v"(z));
  return r;
}
__device__ __forceinline__ f32x2 pk_sub(f32x2 x, f32x2 y) {
  f32x2 r;
  asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(r) : "v"(x), "v"(y));
  return r;
}

// One accumulator element, read where the statement stands.  Plain `acc[xi][r]` lets the compiler copy ALL 16
// accumulator vectors AGPR -> VGPR in front of the output transform (256 VGPRs: everything else is spilled).
__device__ __forceinline__ float acc_elem(float a_elem) {
  float x;
  asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(x) : "a"(a_elem));
  return x;
}

// A fragment read the compiler does not track: with LDS-DMA instructions in the loop hipcc answers every tracked ds_read
// result with `s_waitcnt lgkmcnt(0)` at its first use (tools/check_lds_waits.py showed it in front of every second
// position), which throws the counted waits of the schedule away.  The slot's explicit wait is tied to these registers.
template <int BYTE_OFF> __device__ __forceinline__ void lds_read128(f32x4& dst, const float* p) {   // (immediate offset: no VALU add per read)
  const unsigned a_ = (unsigned)(uintptr_t)(__attribute__((address_space(3))) const float*)p;
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(a_), "n"(BYTE_OFF));
}
// two float2 (16 bytes apart in units of 8: OFF0, OFF1) in one untracked read
template <int OFF0, int OFF1> __device__ __forceinline__ void lds_read2_b64(f32x4& dst, const float* p) {
  const unsigned a_ = (unsigned)(uintptr_t)(__attribute__((address_space(3))) const float*)p;
  asm volatile("ds_read2_b64 %0, %1 offset0:%2 offset1:%3" : "=v"(dst) : "v"(a_), "n"(OFF0), "n"(OFF1));
}
__device__ __forceinline__ void mfma32_inplace(f32x16& c, float a, float b) {
  asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+a"(c) : "v"(a), "v"(b));
}
template <int N> __device__ __forceinline__ void lgkm_wait(f32x4& x, f32x4& y) {
  asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(x), "+v"(y) : "n"(N));
}

// ---- compile-time schedule of one chunk (64 slots = 64 MFMAs per wave)
// Slot s issues the MFMA of position xi(s), k-step e(s); positions are interleaved in pairs so that consecutive
// MFMAs never share an accumulator:  (x0,e0) (x1,e0) (x0,e1) (x1,e1) ... (x0,e3) (x1,e3) (x2,e0) ...
constexpr int wino_xi(int s) { return (s & 1) + 2 * (s >> 3); }
constexpr int wino_e(int s) { return (s >> 1) & 3; }
// the fragments of position xi are fetched right after the MFMA of slot wino_issue(xi) and first used in slot wino_use(xi)
constexpr int wino_use(int xi) { return 8 * (xi >> 1) + (xi & 1); }
constexpr int wino_issue(int xi) { return wino_use(xi) - 8; }              // < 0: fetched before the loop body
constexpr int wino_frag_pos(int s) { return ((s & 7) < 2 && s + 8 < 64) ? wino_xi(s) + 2 : -1; }   // position prefetched in slot s
// side work, one piece per slot (see the slot body):
//   7 S_raw patch of chunk c+2 registers -> LDS (3)
//   3 D_U   U(c+1) global -> LDS by LDS-DMA (global_load_lds_dwordx4: one 1 KiB slab per piece and wave, 8): no staging
//           registers and no ds_write_b128 (13 issue cycles each, and the LDS store path was the chunk's scarce resource)
//   2 G_raw patch loads of chunk c+3 (3), after their registers were stored and AFTER the DMA pieces: the barrier's
//           vmcnt(3) then retires exactly the DMA (memory operations return in order)
//   4 T_rd  patch of chunk c+1 from LDS, two float2 per piece, rows in the order 1,2,0,3 (8)
//   5 T_v   half a row of V = B^T d B: 5 packed adds + 2 LDS writes (8, rows in the order 1,2,0,3)
// pieces of different kinds are interleaved and the loads spread out: 11 global loads in consecutive slots back up
// the CU's one vector-memory pipeline (a patch load touches 32 cache lines) and the stalled wave stops issuing MFMAs
constexpr int wino_find(const int* list, int n, int s) {
  for (int i = 0; i < n; ++i) if (list[i] == s) return i;
  return -1;
}
// LDS WRITE bandwidth is the scarce resource of the chunk (U 32 KB + patch 10 KB + V 32 KB per chunk against ~70 B/clk/CU:
// s_memtime stamps showed the 8 back-to-back U stores of all four waves holding the first four slots for 600 extra
// cycles, every 4-store V row for 100-200): every piece stores at most 2 x 512 B or 1 x 1 KB per wave, and store pieces
// alternate with pieces that do not store.
constexpr int WS_SRAW[3] = {2, 3, 4};
constexpr int WS_GU[8] = {5, 6, 7, 10, 11, 12, 13, 14};
constexpr int WS_GRAW[3] = {15, 18, 19};
// (T_v pieces in consecutive slots -- 42..47, 50, 51 -- cost 1.5-2 % more: their 16 ds_write_b64 per thread back up)
constexpr int WS_TRD[8] = {20, 21, 22, 23, 26, 27, 28, 29};
constexpr int WS_TV[8] = {30, 34, 36, 38, 42, 44, 46, 50};
constexpr int WS_BAR = 54;                  // barrier; the fragments of the next chunk's positions 0 / 1 follow in slots 56 / 57
// developer knob for timing experiments (results are wrong when set): drop 1 the V transform pieces, 2 the patch reads,
// 4 the U DMA, 8 the patch loads / stores, 16 the chunk barrier, 32 the accumulator drain
#ifndef CY_WINO_DBG
#define CY_WINO_DBG 0
#endif
constexpr int WDBG = CY_WINO_DBG;
constexpr int wino_side_kind_all(int s);
constexpr int wino_side_kind(int s) {
  const int k = wino_side_kind_all(s);
  return ((WDBG & 1) && k == 5) || ((WDBG & 2) && k == 4) || ((WDBG & 4) && k == 3) || ((WDBG & 8) && (k == 2 || k == 7)) ? 0 : k;
}
constexpr int wino_side_kind_all(int s) {
  return wino_find(WS_GRAW, 3, s) >= 0 ? 2 : wino_find(WS_GU, 8, s) >= 0 ? 3
       : wino_find(WS_TRD, 8, s) >= 0 ? 4 : wino_find(WS_TV, 8, s) >= 0 ? 5 : wino_find(WS_SRAW, 3, s) >= 0 ? 7 : 0;
}
constexpr int wino_side_idx(int s) {
  const int k = wino_side_kind(s);
  return k == 2 ? wino_find(WS_GRAW, 3, s) : k == 3 ? wino_find(WS_GU, 8, s)
       : k == 4 ? wino_find(WS_TRD, 8, s) : k == 5 ? wino_find(WS_TV, 8, s) : k == 7 ? wino_find(WS_SRAW, 3, s) : 0;
}
constexpr int wino_row_order(int i) { return i == 0 ? 1 : i == 1 ? 2 : i == 2 ? 0 : 3; }
constexpr int wino_side_lds(int s) {
  const int k = wino_side_kind(s);
  // a LOWER bound of the LDS instructions the slot issues (the waits below may never allow more outstanding
  // operations than are really younger): two float2 reads may merge into one ds_read2_b64, the last S_raw store
  // is exec-masked and may be skipped by a whole wave
  return k == 5 ? 2 : (k == 4) ? 1 : (k == 7 && wino_side_idx(s) < 2) ? 1 : 0;
}
constexpr int wino_frag_lds(int s) { return wino_frag_pos(s) >= 0 ? 2 : 0; }
// LDS instructions (lower bound) issued between the last patch read that transform piece `piece` needs and that piece:
// piece 0 needs T_rd pieces 0..3, piece 4 needs 4, 5, piece 6 needs 6, 7
constexpr int wino_trd_younger(int piece) {
  const int last = WS_TRD[piece == 0 ? 3 : piece == 4 ? 5 : 7], use = WS_TV[piece];
  int n = 0;
  for (int s = last + 1; s < use; ++s) n += wino_frag_lds(s) + wino_side_lds(s);
  return n > 14 ? 14 : n;
}
// LDS operations younger than position xi's fragments when its first MFMA issues (s_waitcnt lgkmcnt operand)
constexpr int wino_younger(int xi) {
  const int is = wino_issue(xi), us = wino_use(xi);
  int n = 0;
  if (is < 0) {                            // prologue order: frags of x0, then x1
    if (xi == 0) n += 2;                   // x1's fragments are younger than x0's
    for (int s = 0; s < us; ++s) n += wino_frag_lds(s) + wino_side_lds(s);
  } else {
    n += wino_side_lds(is);
    for (int s = is + 1; s < us; ++s) n += wino_frag_lds(s) + wino_side_lds(s);
  }
  return n > 14 ? 14 : n;
}

// EPI: epilogue variant at kernel level (a test inside the accumulator rows was a uniform branch per row, ~40 cycles each on a
// wave that is alone on its SIMD): 0 plain (input gradient), 1 BatchNorm statistics (training forward), 2 LeakyReLU (eval
// forward with the BatchNorm folded into weights and bias: models.py:349-351 in eval mode is conv' -> LeakyReLU)
template <int EPI>
__global__ __launch_bounds__(256, 1) void wino_conv_kernel(WinoArgs a) {
  constexpr bool STATS = EPI == 1;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* Vs = smem;                         // [2][VU_BUF]
  float* Us = smem + 2 * VU_BUF;            // [2][VU_BUF]
  float* Rs = smem + 4 * VU_BUF;            // [2][RAW_BUF]

  const int t = threadIdx.x, lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int wm = wave & 1, wn = wave >> 1;
  const int li = lane & 31, lh = lane >> 5;

  // Persistent grid (one block per CU): block j walks the tiles vid(j), vid(j) + G, vid(j) + 2G, ... as ONE
  // continuous stream of chunks: while the MFMAs of a tile's last chunks run, the loads and the input transform of the
  // next tile's first chunks are already in flight, so only the accumulator drain (output transform + stores)
  // separates two tiles -- no per-tile prologue latency, no block launch gap (338 tiles per CU at the headline shape).
  // Consecutive block ids go round-robin over the 8 XCDs: each XCD gets a contiguous range of virtual ids, so that the
  // Np/64 tiles that read the same 18x18 patch are worked on at the same time on ONE XCD and share its L2.
  unsigned vid = blockIdx.x;
  if ((gridDim.x & 7u) == 0) vid = (blockIdx.x & 7u) * (gridDim.x >> 3) + (blockIdx.x >> 3);
  const int nblk = a.Np / WN;
  const int ntile_mine = (a.ntiles - (int)vid + (int)gridDim.x - 1) / (int)gridDim.x;   // >= 1 (grid <= ntiles)
  const int nchunk = a.Cin / KC;
  const int share = blockIdx.y;                        // (0 unless the reduction is split)
  const int c0 = share * nchunk;                       // this share's first chunk of U
  struct TilePos { int nb, b, oy0, ox0; };
  auto tile_pos = [&](int k) {              // k-th tile of this block (uniform)
    const int id = (int)vid + k * (int)gridDim.x;
    TilePos p;
    p.nb = id % nblk;
    int rest = id / nblk;
    const int tbx = rest % a.tbw; rest /= a.tbw;
    const int tby = rest % a.tbh;
    p.b = rest / a.tbh; p.oy0 = tby * 16; p.ox0 = tbx * 16;
    return p;
  };

  // ---- per-thread constants of the loaders (those of the raw-patch loader belong to the tile its stream is in)
  // raw patch: 648 float4 items over 256 threads x 3; 32 consecutive items = 16 pixels x 2 k-quads with the pixel in
  // the low 4 bits, so that the 16 lanes of one ds_write_b128 pass hit 16 different pixels of one k-quad (no bank
  // conflict) while a wave's global load still covers both 16-byte halves of each pixel's 32 bytes
  // No branch and no exec mask may stand in the MFMA slots (a uniform branch costs this one-wave-per-SIMD loop ~40
  // cycles): every item has ONE 64-bit lane pointer per tile -- its pixel, or, for padding, a block of zeros
  // (wino_zero_pad) -- so loads and LDS stores are unconditional.  The third round is partial (pixels 256..323 only):
  // the other threads repeat their second item (same address, same LDS slot, same value).
  const char* gptr[3];                      // channel 0 of the item's k-quad in the tile the patch stream is in
  int roff[3];
  int rpy[3], rpx[3];
#pragma unroll
  for (int q = 0; q < 3; ++q) {
    const int pix2 = ((t + 512) >> 5) * 16 + (t & 15);               // pixel of the third item: only 324 exist
    const int item = t + 256 * ((q < 2 || pix2 < 324) ? q : 1);
    const int pix = (item >> 5) * 16 + (item & 15), kq = (item >> 4) & 1;
    rpy[q] = pix / 18; rpx[q] = pix - rpy[q] * 18;
    roff[q] = (kq * RAWP + pix) * 4;
  }
  const int kq_of_thread = (t >> 4) & 1;    // 256 q keeps bit 4 of the item
  auto set_raw_tile = [&](int k) {
    const TilePos p = tile_pos(k);
    const char* img = (const char*)(a.X + (long long)p.b * a.H * a.W * a.Cs + (long long)share * a.Cin) + kq_of_thread * 16;
#pragma unroll
    for (int q = 0; q < 3; ++q) {
      const int iy = p.oy0 - 1 + rpy[q], ix = p.ox0 - 1 + rpx[q];
      const bool ok = (unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W;
      gptr[q] = ok ? img + (size_t)((iy * a.W + ix) * a.Cs) * 4 : (const char*)wino_zero_pad + kq_of_thread * 16;
    }
  };
  // stream cursors (tile index within this block, chunk): advance by one chunk, stop at the very last chunk
  auto advance = [&](int& k, int& c) {
    if (c + 1 < nchunk) { ++c; return false; }
    if (k + 1 < ntile_mine) { ++k; c = 0; return true; }
    return false;
  };
  // U chunk: 32 segments (xi*2+kq) of 64 channels x float4; thread -> float4 f = t + 256q
  const long long useg = (long long)4 * a.Np * 4;      // 4 segments further per q
  const long long uchunk = (long long)32 * a.Np * 4;
  const unsigned uvoff = (unsigned)(((t >> 6) * a.Np + (t & 63)) * 16);   // bytes from the chunk's first float of the co block
  const char* ubase = nullptr;              // uniform: a.U + first output channel of the U stream's tile
  auto set_u_tile = [&](int k) { ubase = (const char*)(a.U + (long long)tile_pos(k).nb * WN * 4); };
  const int uslab = wave * SLAB;                       // the wave's DMA piece q fills slab wave + 4 q (uniform LDS address)
  // transform item: channel pair tch = t & 1 of k-quad tkq = (t >> 1) & 1, tile column (t >> 2) & 7, tile row t >> 5:
  // the thread computes all 16 positions of V for two channels, every add is one v_pk_add_f32 on a float2 that came
  // out of LDS as a pair.  With RAWP = 1 (mod 16) the 64 lanes of one float2 patch read (offsets 2 tch + 4 tkq +
  // 8 tx + 16 ty_low mod 64 banks) are 2-way conflicted, the minimum for 512 bytes (8-way with tile-major lanes).
  const int tch = t & 1, tkq = (t >> 1) & 1, ttile = t >> 2;
  const int tbase = (tkq * RAWP + (2 * (ttile >> 3)) * 18 + 2 * (ttile & 7)) * 4 + 2 * tch;   // patch pixel (0,0)
  const int vdst = tkq * SLAB + ttile * 4 + 2 * tch;                                          // + xi * 2 * SLAB

  f32x4 graw[3];

  // global -> registers (unconditional loads: out-of-image pixels read a valid address and are zeroed, so the
  // chunk body stays straight-line code and hipcc's vmcnt counts stay exact)
  auto Graw = [&](int c, f32x4 (&dst)[3]) {
#pragma unroll
    for (int q = 0; q < 3; ++q) dst[q] = *(const f32x4*)(gptr[q] + (size_t)c * (KC * 4));
  };
  auto DU = [&](int c, int buf) {                   // U chunk c -> LDS U buffer `buf` by LDS-DMA
    float* ub = Us + buf * VU_BUF + uslab;
#pragma unroll
    for (int q = 0; q < 8; ++q)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(ubase + ((long long)(c + c0) * uchunk + q * useg) * 4 + uvoff),
                                       (__attribute__((address_space(3))) void*)(ub + q * 4 * SLAB), 16, 0, 0);
  };
  auto Sraw = [&](int c, const f32x4 (&src)[3]) {   // registers -> LDS raw patch buffer c & 1
    float* rb = Rs + (c & 1) * RAW_BUF;
#pragma unroll
    for (int q = 0; q < 3; ++q) *(f32x4*)(rb + roff[q]) = src[q];
  };
  f32x2 xv[4][4];                           // the thread's 4x4 patch, two channels
  f32x4 xq[4][2];                           // the same patch inside the chunk loop: row r, columns (0,1) / (2,3), read by untracked asm
  auto XQ = [&](int r, int cc) -> f32x2 {
    return (cc & 1) ? __builtin_shufflevector(xq[r][cc >> 1], xq[r][cc >> 1], 2, 3) : __builtin_shufflevector(xq[r][cc >> 1], xq[r][cc >> 1], 0, 1);
  };
  auto Vrow = [&](float* vb, int R) {       // row R of V = B^T d B
    f32x2 t0[4];
#pragma unroll
    for (int cc = 0; cc < 4; ++cc)
      t0[cc] = R == 0 ? pk_sub(xv[0][cc], xv[2][cc]) : R == 1 ? pk_add(xv[1][cc], xv[2][cc])
             : R == 2 ? pk_sub(xv[2][cc], xv[1][cc]) : pk_sub(xv[1][cc], xv[3][cc]);
    *(f32x2*)(vb + (R * 4 + 0) * 2 * SLAB) = pk_sub(t0[0], t0[2]);
    *(f32x2*)(vb + (R * 4 + 1) * 2 * SLAB) = pk_add(t0[1], t0[2]);
    *(f32x2*)(vb + (R * 4 + 2) * 2 * SLAB) = pk_sub(t0[2], t0[1]);
    *(f32x2*)(vb + (R * 4 + 3) * 2 * SLAB) = pk_sub(t0[1], t0[3]);
  };
  f32x2 t0h[4];                             // the row's column differences, computed by the first half, used by both
  auto Vhalf = [&](float* vb, int R, int jp) {   // columns 2 jp, 2 jp + 1 of row R
    if (jp == 0) {
#pragma unroll
      for (int cc = 0; cc < 4; ++cc)
        t0h[cc] = R == 0 ? pk_sub(XQ(0, cc), XQ(2, cc)) : R == 1 ? pk_add(XQ(1, cc), XQ(2, cc))
                : R == 2 ? pk_sub(XQ(2, cc), XQ(1, cc)) : pk_sub(XQ(1, cc), XQ(3, cc));
      *(f32x2*)(vb + (R * 4 + 0) * 2 * SLAB) = pk_sub(t0h[0], t0h[2]);
      *(f32x2*)(vb + (R * 4 + 1) * 2 * SLAB) = pk_add(t0h[1], t0h[2]);
    } else {
      *(f32x2*)(vb + (R * 4 + 2) * 2 * SLAB) = pk_sub(t0h[2], t0h[1]);
      *(f32x2*)(vb + (R * 4 + 3) * 2 * SLAB) = pk_sub(t0h[1], t0h[3]);
    }
  };
  auto T = [&](int c) {                     // raw patch -> V (this thread's tile, 2 channels)
    const float* rb = Rs + (c & 1) * RAW_BUF + tbase;
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int cc = 0; cc < 4; ++cc) xv[r][cc] = *(const f32x2*)(rb + (r * 18 + cc) * 4);
    float* vb = Vs + (c & 1) * VU_BUF + vdst;
#pragma unroll
    for (int R = 0; R < 4; ++R) Vrow(vb, R);
  };

  // ---- prologue (once per block).  Pipeline state at the top of stream position f (tile km, chunk cm):
  // V[f&1] / U[f&1] = transformed input / weights of position f, raw[(f+1)&1] = input patch of position f+1,
  // registers graw = patch of position f+2; the cursors (ku, cu) and (kr, cr) stand at the positions whose loads are
  // issued during f: f+1 for U (LDS-DMA into U[(f+1)&1]), f+3 for the patch.
  int km = 0, cm = 0, ku = 0, cu = 0, kr = 0, cr = 0;
  {
    f32x4 graw1[3];
    set_raw_tile(0);
    set_u_tile(0);
    Graw(0, graw);
    DU(0, 0);
    if (advance(kr, cr)) set_raw_tile(kr);
    Graw(cr, graw1);
    Sraw(0, graw);
    if (advance(kr, cr)) set_raw_tile(kr);
    Graw(cr, graw);
    __syncthreads();                        // (its fence waits vmcnt(0): U(0) has landed)
    T(0);
    Sraw(1, graw1);
    __syncthreads();
    if (advance(kr, cr)) set_raw_tile(kr);
    if (advance(ku, cu)) set_u_tile(ku);
  }
  const int fragA = lh * SLAB + (wm * 32 + li) * 4;
  const int fragB = lh * SLAB + (wm * 0 + wn * 32 + li) * 4;
  // One chunk = 64 MFMAs per wave.  There is ONE wave per SIMD, and a wave issues in order: while it waits to
  // issue the next MFMA (the pipe is busy for 64 cycles) nothing behind that MFMA can issue.  So every other
  // piece of work of the pipeline -- fragment reads for the next position, the input transform of position f+1
  // (LDS reads, adds, LDS writes) and the global loads of positions f+2 / f+3 -- is cut into ~30 small pieces and
  // ONE piece is placed after each MFMA in program order (pinned with sched_barrier).  LDS operations complete in
  // order, so the fragments of position xi+1 (read right after position xi's first MFMA) are ready when
  // `lgkmcnt(N)` is checked.
  // The chunk's barrier stands at slot WS_BAR, not at the end: by then every wave has issued all its reads of
  // V/U[f&1] (the last fragments are fetched in slot 49) and all its writes of V/U[(f+1)&1] and raw; the MFMAs of the
  // remaining slots need nothing from LDS, and under them the first two fragment sets of position f+1 are fetched,
  // so that neither the barrier skew nor an LDS round trip stands between two chunks.
  int c_next = 0;                           // stream position f (only its parity is used)
  f32x4 fa_[4], fb_[4];                     // fragment sets, indexed by position & 3; [0], [1] are loaded a chunk ahead
  lds_read128<0>(fa_[0], Vs + fragA);
  lds_read128<0>(fb_[0], Us + fragB);
  lds_read128<2 * SLAB * 4>(fa_[1], Vs + fragA);
  lds_read128<2 * SLAB * 4>(fb_[1], Us + fragB);
  f32x16 acc[16];
#pragma unroll
  for (int xi = 0; xi < 16; ++xi)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[xi][r] = 0.f;
  for (km = 0; km < ntile_mine; ++km) {
  // the tile's bias is fetched HERE, a whole tile ahead of the drain that adds it (loaded at the drain's top, its L2
  // round trip stood in front of the first accumulator row of every tile)
  const int co_mine = tile_pos(km).nb * WN + wn * 32 + li;
  const float bv = (a.bias != nullptr && co_mine < a.Cout) ? a.bias[co_mine] : 0.f;
  for (cm = 0; cm < nchunk; ++cm, ++c_next) {
    const int c = c_next;
    const float* vb_ = Vs + (c & 1) * VU_BUF + fragA;
    const float* ub_ = Us + (c & 1) * VU_BUF + fragB;
    const float* rb_ = Rs + ((c + 1) & 1) * RAW_BUF + tbase;        // T(f+1) reads ...
    float* vw_ = Vs + ((c + 1) & 1) * VU_BUF + vdst;                // ... and writes (harmless after the last position)
    const size_t gx = (size_t)cr * (KC * 4);                        // G_raw(f+3) (the cursors stop at the last position:
                                                                    //  the tail re-loads valid data), uniform
    const char* gusrc = ubase + (long long)(cu + c0) * uchunk * 4;    // D_U(f+1), uniform (harmless after the last position)
    float* uw_ = Us + ((c + 1) & 1) * VU_BUF + uslab;
    float* rw_ = Rs + (c & 1) * RAW_BUF;                            // S_raw(f+2) -> raw[(f+2)&1]
    const float* vn_ = Vs + ((c + 1) & 1) * VU_BUF + fragA;         // fragments of position f+1
    const float* un_ = Us + ((c + 1) & 1) * VU_BUF + fragB;
#define WSLOT(SIDX)                                                                                 \
    {                                                                                               \
      constexpr int sidx = (SIDX);                                                                  \
      constexpr int xi = wino_xi(sidx), e = wino_e(sidx);                                           \
      if (e == 0) lgkm_wait<wino_younger(xi)>(fa_[xi & 3], fb_[xi & 3]);                            \
      mfma32_inplace(acc[xi], fa_[xi & 3][e], fb_[xi & 3][e]);   /* volatile asm: stays in front of the slot's reads */ \
      constexpr int fp = wino_frag_pos(sidx);                                                       \
      if (fp >= 0) {                                                                                \
        constexpr int fq = fp >= 0 ? fp : 0;                                                        \
        lds_read128<fq * 2 * SLAB * 4>(fa_[fq & 3], vb_);                                           \
        lds_read128<fq * 2 * SLAB * 4>(fb_[fq & 3], ub_);                                           \
      }                                                                                             \
      constexpr int kind = wino_side_kind(sidx), k_ = wino_side_idx(sidx);                          \
      if (kind == 2) {                      /* patch loads of chunk c+3 */                         \
        graw[k_] = *(const f32x4*)(gptr[k_] + gx);                                                  \
      } else if (kind == 3) {               /* weights of chunk c+1: one slab, global -> LDS */    \
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gusrc + k_ * (useg * 4) + uvoff), \
                                         (__attribute__((address_space(3))) void*)(uw_ + k_ * 4 * SLAB), 16, 0, 0);        \
      } else if (kind == 4) {               /* patch of chunk c+1: two float2 */                   \
        constexpr int r_ = wino_row_order((k_ >> 1) & 3), c0_ = 2 * (k_ & 1);                       \
        lds_read2_b64<(r_ * 18 + c0_) * 2, (r_ * 18 + c0_ + 1) * 2>(xq[r_][k_ & 1], rb_);           \
      } else if (kind == 5) {               /* half a row of V */                                  \
        /* rows 1, 2 (pieces 0..3 of T_rd) are first used by piece 0, row 0 by piece 4, row 3 by piece 6 */ \
        if (k_ == 0) asm volatile("s_waitcnt lgkmcnt(%4)" : "+v"(xq[1][0]), "+v"(xq[1][1]), "+v"(xq[2][0]), "+v"(xq[2][1]) : "n"(wino_trd_younger(0))); \
        if (k_ == 4) asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(xq[0][0]), "+v"(xq[0][1]) : "n"(wino_trd_younger(4))); \
        if (k_ == 6) asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(xq[3][0]), "+v"(xq[3][1]) : "n"(wino_trd_younger(6))); \
        Vhalf(vw_, wino_row_order((k_ >> 1) & 3), k_ & 1);                                          \
      } else if (kind == 7) {               /* patch of chunk c+2: registers -> LDS */             \
        *(f32x4*)(rw_ + roff[k_]) = graw[k_];                                                       \
      } else if (sidx == WS_BAR) {          /* the only barrier of the chunk: lgkmcnt(0), and vmcnt(3) = the DMA */ \
        __builtin_amdgcn_s_waitcnt(0x0073); /* pieces have landed, the three younger patch loads stay in flight */  \
        if (!(WDBG & 16)) __builtin_amdgcn_s_barrier();                                             \
      } else if (sidx == WS_BAR + 2) {      /* positions 12, 13 are done with sets 0 and 1 */      \
        lds_read128<0>(fa_[0], vn_);                                                                \
        lds_read128<0>(fb_[0], un_);                                                                \
      } else if (sidx == WS_BAR + 3) {                                                              \
        lds_read128<2 * SLAB * 4>(fa_[1], vn_);                                                     \
        lds_read128<2 * SLAB * 4>(fb_[1], un_);                                                     \
      }                                                                                             \
      __builtin_amdgcn_sched_barrier(0);                                                            \
    }
#define WSLOT4(B) WSLOT((B)) WSLOT((B) + 1) WSLOT((B) + 2) WSLOT((B) + 3)
#define WSLOT16(B) WSLOT4((B)) WSLOT4((B) + 4) WSLOT4((B) + 8) WSLOT4((B) + 12)
    WSLOT16(0) WSLOT16(16) WSLOT16(32) WSLOT16(48)
#undef WSLOT16
#undef WSLOT4
#undef WSLOT
    if (advance(kr, cr)) set_raw_tile(kr);
    if (advance(ku, cu)) set_u_tile(ku);
  }
  if constexpr (!(WDBG & 32)) {
    // ======== tile km is complete: drain the accumulators (c is now the position of the next tile's first chunk)
    const int c = c_next - 1;
    // ---- output transform (lane-local): Y = A^T M A, A^T = [[1,1,1,0],[0,1,-1,-1]]
    // The wave's 32 tiles x 32 channels (128 pixels) go through its private 16 KiB of LDS so that the global
    // stores are 16 bytes per lane (8 lanes per pixel): 16 store instructions per lane instead of 64 -- the
    // store tail of a one-block-per-CU kernel is issue-bound and nothing overlaps it.
    const TilePos tp = tile_pos(km);
    const int nb = tp.nb, b = tp.b, oy0 = tp.oy0, ox0 = tp.ox0;
    const int co = co_mine;
    float ssum = 0.f, ssq = 0.f;
    // scratch: V[f&1] (waves 0,1) and U[f&1] (waves 2,3) were consumed by this position's MFMAs; V/U[(f+1)&1] already
    // hold the next tile's first chunk and must survive
    float* ow = (wave < 2 ? Vs : Us) + (c & 1) * VU_BUF + (wave & 1) * 4096;   // [pixel = tile*4 + 2a + b][32 channels]
    constexpr bool has_stats = STATS;
    const bool full = oy0 + 16 <= a.H && ox0 + 16 <= a.W && nb * WN + WN <= a.Cout && (a.Cout & 3) == 0;   // uniform
  #pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int tloc = (r & 3) + 8 * (r >> 2) + 4 * lh;       // tile within the wave's 32
      float s0[4], s1[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float m0 = acc_elem(acc[0 + j][r]), m1 = acc_elem(acc[4 + j][r]), m2 = acc_elem(acc[8 + j][r]),
                    m3 = acc_elem(acc[12 + j][r]);
        s0[j] = m0 + m1 + m2;
        s1[j] = m1 - m2 - m3;
      }
      float y00 = s0[0] + s0[1] + s0[2] + bv, y01 = s0[1] - s0[2] - s0[3] + bv;
      float y10 = s1[0] + s1[1] + s1[2] + bv, y11 = s1[1] - s1[2] - s1[3] + bv;
      if constexpr (EPI == 2) {             // 0 <= slope <= 1 (checked on the host): lrelu(y) = max(y, slope y)
        y00 = fmaxf(y00, y00 * a.out_slope); y01 = fmaxf(y01, y01 * a.out_slope);
        y10 = fmaxf(y10, y10 * a.out_slope); y11 = fmaxf(y11, y11 * a.out_slope);
      }
      float* op = ow + tloc * 128 + li;
      op[0] = y00; op[32] = y01; op[64] = y10; op[96] = y11;
      if constexpr (has_stats) {
        if (full) {
          ssum += (y00 + y01) + (y10 + y11);
          ssq = __builtin_fmaf(y00, y00, __builtin_fmaf(y01, y01, __builtin_fmaf(y10, y10, __builtin_fmaf(y11, y11, ssq))));
        } else {
          const int tl = wm * 32 + tloc;
          const int oy = oy0 + 2 * (tl >> 3), ox = ox0 + 2 * (tl & 7);
          if (co < a.Cout && oy < a.H && ox < a.W) {            // statistics over the outputs that exist
            const bool vx = ox + 1 < a.W, vy = oy + 1 < a.H;
            ssum += y00; ssq += y00 * y00;
            if (vx) { ssum += y01; ssq += y01 * y01; }
            if (vy) { ssum += y10; ssq += y10 * y10; }
            if (vx && vy) { ssum += y11; ssq += y11 * y11; }
          }
        }
      }
      __builtin_amdgcn_sched_barrier(0);      // one accumulator row at a time: keeps the 256 accumulator reads from piling up
    }
    // the accumulators of the next tile: 16 MFMAs with zero operands (0 * 0 + 0) instead of 256 v_accvgpr_write; they
    // run in the matrix pipe while the stores below are issued
#pragma unroll
    for (int xi = 0; xi < 16; ++xi) acc[xi] = mfma_zero();
    __builtin_amdgcn_sched_barrier(0);
    {
      const int c4 = lane & 7;
      const int cbase = nb * WN + wn * 32 + c4 * 4;
      const bool vec_ok = (a.Cout & 3) == 0;
      if (full) {                             // whole block inside the image: one base pointer, no checks
        float* ybase = a.Y + (long long)share * a.yslab + (((long long)b * a.H + oy0) * a.W + ox0) * a.Cout + cbase;
  #pragma unroll
        for (int it = 0; it < 16; ++it) {
          const int p = it * 8 + (lane >> 3);
          const int tl = wm * 32 + (p >> 2), ab = p & 3;
          const int dy = 2 * (tl >> 3) + (ab >> 1), dx = 2 * (tl & 7) + (ab & 1);
          CY_NT_STORE(*(const f32x4*)(ow + p * 32 + c4 * 4), (f32x4*)(ybase + (dy * a.W + dx) * a.Cout));
        }
      } else {
  #pragma unroll
        for (int it = 0; it < 16; ++it) {
          const int p = it * 8 + (lane >> 3);                   // pixel of the wave: tile*4 + 2a + b
          const int tloc = p >> 2, ab = p & 3;
          const int tl = wm * 32 + tloc;
          const int oy = oy0 + 2 * (tl >> 3) + (ab >> 1), ox = ox0 + 2 * (tl & 7) + (ab & 1);
          const f32x4 v = *(const f32x4*)(ow + p * 32 + c4 * 4);
          if (oy < a.H && ox < a.W) {
            float* yp = a.Y + (long long)share * a.yslab + (((long long)b * a.H + oy) * a.W + ox) * a.Cout + cbase;
            if (vec_ok && cbase + 3 < a.Cout) *(f32x4*)yp = v;
            else {
  #pragma unroll
              for (int k = 0; k < 4; ++k) if (cbase + k < a.Cout) yp[k] = v[k];
            }
          }
        }
      }
    }
      if constexpr (STATS) {
      float* red = Rs + ((c + 1) & 1) * RAW_BUF;   // [2 wm][WN][2] in the raw buffer T(f+1) is done with
      ssum += __shfl_xor(ssum, 32, 64);
      ssq += __shfl_xor(ssq, 32, 64);
      if (lh == 0) {
        red[(wm * WN + wn * 32 + li) * 2 + 0] = ssum;
        red[(wm * WN + wn * 32 + li) * 2 + 1] = ssq;
      }
      __syncthreads();
      if (t < WN && nb * WN + t < a.Cout) {
        double* st = a.stats + (size_t)(blockIdx.x % CY_STATS_COPIES) * a.Cout * 2;
        atomicAdd(st + 2 * (nb * WN + t), (double)red[t * 2] + (double)red[(WN + t) * 2]);
        atomicAdd(st + 2 * (nb * WN + t) + 1, (double)red[t * 2 + 1] + (double)red[(WN + t) * 2 + 1]);
      }
    }

    __syncthreads();                        // scratch and `red` are rewritten by the next position's T / S_raw
  }
  }
}

// U[chunk][xi][kq][co (Np)][e] = (G g G^T)[xi] for ci = chunk*8 + kq*4 + e.
// transpose = 0: g = W[co][ci][.][.] (forward);  1: input gradient, g[kh][kw] = W[ci][co][2-kh][2-kw]
// One thread per (chunk, kq, co, e): the 9 weights are read once and all 16 positions written (one coalesced 256-byte
// segment per wave and position) -- a thread per OUTPUT element read every weight 16 times through the caches: 94 us for a
// 512 -> 1024 layer, 6 such launches per DarkNet step.
__global__ void wino_pack_kernel(const float* __restrict__ W, float* __restrict__ U, int Cout_l, int Cin_l, int transpose,
                                 int Cin, int Np, int Ncols, long long total) {
  const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;   // over total / 16
  if (idx * 16 >= total) return;
  const int e = (int)(idx & 3);
  long long r = idx >> 2;
  const int co = (int)(r % Np); r /= Np;
  const int kq = (int)(r & 1); r >>= 1;
  const int chunk = (int)r;
  const int ci = chunk * 8 + kq * 4 + e;
  float g[3][3];
  const bool ok = ci < Cin && co < Ncols;
#pragma unroll
  for (int kh = 0; kh < 3; ++kh)
#pragma unroll
    for (int kw = 0; kw < 3; ++kw)
      g[kh][kw] = !ok ? 0.f : transpose ? W[(((long long)ci * Cin_l + co) * 3 + (2 - kh)) * 3 + (2 - kw)]
                                        : W[(((long long)co * Cin_l + ci) * 3 + kh) * 3 + kw];
  (void)Cout_l;
  const float Gm[4][3] = {{1.f, 0.f, 0.f}, {0.5f, 0.5f, 0.5f}, {0.5f, -0.5f, 0.5f}, {0.f, 0.f, 1.f}};
  float* out = U + (((long long)chunk * 16 * 2 + kq) * Np + co) * 4 + e;
#pragma unroll
  for (int xi = 0; xi < 16; ++xi) {
    const int i = xi >> 2, j = xi & 3;
    float u = 0.f;
#pragma unroll
    for (int p = 0; p < 3; ++p)
#pragma unroll
      for (int q = 0; q < 3; ++q) u += Gm[i][p] * g[p][q] * Gm[j][q];
    out[(long long)xi * 2 * Np * 4] = u;
  }
}

// ================================================================================================ weight gradient
// dW (3x3) through Winograd F(3x3, 2x2): per 2x2-output tile, dW_tile = A^T [ (G dY G^T) (.) (B^T d B) ] A with
//   B^T = [[1,0,-1,0],[0,1,1,0],[0,-1,1,0],[0,-1,0,1]],  G = [[1,0],[.5,.5],[.5,-.5],[0,1]],  A^T = [[1,1,1,0],[0,1,-1,0],[0,1,1,1]]
// The sum over tiles and images commutes with the output transform, so the heavy part is 16 GEMMs
//   dU_xi[ci][co] = sum_tiles V_xi[tile][ci] * Z_xi[tile][co]        (2.25x fewer multiplies than the direct form)
// with the reduction (tiles) as the MFMA k dimension.  One block = 64 ci x 64 co of ONE image (the per-image
// partial sums are added by the finish kernel, which also applies A^T . A); 4 waves (2 x 2), one per SIMD, 16
// accumulator tiles each, same slot-scheduled pipeline as the forward kernel.  A chunk is 8 tiles (2 x 4):
// its 6x10 input patch and 4x8 dY patch are staged raw in LDS (256-byte rows, fully coalesced), every thread
// transforms two tiles of one channel of each operand into the [xi][k/4][channel][4 tiles] images.
//
// fp32 MFMA and fp32 VALU share the SIMD's ALUs on this chip (tools/probe/mfma_overlap.hip: every VALU
// instruction of the same wave OR of a second wave on the SIMD adds ~3-10 cycles to the 64-cycle MFMA), so the
// loop is written for a minimal VALU count, not only for a balanced slot schedule:
//  * the two tiles a thread transforms are carried as float2 and every add is one v_pk_add_f32; the pairs come
//    straight out of LDS (ds_read2st64_b32 of columns c and c+2), and go back as one ds_write_b64;
//  * G's 0.5 factors are dropped (rows dY0, dY0+dY1, dY0-dY1, dY1) and restored exactly, as powers of two, by the
//    finish kernel;
//  * chunks whose patches are completely inside the image (94 % at 416x416) load through a uniform base + a
//    per-thread constant offset: no address arithmetic, no range checks, no select before the LDS store.
constexpr int GT = 8;                       // tiles per chunk (2 rows x 4 columns of tiles)
constexpr int XP = 60, ZP = 32;             // raw patch pixels: 6x10 input, 4x8 dY
constexpr int XPS = 64;                     // input patch rows allocated (4 spare: every thread stores 4 float4)
constexpr int RAWW_BUF = (XPS + ZP) * 64;   // floats: [pixel][64 channels] for both operands (single buffer)

struct WinoWgradArgs {
  const float* X; const float* dZ; float* slab;
  int B, H, W, Cin, Cout, gh, gw;           // gh x gw tile groups per image
  int nrange;                               // the B * gh * gw tile groups are cut into nrange contiguous ranges, one block (and one
                                            // partial-sum slab) per range and (ci, co) tile: ~one block per CU for any batch / map size
  // BNF variant (BatchNorm + LeakyReLU backward applied on the way in): dZ above is then dA, the gradient with respect to the
  // ACTIVATED output, and the kernel forms dz = scale * (d - mean(d) - xhat * mean(d * xhat)), d = dA * lrelu'(z * scale + shift),
  // from the raw convolution output Z while the patch is in registers; the blocks of the first two input-channel blocks also write
  // dz to dZout (one of the thread's two float4 each) for the input-gradient kernel
  const float* Z; float* dZout; float* dummy;               // dummy: 4 KiB that absorb the stores of blocks / chunks that must not write
  const float *scale, *shift, *mean, *invstd;
  const double* red;                        // [Cout][2] sums of d and d * xhat over all pixels
  double inv_count; float slope;
  int premasked;                            // dA is already d = dA * lrelu'(y) (the stride-2 input-gradient kernel's fused sums)
};

// Side work of one chunk, one piece per MFMA slot:
//   2 T    raw-patch reads of chunk c+1, two ds_read2st64_b32 per piece: dY first, then input rows 1,2,0,3   (10 pieces)
//   5 Z    half a row of G dY G^T for both tiles (at most 2 LDS stores per slot)                             (8 pieces)
//   4 V    half a row of B^T d B for both tiles, rows in the order 1,2,0,3                                    (8 pieces)
//   6      barrier: every wave is past its raw-patch reads
//   7 S    one float4 of chunk c+2 registers -> raw LDS                                                       (6 pieces)
//   1 G    one global load of chunk c+3 (almost a whole chunk of latency cover, one register set)             (6 pieces)
constexpr int WG_T[10] = {2, 3, 4, 5, 6, 7, 10, 11, 12, 13};
constexpr int WG_Z[8] = {14, 15, 18, 19, 20, 21, 22, 23};          // half rows: at most 2 LDS stores per slot (winograd forward)
constexpr int WG_V[8] = {26, 27, 28, 29, 30, 31, 34, 35};
constexpr int WG_BAR = 36;
constexpr int WG_S[6] = {37, 38, 39, 42, 43, 44};
constexpr int WG_G[6] = {45, 46, 47, 50, 51, 52};
// BNF variant only (slots without side work otherwise):
//   8 A    dz of one float4 of chunk c+2 from (dA, Z) in registers, in front of its S slot (43 / 44)       (2 pieces)
//   9 W    its global store, before the G slot (51 / 52) that overwrites the registers                       (2 pieces)
//  10 GZ   one global load of Z of chunk c+3                                                                 (2 pieces)
constexpr int WG_A[2] = {40, 41};
constexpr int WG_W[2] = {48, 49};
constexpr int WG_GZ[2] = {53, 55};
constexpr int wg_side_kind(int s) {
  return wino_find(WG_T, 10, s) >= 0 ? 2 : wino_find(WG_Z, 8, s) >= 0 ? 5 : wino_find(WG_V, 8, s) >= 0 ? 4 : s == WG_BAR ? 6
       : wino_find(WG_S, 6, s) >= 0 ? 7 : wino_find(WG_G, 6, s) >= 0 ? 1 : wino_find(WG_A, 2, s) >= 0 ? 8
       : wino_find(WG_W, 2, s) >= 0 ? 9 : wino_find(WG_GZ, 2, s) >= 0 ? 10 : 0;
}
constexpr int wg_side_idx(int s) {
  const int k = wg_side_kind(s);
  return k == 2 ? wino_find(WG_T, 10, s) : k == 5 ? wino_find(WG_Z, 8, s) : k == 4 ? wino_find(WG_V, 8, s)
       : k == 7 ? wino_find(WG_S, 6, s) : k == 1 ? wino_find(WG_G, 6, s) : k == 8 ? wino_find(WG_A, 2, s)
       : k == 9 ? wino_find(WG_W, 2, s) : k == 10 ? wino_find(WG_GZ, 2, s) : 0;
}
constexpr int wg_row_order(int i) { return i == 0 ? 1 : i == 1 ? 2 : i == 2 ? 0 : 3; }
constexpr int wg_side_lds(int s) {          // LOWER bound of the LDS instructions issued by the slot
  const int k = wg_side_kind(s);
  return k == 2 ? 2 : k == 4 ? 2 : k == 5 ? 2 : k == 7 ? 1 : 0;
}
constexpr int wg_younger(int xi) {
  const int is = wino_issue(xi), us = wino_use(xi);
  int n = 0;
  if (is < 0) {
    if (xi == 0) n += 2;
    for (int s = 0; s < us; ++s) n += wino_frag_lds(s) + wg_side_lds(s);
  } else {
    n += wg_side_lds(is);
    for (int s = is + 1; s < us; ++s) n += wino_frag_lds(s) + wg_side_lds(s);
  }
  return n > 14 ? 14 : n;
}

// (dword[O0 * 64], dword[O1 * 64]) from LDS byte address `addr` as ONE aligned register pair.  Plain C++ loads
// are paired up by the compiler as it likes (adjacent columns) and then shuffled with v_mov + an immediate wait;
// the transform below needs (column c, column c + 2).  The compiler does not count this read: the consumer slot
// waits with an explicit s_waitcnt.
template <int O0, int O1>
__device__ __forceinline__ f32x2 lds_pair_st64(unsigned addr) {
  f32x2 r;
  asm volatile("ds_read2st64_b32 %0, %1 offset0:%2 offset1:%3" : "=v"(r) : "v"(addr), "n"(O0), "n"(O1));
  return r;
}

// BNF: 0 = plain, 1 = BatchNorm + LeakyReLU backward pass 2 on the way in, 2 = the same for a premasked gradient (no y, no select)
template <int BNF>
__global__ __launch_bounds__(256, 1) void wino_wgrad_kernel(WinoWgradArgs a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* Vs = smem;                         // [2][VU_BUF]   A images (input, rows = ci)
  float* Zs = smem + 2 * VU_BUF;            // [2][VU_BUF]   B images (dY, rows = co)
  float* Rw = smem + 4 * VU_BUF;            // [XPS + ZP][64] raw patches

  const int t = threadIdx.x, lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int wm = wave & 1, wn = wave >> 1;
  const int li = lane & 31, lh = lane >> 5;
  const int ncb = a.Cout / 64, nib = a.Cin / 64;
  // consecutive block ids go round-robin over the 8 XCDs: give each XCD a contiguous range of virtual ids, so that
  // the nib*ncb blocks that share one image's patches also share one L2
  int vid = blockIdx.x;
  if ((gridDim.x & 7) == 0) vid = (blockIdx.x & 7) * (gridDim.x >> 3) + (blockIdx.x >> 3);
  const int cob = vid % ncb;
  const int cib = (vid / ncb) % nib;
  const int jr = vid / (ncb * nib);         // range of tile groups
  const int ngroups = a.gh * a.gw;
  const long long gtot = (long long)a.B * ngroups;
  const int cbeg = (int)(gtot * jr / a.nrange), cend = (int)(gtot * (jr + 1) / a.nrange);
  const int nchunk = cend - cbeg;           // >= 1 (nrange <= B * ngroups)

  // ---- loader: raw items (pixel = prow + 16 q, float4 column c4): 4 input + 2 dY float4 per thread
  const int c4 = t & 15, prow = t >> 4;
  const char* xall = (const char*)(a.X + cib * 64);          // + image * ximgb (uniform)
  const char* zall = (const char*)(a.dZ + cob * 64);
  const size_t ximgb = (size_t)a.H * a.W * a.Cin * 4, zimgb = (size_t)a.H * a.W * a.Cout * 4;
  int xpy[4], xpx[4];
  unsigned voffx[4], voffz[2];              // byte offsets of this thread's items from the patch origin
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int pix = prow + 16 * q;
    xpy[q] = pix / 10; xpx[q] = pix - xpy[q] * 10;
    voffx[q] = pix < XP ? (unsigned)(((xpy[q] * a.W + xpx[q]) * a.Cin + c4 * 4) * 4) : (unsigned)(c4 * 16);
  }
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    const int pix = prow + 16 * q;
    voffz[q] = (unsigned)((((pix >> 3) * a.W + (pix & 7)) * a.Cout + c4 * 4) * 4);
  }
  f32x4 gx[4], gz[2];
  f32x4 zz[2], gd[2];                       // BNF: Z next to dA in gz; dz (kept apart: after the stream of a border chunk gz / zz
                                            // already hold the next chunk, gd is still the one to store with bounds)
  // BNF: per-channel constants of this thread's 4 channels (its float4 column is the same in every chunk):
  //   dz = dA * (y > 0 ? sc : sc * slope) + (z - mu) * kb + kc,  y = z * sc + sh,  kb = -sc * invstd * mean(d xhat),  kc = -sc * mean(d)
  f32x2 k_sc[2], k_sh[2], k_scs[2], k_nmu[2], k_b[2], k_c[2];
  unsigned zoff_fast[2] = {0u, 0u};         // lane offset of the in-stream store of float4 q (inside-the-image chunks)
  const char* zrall = nullptr;              // Z and dZout at this block's output channels
  char* zoall = nullptr;
  bool stq[2] = {false, false};             // this block writes dz of its float4 q
  if constexpr (BNF) {
    zrall = (const char*)(a.Z + cob * 64);
    zoall = (char*)(a.dZout + cob * 64);
    stq[0] = nib == 1 || cib == 0;
    stq[1] = nib == 1 || cib == 1;
#pragma unroll
    for (int q = 0; q < 2; ++q) zoff_fast[q] = stq[q] ? voffz[q] : (unsigned)(t * 16);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int ch = cob * 64 + c4 * 4 + k;
      const float sc = a.scale[ch], is = a.invstd[ch];
      const float m1 = (float)(a.red[2 * ch] * a.inv_count), m2 = (float)(a.red[2 * ch + 1] * a.inv_count);
      k_sc[k >> 1][k & 1] = sc; k_sh[k >> 1][k & 1] = a.shift[ch]; k_scs[k >> 1][k & 1] = sc * a.slope;
      k_nmu[k >> 1][k & 1] = -a.mean[ch]; k_b[k >> 1][k & 1] = -sc * is * m2; k_c[k >> 1][k & 1] = -sc * m1;
    }
  }
  auto bn_apply = [&](int q) {              // (gz[q], zz[q]) = (dA, Z) -> gd[q] = dz: 8 packed + 8 plain VALU instructions
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const f32x2 z = f32x2{zz[q][2 * h], zz[q][2 * h + 1]}, g = f32x2{gz[q][2 * h], gz[q][2 * h + 1]};
      f32x2 sel = k_sc[h];
      if constexpr (BNF == 1) {
        const f32x2 y = pk_fma(z, k_sc[h], k_sh[h]);
        sel = f32x2{y[0] > 0.f ? k_sc[h][0] : k_scs[h][0], y[1] > 0.f ? k_sc[h][1] : k_scs[h][1]};
      }
      const f32x2 o = pk_fma(g, sel, pk_fma(pk_add(z, k_nmu[h]), k_b[h], k_c[h]));
      gd[q][2 * h] = o[0]; gd[q][2 * h + 1] = o[1];
    }
  };
  unsigned okm = 0;                         // slow path only: bit q: gx[q] in range, bit 4+q: gz[q]
  bool gfast = false;                       // the chunk held in gx/gz was loaded by the fast path
  // chunk (gy, gxx): fast when the 6x10 input patch and the 4x8 dY patch are inside the image
  auto is_fast = [&](int gy, int gxx) {
    return gy > 0 && gxx > 0 && gy * 4 + 5 <= a.H && gxx * 8 + 9 <= a.W;
  };
  auto Gx = [&](int q, int gb, int gy, int gxx, bool fast) {
    const char* ximg = xall + gb * ximgb;
    const int iy0 = gy * 4 - 1, ix0 = gxx * 8 - 1;
    if (fast) {
      gx[q] = *(const f32x4*)(ximg + (size_t)((iy0 * a.W + ix0) * a.Cin) * 4 + voffx[q]);
    } else {
      const int iy = iy0 + xpy[q], ix = ix0 + xpx[q];
      const bool ok = (prow + 16 * q) < XP && (unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W;
      gx[q] = *(const f32x4*)(ximg + (ok ? (unsigned)(((iy * a.W + ix) * a.Cin + c4 * 4) * 4) : 0u));
      okm = (okm & ~(1u << q)) | ((unsigned)ok << q);
    }
  };
  auto Gz = [&](int q, int gb, int gy, int gxx, bool fast) {
    const char* zimg = zall + gb * zimgb;
    if (fast) {
      const size_t off = (size_t)((gy * 4 * a.W + gxx * 8) * a.Cout) * 4 + voffz[q];
      gz[q] = *(const f32x4*)(zimg + off);
      if constexpr (BNF) zz[q] = *(const f32x4*)(zrall + gb * zimgb + off);
    } else {
      const int pix = prow + 16 * q;
      const int oy = gy * 4 + (pix >> 3), ox = gxx * 8 + (pix & 7);
      const bool ok = oy < a.H && ox < a.W;
      const unsigned off = ok ? (unsigned)(((oy * a.W + ox) * a.Cout + c4 * 4) * 4) : 0u;
      gz[q] = *(const f32x4*)(zimg + off);
      if constexpr (BNF) zz[q] = *(const f32x4*)(zrall + gb * zimgb + off);
      okm = (okm & ~(16u << q)) | ((unsigned)ok << (4 + q));
    }
  };
  // BNF, outside the chunk stream (prologue chunks 0 and 1, border chunks): gd[q] to dZout with bounds
  auto Wz = [&](int q, int gb, int gy, int gxx) {
    const int pix = prow + 16 * q;
    const int oy = gy * 4 + (pix >> 3), ox = gxx * 8 + (pix & 7);
    if (stq[q] && oy < a.H && ox < a.W)
      *(f32x4*)(zoall + gb * zimgb + (size_t)(((oy * a.W + ox) * a.Cout + c4 * 4) * 4)) = gd[q];
  };
  auto Sx = [&](int q) {
    float* dst = Rw + (prow + 16 * q) * 64 + c4 * 4;
    if (gfast) *(f32x4*)dst = gx[q];
    else *(f32x4*)dst = (okm >> q) & 1 ? gx[q] : f32x4{0.f, 0.f, 0.f, 0.f};
  };
  auto Sz = [&](int q) {
    if constexpr (BNF) bn_apply(q);
    const f32x4 v = BNF ? gd[q] : gz[q];
    float* dst = Rw + (XPS + prow + 16 * q) * 64 + c4 * 4;
    if (gfast) *(f32x4*)dst = v;
    else *(f32x4*)dst = (okm >> (4 + q)) & 1 ? v : f32x4{0.f, 0.f, 0.f, 0.f};
  };
  int lgb = 0, lgy = 0, lgx = 0;            // position of the chunk Gall loaded last
  auto Gall = [&](int c) {
    const int gb = (cbeg + c) / ngroups, gr = (cbeg + c) - gb * ngroups;
    const int gy = gr / a.gw, gxx = gr - gy * a.gw;
    lgb = gb; lgy = gy; lgx = gxx;
    const bool fast = is_fast(gy, gxx);
#pragma unroll
    for (int q = 0; q < 4; ++q) Gx(q, gb, gy, gxx, fast);
#pragma unroll
    for (int q = 0; q < 2; ++q) Gz(q, gb, gy, gxx, fast);
    gfast = fast;
  };
  auto Sall = [&]() {
#pragma unroll
    for (int q = 0; q < 4; ++q) Sx(q);
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      Sz(q);
      if constexpr (BNF) Wz(q, lgb, lgy, lgx);
    }
  };

  // ---- transform item: channel tc = t & 63, tile row tr, tile-column pair tp; .x = tile 2 tp, .y = tile 2 tp + 1
  const int tc = t & 63, tr = t >> 7, tp = (t >> 6) & 1;
  const float* xr = Rw + ((2 * tr) * 10 + 4 * tp) * 64 + tc;            // 4 rows x 6 cols of the input patch
  const float* zr = Rw + XPS * 64 + ((2 * tr) * 8 + 4 * tp) * 64 + tc;  // 2 rows x 4 cols of the dY patch
  const int vdst = tr * SLAB + tc * 4 + 2 * tp;                        // + xi * 2 * SLAB
  const unsigned xr_a = (unsigned)(uintptr_t)(__attribute__((address_space(3))) const float*)xr;
  const unsigned zr_a = (unsigned)(uintptr_t)(__attribute__((address_space(3))) const float*)zr;
  f32x2 xv[4][4], zv[2][2];                 // [row][col] pairs: (col, col + 2) = the same column of the two tiles
  auto Tx = [&](int r, int c) { xv[r][c] = f32x2{xr[(r * 10 + c) * 64], xr[(r * 10 + c + 2) * 64]}; };
  auto Tz = [&](int r, int c) { zv[r][c] = f32x2{zr[(r * 8 + c) * 64], zr[(r * 8 + c + 2) * 64]}; };
  auto Vrow = [&](float* vb, int R) {       // row R of B^T d B
    f32x2 t0[4];
#pragma unroll
    for (int c = 0; c < 4; ++c)
      t0[c] = R == 0 ? pk_sub(xv[0][c], xv[2][c]) : R == 1 ? pk_add(xv[1][c], xv[2][c])
            : R == 2 ? pk_sub(xv[2][c], xv[1][c]) : pk_sub(xv[3][c], xv[1][c]);
    *(f32x2*)(vb + (R * 4 + 0) * 2 * SLAB) = pk_sub(t0[0], t0[2]);
    *(f32x2*)(vb + (R * 4 + 1) * 2 * SLAB) = pk_add(t0[1], t0[2]);
    *(f32x2*)(vb + (R * 4 + 2) * 2 * SLAB) = pk_sub(t0[2], t0[1]);
    *(f32x2*)(vb + (R * 4 + 3) * 2 * SLAB) = pk_sub(t0[3], t0[1]);
  };
  auto Vhalf = [&](float* vb, int R, int jp) {   // columns 2 jp, 2 jp + 1 of row R
    f32x2 t0[4];
#pragma unroll
    for (int c = 0; c < 4; ++c)
      t0[c] = R == 0 ? pk_sub(xv[0][c], xv[2][c]) : R == 1 ? pk_add(xv[1][c], xv[2][c])
            : R == 2 ? pk_sub(xv[2][c], xv[1][c]) : pk_sub(xv[3][c], xv[1][c]);
    if (jp == 0) {
      *(f32x2*)(vb + (R * 4 + 0) * 2 * SLAB) = pk_sub(t0[0], t0[2]);
      *(f32x2*)(vb + (R * 4 + 1) * 2 * SLAB) = pk_add(t0[1], t0[2]);
    } else {
      *(f32x2*)(vb + (R * 4 + 2) * 2 * SLAB) = pk_sub(t0[2], t0[1]);
      *(f32x2*)(vb + (R * 4 + 3) * 2 * SLAB) = pk_sub(t0[3], t0[1]);
    }
  };
  auto Zhalf = [&](float* zb, int R, int jp) {
    f32x2 t0[2];
#pragma unroll
    for (int c = 0; c < 2; ++c)
      t0[c] = R == 0 ? zv[0][c] : R == 1 ? pk_add(zv[0][c], zv[1][c]) : R == 2 ? pk_sub(zv[0][c], zv[1][c]) : zv[1][c];
    if (jp == 0) {
      *(f32x2*)(zb + (R * 4 + 0) * 2 * SLAB) = t0[0];
      *(f32x2*)(zb + (R * 4 + 1) * 2 * SLAB) = pk_add(t0[0], t0[1]);
    } else {
      *(f32x2*)(zb + (R * 4 + 2) * 2 * SLAB) = pk_sub(t0[0], t0[1]);
      *(f32x2*)(zb + (R * 4 + 3) * 2 * SLAB) = t0[1];
    }
  };
  auto Zrow = [&](float* zb, int R) {       // row R of (2G) dY (2G)^T
    f32x2 t0[2];
#pragma unroll
    for (int c = 0; c < 2; ++c)
      t0[c] = R == 0 ? zv[0][c] : R == 1 ? pk_add(zv[0][c], zv[1][c]) : R == 2 ? pk_sub(zv[0][c], zv[1][c]) : zv[1][c];
    *(f32x2*)(zb + (R * 4 + 0) * 2 * SLAB) = t0[0];
    *(f32x2*)(zb + (R * 4 + 1) * 2 * SLAB) = pk_add(t0[0], t0[1]);
    *(f32x2*)(zb + (R * 4 + 2) * 2 * SLAB) = pk_sub(t0[0], t0[1]);
    *(f32x2*)(zb + (R * 4 + 3) * 2 * SLAB) = t0[1];
  };

  f32x16 acc[16];
#pragma unroll
  for (int xi = 0; xi < 16; ++xi)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[xi][r] = 0.f;

  // ---- prologue: chunk 0 transformed into buffer 0, chunk 1 raw in LDS, chunk 2 in registers
  Gall(0);
  Sall();
  __syncthreads();
#pragma unroll
  for (int r = 0; r < 4; ++r)
#pragma unroll
    for (int c = 0; c < 4; ++c) Tx(r, c);
#pragma unroll
  for (int r = 0; r < 2; ++r)
#pragma unroll
    for (int c = 0; c < 2; ++c) Tz(r, c);
#pragma unroll
  for (int R = 0; R < 4; ++R) { Vrow(Vs + vdst, R); Zrow(Zs + vdst, R); }
  Gall(nchunk > 1 ? 1 : 0);
  __syncthreads();
  Sall();
  Gall(nchunk > 2 ? 2 : nchunk - 1);
  __syncthreads();

  const int fragA = lh * SLAB + (wm * 32 + li) * 4;
  const int fragB = lh * SLAB + (wn * 32 + li) * 4;
  // position (image, group row, group column) of the chunk whose loads are in flight: chunk min(c + 3, nchunk - 1),
  // stepped without divisions
  int ggb, ggy, ggx;
  {
    const int cg = cbeg + (nchunk > 2 ? 2 : nchunk - 1);
    ggb = cg / ngroups;
    const int ggr = cg - ggb * ngroups;
    ggy = ggr / a.gw; ggx = ggr - ggy * a.gw;
  }
  // as in the forward kernel the end-of-chunk barrier stands at slot WS_BAR (all LDS writes <= slot 44, all fragment
  // fetches <= slot 49) and the next chunk's first two fragment sets are fetched under the last MFMAs
  f32x4 fa_[4], fb_[4];
  fa_[0] = *(const f32x4*)(Vs + fragA);
  fb_[0] = *(const f32x4*)(Zs + fragB);
  fa_[1] = *(const f32x4*)(Vs + fragA + 2 * SLAB);
  fb_[1] = *(const f32x4*)(Zs + fragB + 2 * SLAB);
  for (int c = 0; c < nchunk; ++c) {
    const float* vb_ = Vs + (c & 1) * VU_BUF + fragA;
    const float* ub_ = Zs + (c & 1) * VU_BUF + fragB;
    float* vw_ = Vs + ((c + 1) & 1) * VU_BUF + vdst;                // T(c+1) (harmless after the last chunk)
    float* zw_ = Zs + ((c + 1) & 1) * VU_BUF + vdst;
    // BNF: the chunk in gz (c + 2, or the last one again) is stored by this stream: inside the image through a uniform base +
    // the lane's patch offset; float4s this block must not write and border chunks go to the dummy block instead (no branch,
    // no exec mask in the slots), a border chunk is stored with bounds after the stream
    const int pgb = ggb, pgy = ggy, pgx = ggx;
    char* zo_[2] = {nullptr, nullptr};
    unsigned zoo_[2] = {0u, 0u};
    if constexpr (BNF) {
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        zo_[q] = (gfast && stq[q]) ? zoall + pgb * zimgb + (size_t)((pgy * 4 * a.W + pgx * 8) * a.Cout) * 4 : (char*)a.dummy;
        zoo_[q] = zoff_fast[q] & (gfast ? ~0u : 0xFF0u);
      }
    }
    if (c + 3 < nchunk) {                                           // uniform; the tail re-loads the last chunk
      const bool wx = ggx + 1 == a.gw;
      ggx = wx ? 0 : ggx + 1;
      const bool wy = wx && ggy + 1 == a.gh;
      ggy = wy ? 0 : ggy + (wx ? 1 : 0);
      ggb += wy ? 1 : 0;
    }
    // The MFMA slots hold no branch: a chunk inside the image loads `uniform patch origin + lane offset`; a chunk at
    // the border loads a harmless valid address there and is re-loaded, with bounds, after the stream; LDS stores are
    // unconditional and the padding of a border chunk is zeroed in LDS after the stream.
    const bool gf_next = is_fast(ggy, ggx);
    const char* xb_ = xall + (gf_next ? ggb * ximgb + (size_t)(((ggy * 4 - 1) * a.W + ggx * 8 - 1) * a.Cin) * 4 : (size_t)0);
    const size_t zoff_ = gf_next ? ggb * zimgb + (size_t)((ggy * 4 * a.W + ggx * 8) * a.Cout) * 4 : (size_t)0;
    const char* zb_ = zall + zoff_;
    const char* zrb_ = zrall + zoff_;
    const float* vn_ = Vs + ((c + 1) & 1) * VU_BUF + fragA;         // fragments of chunk c+1
    const float* un_ = Zs + ((c + 1) & 1) * VU_BUF + fragB;
#define WGSLOT(SIDX)                                                                                \
    {                                                                                               \
      constexpr int sidx = (SIDX);                                                                  \
      constexpr int xi = wino_xi(sidx), e = wino_e(sidx);                                           \
      if (e == 0) __builtin_amdgcn_s_waitcnt(0xC07F | (wg_younger(xi) << 8));                       \
      acc[xi] = mfma32(fa_[xi & 3][e], fb_[xi & 3][e], acc[xi]);                                    \
      constexpr int fp = wino_frag_pos(sidx);                                                       \
      if (fp >= 0) {                                                                                \
        constexpr int fq = fp >= 0 ? fp : 0;                                                        \
        fa_[fq & 3] = *(const f32x4*)(vb_ + fq * 2 * SLAB);                                         \
        fb_[fq & 3] = *(const f32x4*)(ub_ + fq * 2 * SLAB);                                         \
      }                                                                                             \
      constexpr int kind = wg_side_kind(sidx), k_ = wg_side_idx(sidx) >= 0 ? wg_side_idx(sidx) : 0; \
      if (kind == 1) {                      /* one global load of chunk c+3 */                     \
        if (k_ < 4) gx[k_ & 3] = *(const f32x4*)(xb_ + (gf_next ? voffx[k_ & 3] : (unsigned)(c4 * 16)));     \
        else gz[k_ & 1] = *(const f32x4*)(zb_ + (gf_next ? voffz[k_ & 1] : (unsigned)(c4 * 16)));   \
      } else if (kind == 2) {               /* raw patches of chunk c+1: 2 pairs */                \
        if (k_ < 2) {                                                                               \
          constexpr int r = k_ & 1;                                                                 \
          zv[r][0] = lds_pair_st64<r * 8 + 0, r * 8 + 2>(zr_a);                                     \
          zv[r][1] = lds_pair_st64<r * 8 + 1, r * 8 + 3>(zr_a);                                     \
        } else {                                                                                    \
          constexpr int L = (k_ - 2) & 7, r = wg_row_order(L >> 1), c0 = 2 * (L & 1);               \
          xv[r][c0] = lds_pair_st64<r * 10 + c0, r * 10 + c0 + 2>(xr_a);                            \
          xv[r][c0 + 1] = lds_pair_st64<r * 10 + c0 + 1, r * 10 + c0 + 3>(xr_a);                    \
        }                                                                                           \
      } else if (kind == 4) {               /* all patch reads are >= 14 LDS operations old */     \
        if (k_ == 0) __builtin_amdgcn_s_waitcnt(0xC07F | (14 << 8));                                \
        Vhalf(vw_, wg_row_order((k_ >> 1) & 3), k_ & 1);                                            \
      } else if (kind == 5) {               /* the dY reads are >= 14 LDS operations old */        \
        if (k_ == 0) __builtin_amdgcn_s_waitcnt(0xC07F | (14 << 8));                                \
        Zhalf(zw_, (k_ >> 1) & 3, k_ & 1);                                                          \
      } else if (kind == 6) {               /* all waves are past their raw-patch reads */         \
        __builtin_amdgcn_s_barrier();                                                               \
      } else if (kind == 8) {               /* BNF: dA -> dz of one float4 of chunk c+2 */         \
        if constexpr (BNF) bn_apply(k_ & 1);                                                        \
      } else if (kind == 9) {               /* BNF: dz of chunk c+2 -> dZout */                    \
        if constexpr (BNF) CY_NT_STORE(gd[k_ & 1], (f32x4*)(zo_[k_ & 1] + zoo_[k_ & 1]));          \
      } else if (kind == 10) {              /* BNF: Z of chunk c+3 */                              \
        if constexpr (BNF) zz[k_ & 1] = *(const f32x4*)(zrb_ + (gf_next ? voffz[k_ & 1] : (unsigned)(c4 * 16))); \
      } else if (kind == 7) {               /* chunk c+2: one float4 of registers -> raw LDS */    \
        if (k_ < 4) *(f32x4*)(Rw + (prow + 16 * (k_ & 3)) * 64 + c4 * 4) = gx[k_ & 3];              \
        else *(f32x4*)(Rw + (XPS + prow + 16 * (k_ & 1)) * 64 + c4 * 4) = BNF ? gd[k_ & 1] : gz[k_ & 1];  \
      } else if (sidx == WS_BAR) {          /* end-of-chunk barrier */                             \
        __builtin_amdgcn_s_waitcnt(0xC07F);                                                         \
        __builtin_amdgcn_s_barrier();                                                               \
      } else if (sidx == WS_BAR + 2) {      /* positions 12, 13 are done with sets 0 and 1 */      \
        fa_[0] = *(const f32x4*)(vn_);                                                              \
        fb_[0] = *(const f32x4*)(un_);                                                              \
      } else if (sidx == WS_BAR + 3) {                                                              \
        fa_[1] = *(const f32x4*)(vn_ + 2 * SLAB);                                                   \
        fb_[1] = *(const f32x4*)(un_ + 2 * SLAB);                                                   \
      }                                                                                             \
      __builtin_amdgcn_sched_barrier(0);                                                            \
    }
#define WGSLOT4(B) WGSLOT((B)) WGSLOT((B) + 1) WGSLOT((B) + 2) WGSLOT((B) + 3)
#define WGSLOT16(B) WGSLOT4((B)) WGSLOT4((B) + 4) WGSLOT4((B) + 8) WGSLOT4((B) + 12)
    WGSLOT16(0) WGSLOT16(16) WGSLOT16(32) WGSLOT16(48)
#undef WGSLOT16
#undef WGSLOT4
#undef WGSLOT
    if (!(gfast && gf_next)) {              // uniform, border chunks only
      if (!gfast) {                         // the patch just stored: zero its padding (okm of its load)
#pragma unroll
        for (int q = 0; q < 4; ++q)
          if (!((okm >> q) & 1)) *(f32x4*)(Rw + (prow + 16 * q) * 64 + c4 * 4) = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int q = 0; q < 2; ++q) {
          if (!((okm >> (4 + q)) & 1)) *(f32x4*)(Rw + (XPS + prow + 16 * q) * 64 + c4 * 4) = f32x4{0.f, 0.f, 0.f, 0.f};
          if constexpr (BNF) Wz(q, pgb, pgy, pgx);
        }
      }
      if (!gf_next) {                       // the patch just requested: load it again with bounds
#pragma unroll
        for (int q = 0; q < 4; ++q) Gx(q, ggb, ggy, ggx, false);
#pragma unroll
        for (int q = 0; q < 2; ++q) Gz(q, ggb, ggy, ggx, false);
      }
      __syncthreads();                      // the zeroed padding must be in LDS before any wave transforms the patch
    }
    gfast = gf_next;
  }

  // ---- per-image partial dU[xi][ci][co] -> slab[b]
  float* out = a.slab + ((long long)jr * 16) * a.Cin * a.Cout;
  const int co = cob * 64 + wn * 32 + li;
#pragma unroll
  for (int xi = 0; xi < 16; ++xi)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int ci = cib * 64 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
      out[((long long)xi * a.Cin + ci) * a.Cout + co] = acc[xi][r];
    }
}

// dW[co][ci][p][q] = sum_{i,j} AT[p][i] AT[q][j] g_i g_j * sum_b slab[b][i*4+j][ci][co],  g = (1, .5, .5, 1) restores G's
// factors that the main kernel leaves out
__global__ void wino_wgrad_finish_kernel(const float* __restrict__ slab, float* __restrict__ dW, int nb, int Cin, int Cout) {
  const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (long long)Cin * Cout) return;
  const int co = (int)(idx % Cout), ci = (int)(idx / Cout);
  float m[16];
#pragma unroll
  for (int xi = 0; xi < 16; ++xi) m[xi] = 0.f;
  for (int bb = 0; bb < nb; ++bb)
#pragma unroll
    for (int xi = 0; xi < 16; ++xi) m[xi] += slab[(((long long)bb * 16 + xi) * Cin + ci) * Cout + co];
#pragma unroll
  for (int xi = 0; xi < 16; ++xi) {
    const int i = xi >> 2, j = xi & 3;
    m[xi] *= ((i == 1 || i == 2) ? 0.5f : 1.f) * ((j == 1 || j == 2) ? 0.5f : 1.f);
  }
  // rows: s[p][j] = sum_i AT[p][i] m[i][j], AT = [[1,1,1,0],[0,1,-1,0],[0,1,1,1]]
  float sr[3][4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    sr[0][j] = m[0 + j] + m[4 + j] + m[8 + j];
    sr[1][j] = m[4 + j] - m[8 + j];
    sr[2][j] = m[4 + j] + m[8 + j] + m[12 + j];
  }
#pragma unroll
  for (int p = 0; p < 3; ++p) {
    float* o = dW + (((long long)co * Cin + ci) * 3 + p) * 3;
    o[0] = sr[p][0] + sr[p][1] + sr[p][2];
    o[1] = sr[p][1] - sr[p][2];
    o[2] = sr[p][1] + sr[p][2] + sr[p][3];
  }
}

}  // namespace

extern "C" long long cy_wino_packed_floats(int Cin, int N) {
  return (long long)((Cin + 7) / 8) * 16 * 2 * ((N + 63) / 64 * 64) * 4;
}

extern "C" int cy_wino_pack_weights(const float* W, float* U, int Cout, int Cin, int transpose, void* stream) {
  CY_REQUIRE(W && U && Cout > 0 && Cin > 0, "cy_wino_pack_weights: bad arguments");
  const int cin_g = transpose ? Cout : Cin, n_g = transpose ? Cin : Cout;
  const int Np = (n_g + 63) / 64 * 64;
  const long long total = cy_wino_packed_floats(cin_g, n_g);
  wino_pack_kernel<<<(unsigned)cy_ceil_div(total / 16, 256), 256, 0, (hipStream_t)stream>>>(W, U, Cout, Cin, transpose, cin_g,
                                                                                            Np, n_g, total);
  CY_LAUNCH_CHECK("cy_wino_pack_weights");
  return 0;
}

namespace {
// out[i] = sum over the shares' slabs, in share order (deterministic)
__global__ void wino_split_sum_kernel(const float* __restrict__ ws, float* __restrict__ Y, int S, long long n4) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n4) return;
  f32x4 v = ((const f32x4*)ws)[i];
  for (int k = 1; k < S; ++k) v += ((const f32x4*)ws)[(long long)k * n4 + i];
  ((f32x4*)Y)[i] = v;
}
// shares of the reduction for a launch without an epilogue (input gradients): at most 4, at least 16 chunks of 8 channels each
int wino_shares(int B, int H, int W, int Cin, int Cout, int plain) {
  if (!plain || Cout % 4 != 0) return 1;
  int dev = 0, ncu = 0;
  if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || ncu <= 0) return 1;
  const long long tiles = (long long)B * ((H + 15) / 16) * ((W + 15) / 16) * ((Cout + 63) / 64);
  long long S = ncu / (tiles > 0 ? tiles : 1);
  if (S > 4) S = 4;
  while (S > 1 && (Cin % (KC * S) != 0 || Cin / (KC * S) < 16)) --S;
  return S < 1 ? 1 : (int)S;
}
}  // namespace

extern "C" long long cy_wino_split_ws_floats(int B, int H, int W, int Cin, int Cout, int plain) {
  if (B <= 0 || H <= 0 || W <= 0 || Cin <= 0 || Cout <= 0) return 0;
  const int S = wino_shares(B, H, W, Cin, Cout, plain);
  return S > 1 ? (long long)S * B * H * W * Cout : 0;
}

extern "C" int cy_conv3x3_winograd(const float* X, const float* U, float* Y, const float* bias, double* stats, float out_slope,
                                   int B, int H, int W, int Cin, int Cout, void* stream) {
  return cy_conv3x3_winograd_ws(X, U, Y, bias, stats, out_slope, B, H, W, Cin, Cout, nullptr, 0, stream);
}

extern "C" int cy_conv3x3_winograd_ws(const float* X, const float* U, float* Y, const float* bias, double* stats, float out_slope,
                                      int B, int H, int W, int Cin, int Cout, float* ws, long long ws_floats, void* stream) {
  CY_REQUIRE(X && U && Y && B > 0 && H > 0 && W > 0 && Cout > 0, "cy_conv3x3_winograd: bad arguments");
  CY_REQUIRE(out_slope >= 0.f && out_slope <= 1.f, "cy_conv3x3_winograd: out_slope=%g must be in [0, 1] (1 = no activation)", (double)out_slope);
  CY_REQUIRE(out_slope == 1.f || stats == nullptr, "cy_conv3x3_winograd: the activation epilogue is for eval-mode forwards (no statistics)");
  CY_REQUIRE(Cin % KC == 0 && Cin >= KC, "cy_conv3x3_winograd: Cin=%d must be a multiple of %d", Cin, KC);
  CY_REQUIRE(Cin <= WINO_MAX_CIN, "cy_conv3x3_winograd: Cin=%d exceeds %d", Cin, WINO_MAX_CIN);
  CY_REQUIRE((((uintptr_t)X | (uintptr_t)U) & 15) == 0, "cy_conv3x3_winograd: operands must be 16-byte aligned");
  CY_REQUIRE((long long)H * W * Cin < (1ll << 29) && (long long)H * W * Cout < (1ll << 29),
             "cy_conv3x3_winograd: image too large for 32-bit byte offsets");
  WinoArgs a;
  a.X = X; a.U = U; a.Y = Y; a.bias = bias; a.stats = stats; a.out_slope = out_slope;
  a.B = B; a.H = H; a.W = W; a.Cin = Cin; a.Cout = Cout; a.Cs = Cin; a.yslab = 0;
  a.Np = (Cout + 63) / 64 * 64;
  a.tbh = (H + 15) / 16; a.tbw = (W + 15) / 16;
  const long long tiles = (long long)B * a.tbh * a.tbw * (a.Np / WN);
  CY_REQUIRE(tiles < (1ll << 31), "cy_conv3x3_winograd: too many tiles");
  // with a workspace, a launch without an epilogue that fills at most half the chip splits its reduction (cy_wino_split_ws_floats)
  int S = 1;
  if (ws != nullptr) {
    S = wino_shares(B, H, W, Cin, Cout, bias == nullptr && stats == nullptr && out_slope == 1.f);
    if (S > 1) {
      CY_REQUIRE(ws_floats >= (long long)S * B * H * W * Cout && ((((uintptr_t)ws | (uintptr_t)Y)) & 15) == 0,
                 "cy_conv3x3_winograd: ws holds %lld floats, the split reduction needs %lld (cy_wino_split_ws_floats), 16-byte aligned",
                 ws_floats, (long long)S * B * H * W * Cout);
      a.Cin = Cin / S; a.yslab = (long long)B * H * W * Cout; a.Y = ws;
    }
  }
  a.ntiles = (int)tiles;
  int dev = 0, ncu = 0;
  hipError_t he = hipGetDevice(&dev);
  if (he == hipSuccess) he = hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev);
  if (he != hipSuccess || ncu <= 0) return cy_set_error((int)he, "cy_conv3x3_winograd: cannot query the CU count: %s", hipGetErrorString(he));
  const long long blocks = tiles < ncu ? tiles : ncu;   // persistent: one block per CU (155 KB of LDS, 512 registers per lane)
  const size_t lds = (size_t)(4 * VU_BUF + 2 * RAW_BUF) * 4;
  int rc = cy_allow_lds(wino_conv_kernel<1>, lds);
  if (rc) return rc;
  rc = cy_allow_lds(wino_conv_kernel<0>, lds);
  if (rc) return rc;
  rc = cy_allow_lds(wino_conv_kernel<2>, lds);
  if (rc) return rc;
  if (a.stats != nullptr) wino_conv_kernel<1><<<(unsigned)blocks, 256, lds, (hipStream_t)stream>>>(a);
  else if (out_slope != 1.f) wino_conv_kernel<2><<<(unsigned)blocks, 256, lds, (hipStream_t)stream>>>(a);
  else wino_conv_kernel<0><<<dim3((unsigned)blocks, (unsigned)S), 256, lds, (hipStream_t)stream>>>(a);
  CY_LAUNCH_CHECK("cy_conv3x3_winograd");
  if (S > 1) {
    const long long n4 = a.yslab / 4;
    wino_split_sum_kernel<<<(unsigned)cy_ceil_div(n4, 256), 256, 0, (hipStream_t)stream>>>(ws, Y, S, n4);
    CY_LAUNCH_CHECK("cy_conv3x3_winograd (split sum)");
  }
  return 0;
}

// ranges of tile groups (= blocks and slabs per (ci, co) tile): about one block per CU whatever the batch and map size
static int wino_wgrad_ranges(int Cin, int Cout) {
  const int tiles = (Cin / 64) * (Cout / 64);
  int r = (256 + tiles - 1) / tiles;
  return r < 1 ? 1 : r;
}
extern "C" long long cy_wino_wgrad_ws_floats(int B, int Cin, int Cout) {
  (void)B;
  return (long long)wino_wgrad_ranges(Cin, Cout) * 16 * Cin * Cout + 1024;   // + the 4 KiB dummy block of the fused-BatchNorm variant
}

static int wino_wgrad_launch(WinoWgradArgs a, bool bnf, float* dW, float* ws, const char* who, hipStream_t s) {
  const int B = a.B, H = a.H, W = a.W, Cin = a.Cin, Cout = a.Cout;
  CY_REQUIRE(a.X && a.dZ && dW && ws && B > 0 && H > 0 && W > 0, "%s: bad arguments", who);
  CY_REQUIRE(Cin % 64 == 0 && Cout % 64 == 0, "%s: Cin=%d and Cout=%d must be multiples of 64", who, Cin, Cout);
  CY_REQUIRE((((uintptr_t)a.X | (uintptr_t)a.dZ) & 15) == 0, "%s: operands must be 16-byte aligned", who);
  CY_REQUIRE((long long)H * W * Cin < (1ll << 29) && (long long)H * W * Cout < (1ll << 29),
             "%s: image too large for 32-bit byte offsets", who);
  a.slab = ws;
  a.gh = (H + 3) / 4; a.gw = (W + 7) / 8;
  a.nrange = wino_wgrad_ranges(Cin, Cout);
  a.dummy = ws + (long long)a.nrange * 16 * Cin * Cout;                 // behind the slabs (cy_wino_wgrad_ws_floats)
  const long long gtot = (long long)B * a.gh * a.gw;
  if (a.nrange > gtot / 8) a.nrange = gtot >= 8 ? (int)(gtot / 8) : 1;   // at least 8 chunks per block (workspace: upper bound)
  CY_REQUIRE(gtot < (1ll << 31), "%s: too many tile groups", who);
  const long long blocks = (long long)a.nrange * (Cin / 64) * (Cout / 64);
  const size_t lds = (size_t)(4 * VU_BUF + RAWW_BUF) * 4;
  int rc = cy_allow_lds(wino_wgrad_kernel<0>, lds);
  if (rc) return rc;
  rc = cy_allow_lds(wino_wgrad_kernel<1>, lds);
  if (rc) return rc;
  rc = cy_allow_lds(wino_wgrad_kernel<2>, lds);
  if (rc) return rc;
  if (bnf && a.premasked) wino_wgrad_kernel<2><<<(unsigned)blocks, 256, lds, s>>>(a);
  else if (bnf) wino_wgrad_kernel<1><<<(unsigned)blocks, 256, lds, s>>>(a);
  else wino_wgrad_kernel<0><<<(unsigned)blocks, 256, lds, s>>>(a);
  CY_LAUNCH_CHECK(who);
  const long long n = (long long)Cin * Cout;
  wino_wgrad_finish_kernel<<<(unsigned)cy_ceil_div(n, 256), 256, 0, s>>>(ws, dW, a.nrange, Cin, Cout);
  CY_LAUNCH_CHECK(who);
  return 0;
}

extern "C" int cy_conv3x3_winograd_wgrad(const float* X, const float* dZ, float* dW, float* ws, int B, int H, int W, int Cin,
                                         int Cout, void* stream) {
  WinoWgradArgs a = {};
  a.X = X; a.dZ = dZ; a.B = B; a.H = H; a.W = W; a.Cin = Cin; a.Cout = Cout;
  return wino_wgrad_launch(a, false, dW, ws, "cy_conv3x3_winograd_wgrad", (hipStream_t)stream);
}

extern "C" int cy_conv3x3_winograd_wgrad_bn(const float* X, const float* Z, const float* dA, float* dZ, const float* scale,
                                            const float* shift, const float* mean, const float* invstd, float slope,
                                            int premasked, const double* red, long long count, float* dW, float* ws, int B,
                                            int H, int W, int Cin, int Cout, void* stream) {
  CY_REQUIRE(Z && dZ && scale && shift && mean && invstd && red && count > 0, "cy_conv3x3_winograd_wgrad_bn: bad arguments");
  CY_REQUIRE((const float*)dZ != dA && (const float*)dZ != Z, "cy_conv3x3_winograd_wgrad_bn: dZ must not alias dA or Z "
             "(several blocks read every element)");
  CY_REQUIRE((((uintptr_t)Z | (uintptr_t)dZ) & 15) == 0, "cy_conv3x3_winograd_wgrad_bn: operands must be 16-byte aligned");
  WinoWgradArgs a = {};
  a.X = X; a.dZ = dA; a.B = B; a.H = H; a.W = W; a.Cin = Cin; a.Cout = Cout;
  a.Z = Z; a.dZout = dZ; a.scale = scale; a.shift = shift; a.mean = mean; a.invstd = invstd; a.red = red;
  a.inv_count = 1.0 / (double)count; a.slope = slope; a.premasked = premasked ? 1 : 0;
  return wino_wgrad_launch(a, true, dW, ws, "cy_conv3x3_winograd_wgrad_bn", (hipStream_t)stream);
}
