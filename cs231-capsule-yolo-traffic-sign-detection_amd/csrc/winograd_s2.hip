// Fused Winograd F(2x2, 2x2) forward for 4x4 / stride 2 / pad 1 convolutions on NHWC fp32 (DarkCapsuleNet conv_3..5,
// models.py:352-363: 24 % of the step), gfx950.
//
// A 4x4 stride-2 pad-1 convolution is a 2x2 stride-1 convolution of the space-to-depth view shifted by one pixel,
//   X'(Y, X, (py, px, c)) = in(2Y - 1 + py, 2X - 1 + px, c)      (zero outside the image),
//   out(y, x, co) = sum_{a,b in {0,1}} sum_q X'(y + a, x + b, q) g'(co, q, a, b),   g'(co,(py,px,c),a,b) = w[co][c][2a+py][2b+px],
// and a 2x2 kernel has the minimal-filtering form F(2x2, 2x2): 9 multiplies per 2x2 outputs instead of 16,
//   Y = A^T [(G g' G^T) (.) (B^T d B)] A,  B^T = [[1,-1,0],[0,1,0],[0,-1,1]], G = [[1,0],[1,1],[0,1]], A^T = [[1,1,0],[0,1,1]]
// (only +-1 coefficients: no rounding beyond the additions).  The kernel never materialises X': a chunk of the
// reduction is one (py, px) and 8 input channels, its 17x33 patch of X' is a stride-2 sampling of the input.
//
// Structure = winograd.hip's forward kernel (see the comments there for the measured cost model): one block = 8 x 16
// tiles (16 x 32 output pixels) x 64 output channels, 4 waves (2 x 2), ONE wave per SIMD; wave (wm, wn) owns 64
// tiles x 32 channels x 9 positions = 18 accumulator tiles (288 registers); per chunk 72 MFMAs per wave, every other
// piece of work (U / patch global loads, LDS stores, the input transform on channel pairs with v_pk_add_f32) in one
// of the 72 slots between them.  9 global loads per 72 MFMAs (the 3x3 kernel has 11 per 64).
//
// Developer instrumentation, compiled out by default (tools/w2prof.py, tools/w2prof_fwd.py):
//   -DW2_PROF    per-block cycle counts of the chunk loop / drain phases and s_memtime stamps inside one chunk per tile
//                (the buffers are passed through otherwise unused arguments named by CY_W2_PROF / CY_W2_STAMPS)
//   -DW2_SKIP=m  knock-out builds: bit 0 drops the input transform, bit 1 the LDS stores of U / patch, bit 2 the global
//                loads, bit 3 reads every patch from the same L2-resident bytes (results are wrong: timing only)
#include <type_traits>
#ifndef W2_SKIP
#define W2_SKIP 0
#endif
#define W2_SKIPPED(k) ((((W2_SKIP) & 1) && ((k) == 4 || (k) == 5)) || (((W2_SKIP) & 2) && ((k) == 1 || (k) == 7)) || (((W2_SKIP) & 4) && ((k) == 2 || (k) == 3)))
#include "common.h"

// The outputs of these layers (and the z tensor the fused BatchNorm sums read) are streamed once and are far larger than the
// 4 MB L2 of an XCD: marked nontemporal they do not push the transformed weights, which every tile re-reads, out of it
// (conv_3 input gradient: 0.43 instead of 3.77 GB fetched per launch, at an unchanged 5.0 ms).
#ifndef CY_NT
#define CY_NT 1
#endif
#if CY_NT
#define CY_NT_STORE(v, p) __builtin_nontemporal_store((v), (p))
#define CY_NT_LOAD(p) __builtin_nontemporal_load((p))
#else
#define CY_NT_STORE(v, p) (*(p) = (v))
#define CY_NT_LOAD(p) (*(p))
#endif


namespace {

constexpr int TR2 = 8, TC2 = 16;            // tile rows / columns per block
constexpr int NT2 = TR2 * TC2;              // 128 tiles
constexpr int PR2 = 2 * TR2 + 1, PC2 = 2 * TC2 + 1;   // 17 x 33 patch of X'
constexpr int NPIX2 = PR2 * PC2;            // 561
constexpr int RAWP2 = 577;                  // >= 561, = 1 (mod 16): the k-quad stride is 4 banks (mod 64)
constexpr int RAW2_BUF = 2 * RAWP2 * 4;     // floats: [kq][pixel][4]
constexpr int SLABV = NT2 * 4 + 4;          // floats per (xi, kq) slab of V
constexpr int SLABU = 64 * 4 + 4;           // floats per (xi, kq) slab of U
constexpr int V2_BUF = 18 * SLABV, U2_BUF = 18 * SLABU;
constexpr int NRAWQ = 5, NUQ = 5;           // float4 items per thread: 1122 patch items, 1152 U items over 256 threads

struct Wino2Args {
  const float* X; const float* U; float* Y; const float* bias; double* stats;
  const float* in_scale; const float* in_shift; float in_slope;   // optional: the input is lrelu(X * scale[c] + shift[c])
  int B, H, W, Cin, Cout, Np, Ho, Wo, tbh, tbw;
  int ntiles;                               // B * tbh * tbw * Np/64 output tiles, walked by a persistent grid
  // input-gradient mode only: Y = dX [B][Hx][Wx][Cx] of the layer, scattered from the (Ho x Wo = Hx/2+1 x Wx/2+1) grid of
  // the space-to-depth view; optional BatchNorm-backward sums of the producer block (cy_conv_gemm_t.bn_*)
  int Hx, Wx, Cx;
  const float* bn_z; const float* bn_scale; const float* bn_shift; const float* bn_mean; const float* bn_invstd;
  double* bn_red; float bn_slope;
  float out_slope;                          // ACT (forward, eval mode): Y = lrelu(conv + bias) with this slope, BatchNorm folded into U / bias
};

// 18 accumulator tiles = 288 registers, the AGPR file has 256: left to itself the compiler shuttles accumulators
// between the two files in every chunk (288 v_accvgpr_write per chunk measured).  The MFMAs are therefore written
// with an explicit register class: positions 0..7 accumulate in AGPRs, position 8 (32 registers) in arch VGPRs.
__device__ __forceinline__ void mfma_a(f32x16& c, float a, float b) {
  asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+a"(c) : "v"(a), "v"(b));
}
__device__ __forceinline__ void mfma_v(f32x16& c, float a, float b) {
  asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+v"(c) : "v"(a), "v"(b));
}
__device__ __forceinline__ f32x2 pk_sub(f32x2 x, f32x2 y) {
  f32x2 r;
  asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(r) : "v"(x), "v"(y));
  return r;
}
// lrelu(v * sc + sh) on 4 channels with packed math (the producer's BatchNorm + LeakyReLU applied on the way into LDS,
// so that the activation tensor never exists in HBM); slope in (0, 1]: lrelu(y) = max(y, slope * y)
__device__ __forceinline__ f32x4 affine_lrelu4(f32x4 v, f32x4 sc, f32x4 sh, float slope) {
  f32x2 lo = {v[0], v[1]}, hi = {v[2], v[3]};
  const f32x2 sl = {slope, slope};
  asm("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(lo) : "v"(lo), "v"(f32x2{sc[0], sc[1]}), "v"(f32x2{sh[0], sh[1]}));
  asm("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(hi) : "v"(hi), "v"(f32x2{sc[2], sc[3]}), "v"(f32x2{sh[2], sh[3]}));
  f32x2 lo2, hi2;
  asm("v_pk_mul_f32 %0, %1, %2" : "=v"(lo2) : "v"(lo), "v"(sl));
  asm("v_pk_mul_f32 %0, %1, %2" : "=v"(hi2) : "v"(hi), "v"(sl));
  f32x4 r;                                  // plain v_max_f32: fmaxf() would first canonicalise both operands (3 ops)
  asm("v_max_f32 %0, %1, %2" : "=v"(r[0]) : "v"(lo[0]), "v"(lo2[0]));
  asm("v_max_f32 %0, %1, %2" : "=v"(r[1]) : "v"(lo[1]), "v"(lo2[1]));
  asm("v_max_f32 %0, %1, %2" : "=v"(r[2]) : "v"(hi[0]), "v"(hi2[0]));
  asm("v_max_f32 %0, %1, %2" : "=v"(r[3]) : "v"(hi[1]), "v"(hi2[1]));
  return r;
}
__device__ __forceinline__ float acc_elem(float a_elem) {   // one accumulator element, read where the statement stands
  float x;
  asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(x) : "a"(a_elem));
  return x;
}

// ---- compile-time schedule of one chunk: 72 slots; slot s issues the MFMA of position s / 8, row tile s & 1,
// k-step (s >> 1) & 3 (consecutive MFMAs alternate between the position's two accumulators).  Position xi + 1's three
// fragments (A0, A1, B) are fetched in the first three slots of position xi.
// Loads retire in order: the U loads of a chunk are issued BEFORE its patch loads because S_U (slots 3, 4 of the next
// chunk) consumes them first -- with the two interleaved, S_U waited for the youngest load and so for every patch load
// (HBM latency) of the chunk.  Each patch load directly follows the LDS store that frees its registers.
constexpr int S2_SU[2] = {3, 4};
constexpr int S2_GU[5] = {5, 6, 7, 11, 12};
constexpr int S2_SRAW[5] = {13, 15, 20, 22, 27};
constexpr int S2_GRAW[5] = {14, 19, 21, 23, 28};
constexpr int S2_TRD[9] = {29, 30, 31, 35, 36, 37, 38, 39, 43};
constexpr int S2_TV[6] = {44, 45, 46, 47, 51, 52};
// cursor bookkeeping for the next chunk (scalar work, hidden under the MFMAs instead of standing between two chunks)
constexpr int S2_ADV_RAW = 53, S2_ADV_U = 54;
constexpr int s2_find(const int* list, int n, int s) {
  for (int i = 0; i < n; ++i) if (list[i] == s) return i;
  return -1;
}
// kinds: 1 S_U (LDS writes of U(c+1): 3 + 2)  7 S_raw (one float4 of patch c+2)  2 G_raw (one load of patch c+3)
//        3 G_U (one load of U c+2)  4 T_rd (two float2 of patch c+1)  5 T_v (one row of V for one of the two items)
//        8 / 9 advance the patch / U cursor to the next chunk (after this chunk's loads were issued)
constexpr int s2_kind(int s) {
  return s2_find(S2_SU, 2, s) >= 0 ? 1 : s2_find(S2_SRAW, 5, s) >= 0 ? 7 : s2_find(S2_GRAW, 5, s) >= 0 ? 2
       : s2_find(S2_GU, 5, s) >= 0 ? 3 : s2_find(S2_TRD, 9, s) >= 0 ? 4 : s2_find(S2_TV, 6, s) >= 0 ? 5
       : s == S2_ADV_RAW ? 8 : s == S2_ADV_U ? 9 : 0;
}
constexpr int s2_idx(int s) {
  const int k = s2_kind(s);
  return k == 1 ? s2_find(S2_SU, 2, s) : k == 7 ? s2_find(S2_SRAW, 5, s) : k == 2 ? s2_find(S2_GRAW, 5, s)
       : k == 3 ? s2_find(S2_GU, 5, s) : k == 4 ? s2_find(S2_TRD, 9, s) : k == 5 ? s2_find(S2_TV, 6, s) : 0;
}
// LOWER bound of the LDS instructions a slot's side work issues (two float2 reads may merge into one ds_read2_b64;
// exec-masked stores may be skipped by a whole wave)
constexpr int s2_side_lds(int s) {
  const int k = W2_SKIPPED(s2_kind(s)) ? 0 : s2_kind(s), i = s2_idx(s);
  return k == 1 ? (i == 0 ? 3 : 2) : k == 7 ? 1 : k == 4 ? 1 : k == 5 ? 3 : 0;
}
constexpr int s2_frag_lds(int s) { return ((s & 7) < 3 && s + 8 < 72) ? 1 : 0; }
// LDS operations younger than position xi's last fragment (B) when its first MFMA issues
constexpr int s2_younger(int xi) {
  if (xi == 0) return 0;
  int n = s2_side_lds(8 * (xi - 1) + 2);
  for (int s = 8 * (xi - 1) + 3; s < 8 * xi; ++s) n += s2_frag_lds(s) + s2_side_lds(s);
  return n > 14 ? 14 : n;
}

// MODE 0: forward (X = layer input, sampled with stride 2 per (py, px) class).
// MODE 1: input gradient.  dX'(Y, X, q) = sum_{a',b'} D(Y + a', X + b', co) h(q, co, a', b') with D(Y, X) = dY(Y - 1, X - 1)
// (zero outside) and h(q, co, a', b') = g'(co, q, 1 - a', 1 - b'): the same 2x2 "valid" convolution, over the plain NHWC
// tensor dY shifted by one pixel, with K = Cout of the layer (chunks of 8 output channels) and N = 4 Cin; the epilogue
// scatters element (Y, X, q = (py, px, c)) to dX(2Y - 1 + py, 2X - 1 + px, c).
// AFFINE: the input is lrelu(X * in_scale[c] + in_shift[c]) (forward only), applied on the way from registers to LDS.
template <int MODE, bool AFFINE, bool ACT = false>
__global__ __launch_bounds__(256, 1) void wino2_conv_kernel(Wino2Args a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* Vs = smem;                         // [2][V2_BUF]
  float* Us = smem + 2 * V2_BUF;            // [2][U2_BUF]
  float* Rs = smem + 2 * V2_BUF + 2 * U2_BUF;   // [2][RAW2_BUF]
  float* Aff = Rs + 2 * RAW2_BUF;           // [2][Cin] input scale / shift (only with in_scale)

  const int t = threadIdx.x, lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int wm = wave & 1, wn = wave >> 1;
  const int li = lane & 31, lh = lane >> 5;
  constexpr bool affine = AFFINE;
  if (affine) {
    for (int i = t; i < a.Cin; i += 256) { Aff[i] = a.in_scale[i]; Aff[a.Cin + i] = a.in_shift[i]; }
    __syncthreads();
  }

  // Persistent grid, one block per CU, as in winograd.hip: block j walks the tiles vid(j), vid(j) + G, ... as ONE
  // continuous stream of chunks (the input gradient has only Cout/8 chunks per tile: without this every tile pays a
  // prologue and a block launch).  XCD-aware ids: the Np/64 tiles of one patch are in flight together on one L2.
  unsigned vid = blockIdx.x;
  if ((gridDim.x & 7u) == 0) vid = (blockIdx.x & 7u) * (gridDim.x >> 3) + (blockIdx.x >> 3);
  const int nblk = a.Np / 64;
  const int ntile_mine = (a.ntiles - (int)vid + (int)gridDim.x - 1) / (int)gridDim.x;   // >= 1 (grid <= ntiles)
  const int cpp = a.Cin / 8;                // chunks per (py, px)
  const int nchunk = MODE == 0 ? 4 * cpp : cpp;
  constexpr int SXY = MODE == 0 ? 2 : 1;    // input pixels per grid step
  struct TilePos { int nb, b, Y0, X0; };    // Y0 / X0 = first output row / column = first X' row / column
  auto tile_pos = [&](int k) {              // k-th tile of this block (uniform)
    const int id = (int)vid + k * (int)gridDim.x;
    TilePos p;
    p.nb = id % nblk;
    int rest = id / nblk;
    const int tbx = rest % a.tbw; rest /= a.tbw;
    const int tby = rest % a.tbh;
    p.b = rest / a.tbh; p.Y0 = tby * (2 * TR2); p.X0 = tbx * (2 * TC2);
    return p;
  };

  // ---- patch loader: item = t + 256 q -> 16 pixels x 2 k-quads per 32 items (conflict-free b128 LDS stores, both
  // 16-byte halves of a pixel's 32 bytes in one wave-load).  pix(q) = pix0 + 128 q, LDS offset roff0 + 512 q; only the
  // last round (q = 4) is partial.  The tile-dependent values belong to the tile the patch stream is in (set_raw_tile).
  const int kq_of_thread = (t >> 4) & 1;    // k-quad of every item of this thread (256 q keeps bit 4)
  const int pix0 = (t >> 5) * 16 + (t & 15);
  const int roff0 = (kq_of_thread * RAWP2 + pix0) * 4;
  const bool rlast_ok = pix0 + 128 * (NRAWQ - 1) < NPIX2;
  const int roff4 = roff0 + 512 * (rlast_ok ? NRAWQ - 1 : NRAWQ - 2);   // no exec masks in the loop: duplicates instead
  constexpr int NCLS = MODE == 0 ? 4 : 1;   // (py, px) classes of a tile's patches
  unsigned gvoff[NRAWQ];                    // byte offset of class (0, 0), channel 0 from the image base (mod 2^32: the
                                            // pixel of class (0, 0) may lie above / left of the image)
  unsigned okbits = 0;                      // border tiles: bit 5 cls + q = item q of class cls is an image pixel
  bool blk_fast = false;                    // uniform: the whole patch of the stream's tile lies inside the image
  const char* ximg = nullptr;               // uniform
  auto set_raw_tile = [&](int k) {
    const TilePos p = tile_pos(k);
    blk_fast = p.Y0 >= 1 && p.X0 >= 1 && SXY * (p.Y0 + PR2 - 1) - 1 + (MODE == 0 ? 1 : 0) <= a.H - 1 &&
               SXY * (p.X0 + PC2 - 1) - 1 + (MODE == 0 ? 1 : 0) <= a.W - 1;
    ximg = (const char*)(a.X + (long long)p.b * a.H * a.W * a.Cin);
    okbits = 0;
#pragma unroll
    for (int q = 0; q < NRAWQ; ++q) {
      const int pix = pix0 + 128 * ((q < NRAWQ - 1 || rlast_ok) ? q : q - 1);   // past the patch: repeat the thread's item q - 1
      const int pr = pix / PC2, pc = pix - pr * PC2;
      const int iy = SXY * (p.Y0 + pr) - 1, ix = SXY * (p.X0 + pc) - 1;
      constexpr bool item = true;
      gvoff[q] = (unsigned)(((iy * a.W + ix) * a.Cin + kq_of_thread * 4) * 4);
      if (!blk_fast) {
#pragma unroll
        for (int cls = 0; cls < NCLS; ++cls) {
          const bool ok = item && (unsigned)(iy + (cls >> 1)) < (unsigned)a.H && (unsigned)(ix + (cls & 1)) < (unsigned)a.W;
          okbits |= (unsigned)ok << (5 * cls + q);
        }
      }
    }
  };
  // stream cursors (tile index within this block, chunk): advance by one chunk, stop at the very last chunk
  auto advance = [&](int& k, int& c) {
    if (c + 1 < nchunk) { ++c; return false; }
    if (k + 1 < ntile_mine) { ++k; c = 0; return true; }
    return false;
  };
  // ---- U loader: 18 segments (xi * 2 + kq) of 64 channels x float4 per chunk; item t + 256 q = segment (t >> 6) + 4 q,
  // channel t & 63: one lane offset, uniform strides per q; only the last round (q = 4: segments 16, 17) is partial
  const long long uchunk = (long long)18 * a.Np * 4;           // floats per chunk
  const bool ulast_ok = t < 128;
  const unsigned uvoff0 = (unsigned)((((t >> 6) * a.Np) + (t & 63)) * 16);   // bytes from the tile's first output channel
  const long long useg4 = (long long)4 * a.Np * 16;                          // bytes per q
  const unsigned uvoff4 = ulast_ok ? uvoff0 : uvoff0 - (unsigned)(2 * a.Np * 16);   // q = 4: segments 18, 19 do not exist
  const int uoff0 = (t >> 6) * SLABU + (t & 63) * 4;                         // + q * 4 * SLABU
  const int uoff4 = uoff0 + (NUQ - 1) * 4 * SLABU - (ulast_ok ? 0 : 2 * SLABU);   // waves 2, 3 repeat segments 16, 17
  const char* ubase = nullptr;              // uniform: a.U + first output channel of the weight stream's tile
  auto set_u_tile = [&](int k) { ubase = (const char*)(a.U + (long long)tile_pos(k).nb * 64 * 4); };
  // ---- transform items: channel pair tch = t & 1 of k-quad tkq, tile column (t >> 2) & 15, tile rows t >> 6 and + 4
  const int tch = t & 1, tkq = (t >> 1) & 1, ttx = (t >> 2) & 15, tty = t >> 6;
  const int tbase = (tkq * RAWP2 + (2 * tty) * PC2 + 2 * ttx) * 4 + 2 * tch;     // item 1: + 8 * PC2 * 4
  const int vdst = tkq * SLABV + (tty * TC2 + ttx) * 4 + 2 * tch;                // item 1: + 4 * TC2 * 4; + xi * 2 * SLABV

  int km = 0, cm = 0, ku = 0, cu = 0, kr = 0, cr = 0;   // stream cursors (tile of this block, chunk): MFMAs, U loads, patch loads
  f32x4 graw[NRAWQ], gu[NUQ];
  // The slots of the MFMA stream must not branch (a uniform branch costs a one-wave-per-SIMD kernel ~40 cycles, ten of
  // them 7 % of a chunk): every patch load is `uniform base + goff[q]`, every LDS store a select on one mask bit.  goff /
  // okm_cur belong to the (tile, class) the patch cursor stands in and change only when it enters a new class or tile;
  // pad items read the image's first bytes (a valid address) and are stored as exact zeros.
  unsigned goff[NRAWQ];                     // byte offset of the cursor's class from the image base, 0 for pad items
  unsigned okm_cur = 0;                     // bit q = item q of the cursor's (tile, class) is an image pixel
  unsigned smask = 0;                       // the same for the patch held in graw (S_raw stores it next)
  int rcc = 0, rcls = 0;                    // uniform: chunk within the class, class of the patch cursor
  auto set_raw_class = [&](int cls) {
    okm_cur = blk_fast ? 0x1fu : (okbits >> (5 * cls)) & 0x1fu;
    const unsigned class_off = (unsigned)((((cls >> 1) * a.W + (cls & 1)) * a.Cin) * 4);
#pragma unroll
    for (int q = 0; q < NRAWQ; ++q) goff[q] = (okm_cur >> q) & 1 ? gvoff[q] + class_off : 0u;
  };
  auto advance_raw = [&]() {                // one chunk further; stops at the very last chunk of the block's stream
    if (cr + 1 < nchunk) {
      ++cr; ++rcc;
      if (MODE == 0 && rcc == cpp) { rcc = 0; ++rcls; set_raw_class(rcls); }
    } else if (kr + 1 < ntile_mine) {
      ++kr; cr = 0; rcc = 0; rcls = 0;
      set_raw_tile(kr);
      set_raw_class(0);
    }
  };
  auto Graw1 = [&](int q, const char* xc, f32x4& dst) { dst = *(const f32x4*)(xc + goff[q]); };
  f32x4 asc = {1.f, 1.f, 1.f, 1.f}, ash = {0.f, 0.f, 0.f, 0.f};     // scale / shift of the chunk that is being stored
  auto set_affine = [&](int c0) {           // c0 = first channel of the chunk whose patch goes to LDS next
    if (!affine) return;
    asc = *(const f32x4*)(Aff + c0 + kq_of_thread * 4); ash = *(const f32x4*)(Aff + a.Cin + c0 + kq_of_thread * 4);
  };
  auto Sraw1 = [&](float* rb, int q, const f32x4& src, unsigned mask) {   // mask: bit q = src holds an image pixel
    const f32x4 v = affine ? affine_lrelu4(src, asc, ash, a.in_slope) : src;
    *(f32x4*)(rb + (q < NRAWQ - 1 ? roff0 + 512 * q : roff4)) = (mask >> q) & 1 ? v : f32x4{0.f, 0.f, 0.f, 0.f};   // padding: exact 0
  };
  auto GU1 = [&](int q, const char* uchunk_p, f32x4& dst) {
    dst = *(const f32x4*)(uchunk_p + q * useg4 + (q < NUQ - 1 ? uvoff0 : uvoff4));
  };
  auto uchunk_of = [&](int f) { return ubase + (long long)f * uchunk * 4; };   // uniform
  auto SU1 = [&](float* ub, int q, const f32x4& src) {
    *(f32x4*)(ub + (q < NUQ - 1 ? uoff0 + q * 4 * SLABU : uoff4)) = src;
  };
  f32x2 xv[2][3][3];                        // the two 3x3 patches of this thread, two channels each
  auto Vrow = [&](float* vb, int it, int R) {   // row R of V = B^T d B of item `it`
    f32x2 t0[3];
#pragma unroll
    for (int cc = 0; cc < 3; ++cc)
      t0[cc] = R == 0 ? pk_sub(xv[it][0][cc], xv[it][1][cc]) : R == 1 ? xv[it][1][cc] : pk_sub(xv[it][2][cc], xv[it][1][cc]);
    float* v = vb + it * (4 * TC2 * 4);
    *(f32x2*)(v + (R * 3 + 0) * 2 * SLABV) = pk_sub(t0[0], t0[1]);
    *(f32x2*)(v + (R * 3 + 1) * 2 * SLABV) = t0[1];
    *(f32x2*)(v + (R * 3 + 2) * 2 * SLABV) = pk_sub(t0[2], t0[1]);
  };
  auto Tall = [&](int buf_raw, int buf_v) {
    const float* rb = Rs + buf_raw * RAW2_BUF + tbase;
#pragma unroll
    for (int it = 0; it < 2; ++it)
#pragma unroll
      for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int cc = 0; cc < 3; ++cc) xv[it][r][cc] = *(const f32x2*)(rb + it * (8 * PC2 * 4) + (r * PC2 + cc) * 4);
    float* vb = Vs + buf_v * V2_BUF + vdst;
#pragma unroll
    for (int it = 0; it < 2; ++it)
#pragma unroll
      for (int R = 0; R < 3; ++R) Vrow(vb, it, R);
  };

  // ---- prologue (once per block).  State at the top of stream position f (tile km, chunk cm): V[f&1] / U[f&1] =
  // position f, raw[(f+1)&1] = patch of position f+1, registers gu = U of position f+1, graw = patch of position f+2;
  // the cursors (ku, cu) / (kr, cr) stand at the positions whose loads are issued during f: f+2 for U, f+3 for the patch.
  int s_c0 = 0;                             // first channel of the patch held in graw
  {
    f32x4 graw1[NRAWQ];
    set_raw_tile(0);
    set_raw_class(0);
    set_u_tile(0);
#pragma unroll
    for (int q = 0; q < NRAWQ; ++q) Graw1(q, ximg, graw[q]);
#pragma unroll
    for (int q = 0; q < NUQ; ++q) GU1(q, uchunk_of(0), gu[q]);
    set_affine(0);
#pragma unroll
    for (int q = 0; q < NRAWQ; ++q) Sraw1(Rs, q, graw[q], okm_cur);
    advance_raw();
    const int c1 = rcc * 8;
    const unsigned m1 = okm_cur;
#pragma unroll
    for (int q = 0; q < NRAWQ; ++q) Graw1(q, ximg + rcc * 32, graw1[q]);
#pragma unroll
    for (int q = 0; q < NUQ; ++q) SU1(Us, q, gu[q]);
    if (advance(ku, cu)) set_u_tile(ku);
#pragma unroll
    for (int q = 0; q < NUQ; ++q) GU1(q, uchunk_of(cu), gu[q]);
    __syncthreads();
    Tall(0, 0);
    set_affine(c1);
#pragma unroll
    for (int q = 0; q < NRAWQ; ++q) Sraw1(Rs + RAW2_BUF, q, graw1[q], m1);
    advance_raw();
    s_c0 = rcc * 8; smask = okm_cur;
#pragma unroll
    for (int q = 0; q < NRAWQ; ++q) Graw1(q, ximg + rcc * 32, graw[q]);
    __syncthreads();
    advance_raw();
    if (advance(ku, cu)) set_u_tile(ku);
  }
  const int fragA = lh * SLABV + (wm * 64 + li) * 4;     // + mi * 128
  const int fragB = lh * SLABU + (wn * 32 + li) * 4;
  int c_next = 0;                           // stream position f (only its parity is used)
  const char* up_ = uchunk_of(cu);          // G_U(f+2); at the stream's end the cursors stop: the tail re-loads valid data
  const char* xp_ = ximg + rcc * 32;        // G_raw(f+3)
  set_affine(s_c0);                         // of the chunk S_raw stores next
#ifdef W2_PROF
  long long pf_loop = 0, pf_ep1 = 0, pf_ep2 = 0, pf_t0 = 0, pf_t1 = 0, pf_t2 = 0, pf_nfast = 0;
#endif
  // input-gradient mode: the producer's BatchNorm-backward sums (sum d, sum d * xhat per channel) stay in the lanes'
  // registers over ALL tiles of the block and are reduced (wave shuffles, LDS, double atomics) when the block's
  // channel block changes -- once, at the end, when the grid is a multiple of N/64.  Reduced per tile, the 24 dependent
  // shuffles, the barrier and the atomics stood in every drain (~4k of its 22k cycles at conv_3).
  f32x4 b1 = {0.f, 0.f, 0.f, 0.f}, b2 = b1;
  int bn_nb = -1;
  auto flush_bn = [&](int nbx, float* bred) {          // bred: 1 KiB of LDS nobody else uses right now
    const int q0x = nbx * 64;
    const int cb0x = q0x - (q0x / a.Cx) * a.Cx;
    const int c4x = lane & 7;
#pragma unroll
    for (int k = 0; k < 4; ++k)
#pragma unroll
      for (int msk = 8; msk < 64; msk <<= 1) { b1[k] += __shfl_xor(b1[k], msk, 64); b2[k] += __shfl_xor(b2[k], msk, 64); }
    if (lane < 8) {
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int col = wn * 32 + c4x * 4 + k;
        bred[(wm * 64 + col) * 2 + 0] = b1[k];
        bred[(wm * 64 + col) * 2 + 1] = b2[k];
      }
    }
    __syncthreads();
    if (t < 64) {
      double* rd = a.bn_red + (size_t)(blockIdx.x % CY_STATS_COPIES) * a.Cx * 2;
      atomicAdd(rd + 2 * (cb0x + t), (double)bred[t * 2] + (double)bred[(64 + t) * 2]);
      atomicAdd(rd + 2 * (cb0x + t) + 1, (double)bred[t * 2 + 1] + (double)bred[(64 + t) * 2 + 1]);
    }
    __syncthreads();
    b1 = f32x4{0.f, 0.f, 0.f, 0.f}; b2 = b1;
  };
  for (km = 0; km < ntile_mine; ++km) {
#ifdef W2_PROF
  pf_t0 = clock64();
#endif
  f32x16 acc[9][2];
#pragma unroll
  for (int xi = 0; xi < 9; ++xi)
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[xi][mi][r] = 0.f;
  for (cm = 0; cm < nchunk; ++cm, ++c_next) {
    const int c = c_next;
    const float* vb_ = Vs + (c & 1) * V2_BUF + fragA;
    const float* ub_ = Us + (c & 1) * U2_BUF + fragB;
    const float* rb_ = Rs + ((c + 1) & 1) * RAW2_BUF + tbase;       // T(f+1) reads ...
    float* vw_ = Vs + ((c + 1) & 1) * V2_BUF + vdst;                // ... and writes (harmless after the last position)
    float* uw_ = Us + ((c + 1) & 1) * U2_BUF;                       // S_U(f+1)
    float* rw_ = Rs + (c & 1) * RAW2_BUF;                           // S_raw(f+2) -> raw[(f+2)&1]
#ifdef W2_PROF
    unsigned long long st_[6];
    st_[0] = __builtin_amdgcn_s_memtime();
#define W2_STAMP(s_) if ((s_) == 24) st_[1] = __builtin_amdgcn_s_memtime(); if ((s_) == 48) st_[2] = __builtin_amdgcn_s_memtime();
#else
#define W2_STAMP(s_)
#endif
    f32x4 fa_[2][2], fb_[2];                // fragment sets, indexed by position & 1
    fa_[0][0] = *(const f32x4*)(vb_);
    fa_[0][1] = *(const f32x4*)(vb_ + 128);
    fb_[0] = *(const f32x4*)(ub_);
#define W2SLOT(SIDX)                                                                                \
    {                                                                                               \
      constexpr int sidx = (SIDX);                                                                  \
      constexpr int xi = sidx >> 3, w_ = sidx & 7, mi = w_ & 1, e = w_ >> 1;                        \
      W2_STAMP(sidx)                                                                                \
      if (w_ == 0) __builtin_amdgcn_s_waitcnt(0xC07F | (s2_younger(xi) << 8));                      \
      if (xi < 8) mfma_a(acc[xi][mi], fa_[xi & 1][mi][e], fb_[xi & 1][e]);                          \
      else mfma_v(acc[xi][mi], fa_[xi & 1][mi][e], fb_[xi & 1][e]);                                 \
      if (w_ < 3 && xi + 1 < 9) {                                                                   \
        constexpr int nx = (xi + 1 < 9) ? xi + 1 : 0;                                               \
        if (w_ == 0) fa_[nx & 1][0] = *(const f32x4*)(vb_ + nx * 2 * SLABV);                        \
        if (w_ == 1) fa_[nx & 1][1] = *(const f32x4*)(vb_ + nx * 2 * SLABV + 128);                  \
        if (w_ == 2) fb_[nx & 1] = *(const f32x4*)(ub_ + nx * 2 * SLABU);                           \
      }                                                                                             \
      constexpr int kind0 = s2_kind(sidx), k_ = s2_idx(sidx);                                       \
      constexpr int kind = W2_SKIPPED(kind0) ? 0 : kind0;                                           \
      if (kind == 1) {                      /* U(c+1): registers -> LDS */                         \
        if (k_ == 0) { SU1(uw_, 0, gu[0]); SU1(uw_, 1, gu[1]); SU1(uw_, 2, gu[2]); }                \
        else { SU1(uw_, 3, gu[3]); SU1(uw_, 4, gu[4]); }                                            \
      } else if (kind == 7) {               /* patch of chunk c+2: registers -> LDS */             \
        Sraw1(rw_, k_ % NRAWQ, graw[k_ % NRAWQ], smask);                                            \
      } else if (kind == 2) {               /* patch load of chunk c+3 */                          \
        Graw1(k_ % NRAWQ, xp_, graw[k_ % NRAWQ]);                                                   \
      } else if (kind == 3) {               /* weight load of chunk c+2 */                         \
        GU1(k_ % NUQ, up_, gu[k_ % NUQ]);                                                           \
      } else if (kind == 4) {               /* patch of chunk c+1: two float2 */                   \
        constexpr int i0 = 2 * (k_ % 9), i1 = i0 + 1;                                               \
        xv[i0 / 9][(i0 % 9) / 3][i0 % 3] = *(const f32x2*)(rb_ + (i0 / 9) * (8 * PC2 * 4) + (((i0 % 9) / 3) * PC2 + i0 % 3) * 4); \
        xv[i1 / 9][(i1 % 9) / 3][i1 % 3] = *(const f32x2*)(rb_ + (i1 / 9) * (8 * PC2 * 4) + (((i1 % 9) / 3) * PC2 + i1 % 3) * 4); \
      } else if (kind == 5) {               /* one row of V of one item */                         \
        Vrow(vw_, (k_ % 6) / 3, (k_ % 6) % 3);                                                      \
      } else if (kind == 8) {               /* patch cursor -> f+4; graw now holds f+3 */          \
        s_c0 = rcc * 8; smask = okm_cur;                                                            \
        advance_raw();                                                                              \
        xp_ = ximg + rcc * 32;                                                                      \
        set_affine(s_c0);                                                                           \
      } else if (kind == 9) {               /* U cursor -> f+3 */                                  \
        if (advance(ku, cu)) set_u_tile(ku);                                                        \
        up_ = uchunk_of(cu);                                                                        \
      }                                                                                             \
      __builtin_amdgcn_sched_barrier(0);                                                            \
    }
#define W2SLOT8(B) W2SLOT((B)) W2SLOT((B) + 1) W2SLOT((B) + 2) W2SLOT((B) + 3) W2SLOT((B) + 4) W2SLOT((B) + 5) W2SLOT((B) + 6) W2SLOT((B) + 7)
    W2SLOT8(0) W2SLOT8(8) W2SLOT8(16) W2SLOT8(24) W2SLOT8(32) W2SLOT8(40) W2SLOT8(48) W2SLOT8(56) W2SLOT8(64)
#undef W2SLOT8
#undef W2SLOT
#undef W2_STAMP
#ifdef W2_PROF
    st_[3] = __builtin_amdgcn_s_memtime();
#endif
    __syncthreads();                        // the only barrier of the chunk
#ifdef W2_PROF
    st_[4] = __builtin_amdgcn_s_memtime();
#endif
#ifdef W2_PROF
    st_[5] = __builtin_amdgcn_s_memtime();
    if (MODE == 0 && a.bn_scale != nullptr && t == 0 && cm == (nchunk > 5 ? 5 : 0)) {      // MODE 0 hack: stamps of one chunk per tile
      unsigned long long* sb = (unsigned long long*)a.bn_scale + ((size_t)blockIdx.x * 16 + (km & 15)) * 8;
      for (int i = 0; i < 6; ++i) sb[i] = st_[i];
    }
#endif
  }
  // ======== tile km is complete: drain the accumulators.  V[f&1] / U[f&1] (f = the tile's last position) and
  // raw[(f+1)&1] were consumed; V / U[(f+1)&1] and raw[f&1] already hold the next tile's first chunks and must survive.
  const int cl = c_next - 1;
#ifdef W2_PROF
  pf_t1 = clock64(); pf_loop += pf_t1 - pf_t0;
  {
    const TilePos q_ = tile_pos(km);
    const bool tf = q_.Y0 >= 1 && q_.X0 >= 1 && SXY * (q_.Y0 + PR2 - 1) - 1 + (MODE == 0 ? 1 : 0) <= a.H - 1 &&
                    SXY * (q_.X0 + PC2 - 1) - 1 + (MODE == 0 ? 1 : 0) <= a.W - 1;
    if (tf) { pf_ep2 += pf_t1 - pf_t0; pf_nfast += 1; }   // MODE 0 only: loop cycles and count of tiles inside the image
  }
#endif
  const TilePos tp = tile_pos(km);
  const int nb = tp.nb, b = tp.b, Y0 = tp.Y0, X0 = tp.X0;
  float* ow = wave < 2 ? Vs + (cl & 1) * V2_BUF + wave * 4096 : wave == 2 ? Us + (cl & 1) * U2_BUF
                                                                            : Rs + ((cl + 1) & 1) * RAW2_BUF;   // 16 KiB per wave
  float* red = Vs + (cl & 1) * V2_BUF + 8192;     // [2 wm][64][2], behind the scratch of waves 0 and 1
  if (MODE == 1) {
    // ---- input gradient: the same A^T M A; grid point (Yg, Xg) of q block nb = one (py, px) class and 64 channels of it.
    // One wave is alone on its SIMD: every instruction of this drain is exposed, so addresses are a uniform pointer per
    // (half, step) plus ONE 32-bit lane offset, validity is two per-lane bit masks (none at all for tiles inside the
    // image), and the BatchNorm z values of a half are requested before its output transform runs.
    const int q0 = nb * 64;
    const int ph = q0 / a.Cx, cb0 = q0 - ph * a.Cx, py = ph >> 1, px = ph & 1;
    const int c4 = lane & 7;
    const int cch = cb0 + wn * 32 + c4 * 4;               // first of this lane's 4 channels
    const bool bnb = a.bn_red != nullptr;
    f32x4 bsc = {0.f, 0.f, 0.f, 0.f}, bsh = bsc, bnm = bsc, bis = bsc;
    if (bnb && bn_nb >= 0 && bn_nb != nb) {               // (uniform, rare) another channel block: hand over the sums so far
      __syncthreads();
      flush_bn(bn_nb, red);
    }
    bn_nb = nb;
    if (bnb) {
      bsc = *(const f32x4*)(a.bn_scale + cch); bsh = *(const f32x4*)(a.bn_shift + cch);
      bis = *(const f32x4*)(a.bn_invstd + cch);
      bnm = -*(const f32x4*)(a.bn_mean + cch) * bis;      // xhat = z * invstd - mean * invstd
    }
    // step `it` of half mi covers grid rows Y0 + 8 wm + 4 mi + 2 (it >> 3) + lr, columns X0 + 4 (it & 7) + lc
    const int lr = (lane >> 4) & 1, lc = 2 * lh + ((lane >> 3) & 1);
    const int rowstride = 2 * a.Wx * a.Cx, colstride = 2 * a.Cx;                       // floats per grid row / column
    const unsigned lane_off = (unsigned)(lr * rowstride + lc * colstride + c4 * 4);
    const long long tile_off = ((long long)b * a.Hx + (2 * (Y0 + 8 * wm) - 1 + py)) * a.Wx * a.Cx +
                               (long long)(2 * X0 - 1 + px) * a.Cx + cb0 + wn * 32;   // uniform; rows above the image: masked
    float* ytile = a.Y + tile_off;
    const float* ztile = a.bn_z + tile_off;
    // valid grid rows [rlo, rhi), columns [clo, chi) relative to the tile: iy = 2 Yg - 1 + py in [0, Hx) <=> 1 - py <= Yg < Ho - py
    const int rlo = max(1 - py - Y0, 0), rhi = min(a.Ho - py - Y0, 16), clo = max(1 - px - X0, 0), chi = min(a.Wo - px - X0, 32);
    const bool full = rlo == 0 && rhi == 16 && clo == 0 && chi == 32;                  // uniform
    const unsigned rowmask = rhi > rlo ? ((0xffffu >> (16 - rhi + rlo)) << rlo) : 0u;
    const unsigned colmask = chi > clo ? ((0xffffffffu >> (32 - chi + clo)) << clo) : 0u;
    const unsigned myrows = rowmask >> (lr + 8 * wm), mycols = colmask >> lc;          // bit 4 mi + 2 (it >> 3) / 4 (it & 7)
    auto half = [&](auto full_c, auto mi_c) {
      constexpr bool FULL = decltype(full_c)::value;
      constexpr int mi = decltype(mi_c)::value;
      f32x4 zq_[16];
      if (bnb) {
#pragma unroll
        for (int it = 0; it < 16; ++it) {
          const bool ok = FULL || (((myrows >> (4 * mi + 2 * (it >> 3))) & (mycols >> (4 * (it & 7))) & 1u) != 0u);
          const float* zp = ztile + (4 * mi + 2 * (it >> 3)) * rowstride + 4 * (it & 7) * colstride;
          if (ok) zq_[it] = CY_NT_LOAD((const f32x4*)(zp + lane_off));
        }
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int tloc = (r & 3) + 8 * (r >> 2) + 4 * lh;
        float m[9];
#pragma unroll
        for (int xi = 0; xi < 8; ++xi) m[xi] = acc_elem(acc[xi][mi][r]);
        m[8] = acc[8][mi][r];
        float* op = ow + tloc * 128 + li;
        op[0] = (m[0] + m[1]) + (m[3] + m[4]); op[32] = (m[1] + m[2]) + (m[4] + m[5]);
        op[64] = (m[3] + m[4]) + (m[6] + m[7]); op[96] = (m[4] + m[5]) + (m[7] + m[8]);
        __builtin_amdgcn_sched_barrier(0);
      }
#pragma unroll
      for (int it = 0; it < 16; ++it) {
        const int p = it * 8 + (lane >> 3);
        const f32x4 v = *(const f32x4*)(ow + p * 32 + c4 * 4);
        const bool ok = FULL || (((myrows >> (4 * mi + 2 * (it >> 3))) & (mycols >> (4 * (it & 7))) & 1u) != 0u);
        float* yp = ytile + (4 * mi + 2 * (it >> 3)) * rowstride + 4 * (it & 7) * colstride;
        if (ok) {
          f32x4 sv = v;
          if (bnb) {
            // with the fused sums the kernel has d = dX * lrelu'(z * scale + shift) in registers: THAT is what it stores (the
            // producer block's backward then starts from the gradient at the BatchNorm output and needs no mask of its own)
#pragma unroll
            for (int k = 0; k < 4; ++k) {
              const float z = zq_[it][k];
              const float y = __builtin_fmaf(z, bsc[k], bsh[k]);
              const float d = y > 0.f ? v[k] : v[k] * a.bn_slope;
              sv[k] = d;
              b1[k] += d;
              b2[k] = __builtin_fmaf(d, __builtin_fmaf(z, bis[k], bnm[k]), b2[k]);
            }
          }
#ifdef W2_PROF
          if (a.in_slope != 2.f)
#endif
          CY_NT_STORE(sv, (f32x4*)(yp + lane_off));
        }
      }
      __builtin_amdgcn_sched_barrier(0);
    };
    constexpr std::integral_constant<int, 0> H0{};
    constexpr std::integral_constant<int, 1> H1{};
    if (full) { half(std::true_type{}, H0); half(std::true_type{}, H1); }
    else { half(std::false_type{}, H0); half(std::false_type{}, H1); }
#ifdef W2_PROF
    pf_t2 = clock64(); pf_ep1 += pf_t2 - pf_t1;
#endif
  } else {
  // ---- output transform (lane-local) Y = A^T M A, A^T = [[1,1,0],[0,1,1]], one 32-tile half of the wave at a time
  // through the wave's private 16 KiB of LDS (16-byte global stores), BatchNorm statistics from the same registers
  const int co = nb * 64 + wn * 32 + li;
  const float bv = (a.bias != nullptr && co < a.Cout) ? a.bias[co] : 0.f;
  float ssum = 0.f, ssq = 0.f;              // ow: [pixel = tile*4 + 2a + b][32 channels]
  const bool has_stats = a.stats != nullptr;
  const int oy0 = Y0, ox0 = X0;
  const int c4 = lane & 7;
  const int cbase = nb * 64 + wn * 32 + c4 * 4;
  const bool vec_ok = (a.Cout & 3) == 0;
#pragma unroll
  for (int mi = 0; mi < 2; ++mi) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int tloc = (r & 3) + 8 * (r >> 2) + 4 * lh;       // tile within this half's 32
      const int tl = wm * 64 + mi * 32 + tloc;                // tile within the block: row tl / 16, column tl % 16
      float m[9];
#pragma unroll
      for (int xi = 0; xi < 8; ++xi) m[xi] = acc_elem(acc[xi][mi][r]);
      m[8] = acc[8][mi][r];
      float y00 = (m[0] + m[1]) + (m[3] + m[4]) + bv, y01 = (m[1] + m[2]) + (m[4] + m[5]) + bv;
      float y10 = (m[3] + m[4]) + (m[6] + m[7]) + bv, y11 = (m[4] + m[5]) + (m[7] + m[8]) + bv;
      if constexpr (ACT) {                  // 0 <= slope <= 1 (checked on the host): lrelu(y) = max(y, slope y)
        y00 = fmaxf(y00, y00 * a.out_slope); y01 = fmaxf(y01, y01 * a.out_slope);
        y10 = fmaxf(y10, y10 * a.out_slope); y11 = fmaxf(y11, y11 * a.out_slope);
      }
      float* op = ow + tloc * 128 + li;
      op[0] = y00; op[32] = y01; op[64] = y10; op[96] = y11;
      if (has_stats) {
        const int oy = oy0 + 2 * (tl >> 4), ox = ox0 + 2 * (tl & 15);
        if (co < a.Cout && oy < a.Ho && ox < a.Wo) {           // statistics over the outputs that exist
          const bool vx = ox + 1 < a.Wo, vy = oy + 1 < a.Ho;
          ssum += y00; ssq += y00 * y00;
          if (vx) { ssum += y01; ssq += y01 * y01; }
          if (vy) { ssum += y10; ssq += y10 * y10; }
          if (vx && vy) { ssum += y11; ssq += y11 * y11; }
        }
      }
      __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int it = 0; it < 16; ++it) {
      const int p = it * 8 + (lane >> 3);                     // pixel of this half: tile*4 + 2a + b
      const int tloc = p >> 2, ab = p & 3;
      const int tl = wm * 64 + mi * 32 + tloc;
      const int oy = oy0 + 2 * (tl >> 4) + (ab >> 1), ox = ox0 + 2 * (tl & 15) + (ab & 1);
      const f32x4 v = *(const f32x4*)(ow + p * 32 + c4 * 4);
      if (oy < a.Ho && ox < a.Wo) {
        float* yp = a.Y + (((long long)b * a.Ho + oy) * a.Wo + ox) * a.Cout + cbase;
        if (vec_ok && cbase + 3 < a.Cout) CY_NT_STORE(v, (f32x4*)yp);
        else {
#pragma unroll
          for (int k = 0; k < 4; ++k) if (cbase + k < a.Cout) yp[k] = v[k];
        }
      }
    }
    __builtin_amdgcn_sched_barrier(0);
  }
  if (has_stats) {
    ssum += __shfl_xor(ssum, 32, 64);
    ssq += __shfl_xor(ssq, 32, 64);
    if (lh == 0) {
      red[(wm * 64 + wn * 32 + li) * 2 + 0] = ssum;
      red[(wm * 64 + wn * 32 + li) * 2 + 1] = ssq;
    }
    __syncthreads();
    if (t < 64 && nb * 64 + t < a.Cout) {
      double* st = a.stats + (size_t)(blockIdx.x % CY_STATS_COPIES) * a.Cout * 2;
      atomicAdd(st + 2 * (nb * 64 + t), (double)red[t * 2] + (double)red[(64 + t) * 2]);
      atomicAdd(st + 2 * (nb * 64 + t) + 1, (double)red[t * 2 + 1] + (double)red[(64 + t) * 2 + 1]);
    }
  }
  }
  __syncthreads();                          // scratch and `red` are rewritten by the next position's T / S_U / S_raw
#ifdef W2_PROF
  if (MODE == 1) pf_ep2 = clock64() - pf_t2;
  else pf_ep1 += clock64() - pf_t1;
#endif
  }
  if (MODE == 1 && a.bn_red != nullptr && bn_nb >= 0) flush_bn(bn_nb, Vs);
#ifdef W2_PROF
  const void* pfp = MODE == 1 ? (const void*)a.bias : (const void*)a.bn_red;
  if (pfp != nullptr && t == 0) {
    long long* pf = (long long*)pfp + blockIdx.x * 4;
    pf[0] = pf_loop; pf[1] = pf_ep1; pf[2] = pf_ep2; pf[3] = ntile_mine | (pf_nfast << 32);
  }
#endif
}

// U[chunk f = (py*2+px) * Cin/8 + c/8][xi = i*3+j][kq][co (Np)][e] = (G g' G^T)[i][j] for c = (f % (Cin/8)) * 8 + kq*4 + e,
// g'[a][b] = W[co][c][2a+py][2b+px], G = [[1,0],[1,1],[0,1]]
__global__ void wino2_pack_kernel(const float* __restrict__ W, float* __restrict__ U, int Cout, int Cin, int Np, long long total) {
  const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= total) return;
  const int e = (int)(idx & 3);
  long long r = idx >> 2;
  const int co = (int)(r % Np); r /= Np;
  const int kq = (int)(r & 1); r >>= 1;
  const int xi = (int)(r % 9); r /= 9;
  const int cpp = Cin / 8;
  const int f = (int)r, ph = f / cpp, c = (f - ph * cpp) * 8 + kq * 4 + e;
  const int py = ph >> 1, px = ph & 1;
  float u = 0.f;
  if (co < Cout) {
    float g[2][2];
#pragma unroll
    for (int aa = 0; aa < 2; ++aa)
#pragma unroll
      for (int bb = 0; bb < 2; ++bb) g[aa][bb] = W[(((long long)co * Cin + c) * 4 + (2 * aa + py)) * 4 + (2 * bb + px)];
    const int i = xi / 3, j = xi % 3;
    const float r0 = i == 0 ? g[0][0] : i == 1 ? g[0][0] + g[1][0] : g[1][0];
    const float r1 = i == 0 ? g[0][1] : i == 1 ? g[0][1] + g[1][1] : g[1][1];
    u = j == 0 ? r0 : j == 1 ? r0 + r1 : r1;
  }
  U[idx] = u;
}

// ================================================================================================ weight gradient
// dW of the same layers through F(2x2, 2x2): with z the 2x2 tile of dY and d the 3x3 patch of X',
//   dg'_tile = A^T [(G z G^T) (.) (B^T d B)] A     (same B, G, A as the forward; 9 instead of 16 multiplies),
// and the sum over tiles and images commutes with A^T . A, so the kernel accumulates 9 GEMMs
//   dU_xi[q][co] = sum_tiles V_xi[tile][q] Z_xi[tile][co]        (tiles = MFMA k dimension)
// and wino2_wgrad_finish_kernel adds the per-image partial sums in a fixed order, applies A^T . A and scatters g' back to
// W[co][c][2a+py][2b+px].  One block = 128 q x 64 co of ONE image, 4 waves (2 x 2), wave = 64 q x 32 co x 9 positions
// (the forward kernel's MFMA / fragment / slot machinery unchanged); a chunk is 8 tiles (2 x 4): its 5x9 patch of X'
// (512-byte rows) and 4x8 patch of dY are staged raw in LDS, every thread transforms two tile pairs of one q channel and
// one tile pair of one output channel as float2 (paired ds_read2st64_b32 of columns c and c + 2, winograd.hip).
constexpr int WQ = 128, XPW = 45, XPWS = 48, ZPW = 32;   // q channels per block; X' / dY patch pixels (X' rows allocated: 48)
constexpr int RAWG_BUF = XPWS * WQ + ZPW * 64;           // floats, single buffer

struct Wino2WgradArgs {
  const float* X; const float* dZ; float* slab;
  const float* in_scale; const float* in_shift; float in_slope;   // optional: the layer input is lrelu(X * scale[c] + shift[c])
  int B, H, W, Cin, Cout, Ho, Wo, gh, gw;
  int nsplit;                               // tile groups of one image are cut into nsplit ranges (more blocks for small layers)
};

template <int O0, int O1>
__device__ __forceinline__ f32x2 lds_pair_st64(unsigned addr) {   // (dword[O0*64], dword[O1*64]) as one register pair
  f32x2 r;
  asm volatile("ds_read2st64_b32 %0, %1 offset0:%2 offset1:%3" : "=v"(r) : "v"(addr), "n"(O0), "n"(O1));
  return r;
}
__device__ __forceinline__ f32x2 pk_add(f32x2 x, f32x2 y) {
  f32x2 r;
  asm("v_pk_add_f32 %0, %1, %2" : "=v"(r) : "v"(x), "v"(y));
  return r;
}

// side work of one chunk (72 slots, fragment fetches in slots 0..2 of every position):
//   4 T   two paired reads of the raw patches of chunk c+1: dY first (2 pieces), then X' (9 pieces)
//   5 Z   one row of G z G^T (3 pieces)      6 V  one row of B^T d B of one of the two items (6 pieces)
//   8     barrier: every wave is past its raw-patch reads
//   7 S   one float4 of chunk c+2 registers -> raw LDS (8 pieces)      2 G  one global load of chunk c+3 (8 pieces)
constexpr int G2_T[11] = {3, 4, 5, 6, 7, 11, 12, 13, 14, 15, 19};
constexpr int G2_Z[3] = {20, 21, 22};
constexpr int G2_V[6] = {23, 27, 28, 29, 30, 31};
constexpr int G2_BAR = 35;
constexpr int G2_S[8] = {36, 37, 38, 39, 43, 44, 45, 46};
constexpr int G2_G[8] = {47, 51, 52, 53, 54, 55, 59, 60};
constexpr int g2_kind(int s) {
  return s2_find(G2_T, 11, s) >= 0 ? 4 : s2_find(G2_Z, 3, s) >= 0 ? 5 : s2_find(G2_V, 6, s) >= 0 ? 6 : s == G2_BAR ? 8
       : s2_find(G2_S, 8, s) >= 0 ? 7 : s2_find(G2_G, 8, s) >= 0 ? 2 : 0;
}
constexpr int g2_idx(int s) {
  const int k = g2_kind(s);
  return k == 4 ? s2_find(G2_T, 11, s) : k == 5 ? s2_find(G2_Z, 3, s) : k == 6 ? s2_find(G2_V, 6, s)
       : k == 7 ? s2_find(G2_S, 8, s) : k == 2 ? s2_find(G2_G, 8, s) : 0;
}
constexpr int g2_side_lds(int s) {          // exact: every LDS instruction of the side work is unconditional
  const int k = g2_kind(s);
  return k == 4 ? 2 : k == 5 ? 3 : k == 6 ? 3 : k == 7 ? 1 : 0;
}
constexpr int g2_younger(int xi) {          // LDS operations younger than position xi's last fragment at its first MFMA
  if (xi == 0) return 0;
  int n = g2_side_lds(8 * (xi - 1) + 2);
  for (int s = 8 * (xi - 1) + 3; s < 8 * xi; ++s) n += s2_frag_lds(s) + g2_side_lds(s);
  return n > 14 ? 14 : n;
}
// LDS operations issued after the side work of slot `from` and before the side work of slot `to`
constexpr int g2_between(int from, int to) {
  int n = 0;
  for (int s = from + 1; s < to; ++s) n += s2_frag_lds(s) + g2_side_lds(s);
  n += s2_frag_lds(to);
  return n > 14 ? 14 : n;
}

template <bool AFFINE>
__global__ __launch_bounds__(256, 1) void wino2_wgrad_kernel(Wino2WgradArgs a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* Vs = smem;                         // [2][V2_BUF]   A images (X', rows = q)
  float* Zs = smem + 2 * V2_BUF;            // [2][U2_BUF]   B images (dY, rows = co)
  float* Rw = smem + 2 * V2_BUF + 2 * U2_BUF;   // raw X' [XPWS][128] then raw dY [32][64]

  const int t = threadIdx.x, lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int wm = wave & 1, wn = wave >> 1;
  const int li = lane & 31, lh = lane >> 5;
  const int nqb = 4 * a.Cin / WQ, ncb = a.Cout / 64;
  int vid = blockIdx.x;                     // XCD-aware ids: the blocks of one image share an L2
  if ((gridDim.x & 7) == 0) vid = (blockIdx.x & 7) * (gridDim.x >> 3) + (blockIdx.x >> 3);
  const int cob = vid % ncb;
  const int qb = (vid / ncb) % nqb;
  const int bs = vid / (ncb * nqb);         // image * nsplit + split
  const int b = bs / a.nsplit, sp = bs - b * a.nsplit;
  const int ngroups = a.gh * a.gw;
  const int cbeg = (int)((long long)ngroups * sp / a.nsplit), cend = (int)((long long)ngroups * (sp + 1) / a.nsplit);
  const int nchunk = cend - cbeg;           // >= 1 (nsplit <= ngroups)

  // ---- loaders: X' item = (pixel = (t >> 5) + 8 q, float4 column t & 31), dY item = (pixel = (t >> 4) + 16 q, t & 15)
  const int xc4 = t & 31, xprow = t >> 5, zc4 = t & 15, zprow = t >> 4;
  const int qch = qb * WQ + xc4 * 4;        // this thread's q channel quad: one (py, px) and 4 consecutive c
  const int qph = qch / a.Cin, qc = qch - qph * a.Cin, qpy = qph >> 1, qpx = qph & 1;
  const char* ximg = (const char*)(a.X + (long long)b * a.H * a.W * a.Cin);
  const char* zimg = (const char*)(a.dZ + (long long)b * a.Ho * a.Wo * a.Cout + cob * 64);
  int xpr[6], xpc[6];
  unsigned voffx[6], voffz[2];              // fast path: byte offsets from the chunk's first pixel
#pragma unroll
  for (int q = 0; q < 6; ++q) {
    const int pix = xprow + 8 * q;
    xpr[q] = pix / 9; xpc[q] = pix - xpr[q] * 9;
    voffx[q] = pix < XPW ? (unsigned)((((2 * xpr[q] + qpy) * a.W + 2 * xpc[q] + qpx) * a.Cin + qc) * 4) : (unsigned)((((qpy) * a.W + qpx) * a.Cin + qc) * 4);
  }
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    const int pix = zprow + 16 * q;
    voffz[q] = (unsigned)((((pix >> 3) * a.Wo + (pix & 7)) * a.Cout + zc4 * 4) * 4);
  }
  f32x4 gx[6], gz[2];
  unsigned okm = 0;                         // slow path: bit q: gx[q] in range, bit 6+q: gz[q]
  bool gfast = false;                       // the chunk held in gx/gz was loaded by the fast path
  auto is_fast = [&](int gy, int gxx) {     // all of the chunk's X' pixels (any py, px) and dY pixels are inside
    return gy > 0 && gxx > 0 && 8 * gy + 8 <= a.H - 1 && 16 * gxx + 16 <= a.W - 1 && 4 * gy + 4 <= a.Ho && 8 * gxx + 8 <= a.Wo;
  };
  auto Gx = [&](int q, int gy, int gxx, bool fast) {
    if (fast) {
      gx[q] = *(const f32x4*)(ximg + (size_t)(((8 * gy - 1) * a.W + (16 * gxx - 1)) * a.Cin) * 4 + voffx[q]);
    } else {
      const int iy = 2 * (4 * gy + xpr[q]) - 1 + qpy, ix = 2 * (8 * gxx + xpc[q]) - 1 + qpx;
      const bool ok = (xprow + 8 * q) < XPW && (unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W;
      gx[q] = *(const f32x4*)(ximg + (ok ? (unsigned)(((iy * a.W + ix) * a.Cin + qc) * 4) : 0u));
      okm = (okm & ~(1u << q)) | ((unsigned)ok << q);
    }
  };
  auto Gz = [&](int q, int gy, int gxx, bool fast) {
    if (fast) {
      gz[q] = *(const f32x4*)(zimg + (size_t)((4 * gy * a.Wo + 8 * gxx) * a.Cout) * 4 + voffz[q]);
    } else {
      const int pix = zprow + 16 * q;
      const int oy = 4 * gy + (pix >> 3), ox = 8 * gxx + (pix & 7);
      const bool ok = oy < a.Ho && ox < a.Wo;
      gz[q] = *(const f32x4*)(zimg + (ok ? (unsigned)(((oy * a.Wo + ox) * a.Cout + zc4 * 4) * 4) : 0u));
      okm = (okm & ~(64u << q)) | ((unsigned)ok << (6 + q));
    }
  };
  constexpr bool affine = AFFINE;           // this thread's 4 input channels never change
  const f32x4 asc = affine ? *(const f32x4*)(a.in_scale + qc) : f32x4{1.f, 1.f, 1.f, 1.f};
  const f32x4 ash = affine ? *(const f32x4*)(a.in_shift + qc) : f32x4{0.f, 0.f, 0.f, 0.f};
  auto Sx = [&](int q) {
    float* dst = Rw + (xprow + 8 * q) * WQ + xc4 * 4;
    const f32x4 v = affine ? affine_lrelu4(gx[q], asc, ash, a.in_slope) : gx[q];
    if (gfast) *(f32x4*)dst = v;
    else *(f32x4*)dst = (okm >> q) & 1 ? v : f32x4{0.f, 0.f, 0.f, 0.f};    // padding stays exactly 0
  };
  auto Sz = [&](int q) {
    float* dst = Rw + XPWS * WQ + (zprow + 16 * q) * 64 + zc4 * 4;
    if (gfast) *(f32x4*)dst = gz[q];
    else *(f32x4*)dst = (okm >> (6 + q)) & 1 ? gz[q] : f32x4{0.f, 0.f, 0.f, 0.f};
  };
  auto Gall = [&](int c) {
    const int gy = (cbeg + c) / a.gw, gxx = (cbeg + c) - gy * a.gw;
    const bool fast = is_fast(gy, gxx);
#pragma unroll
    for (int q = 0; q < 6; ++q) Gx(q, gy, gxx, fast);
#pragma unroll
    for (int q = 0; q < 2; ++q) Gz(q, gy, gxx, fast);
    gfast = fast;
  };
  auto Sall = [&]() {
#pragma unroll
    for (int q = 0; q < 6; ++q) Sx(q);
#pragma unroll
    for (int q = 0; q < 2; ++q) Sz(q);
  };

  // ---- transform items.  V: q channel vc = t & 127, tile row vr = t >> 7, tile pairs 0 and 1 (two items);
  //      Z: output channel zc = t & 63, tile row zr = (t >> 6) & 1, tile pair zp = t >> 7.   .x / .y = the pair's two tiles
  const int vc = t & 127, vr = t >> 7;
  const int zc = t & 63, zr = (t >> 6) & 1, zp = t >> 7;
  const float* xraw = Rw + ((2 * vr) * 9) * WQ + vc;                      // X' pixel (2 vr + r, 4 tp + c) -> + ((r*9 + 4tp + c) * 128)
  const float* zraw = Rw + XPWS * WQ + ((2 * zr) * 8 + 4 * zp) * 64 + zc;  // dY pixel (2 zr + r, 4 zp + c) -> + (r*8 + c) * 64
  const unsigned xr_a = (unsigned)(uintptr_t)(__attribute__((address_space(3))) const float*)xraw;
  const unsigned zr_a = (unsigned)(uintptr_t)(__attribute__((address_space(3))) const float*)zraw;
  const int vdst = vr * SLABV + vc * 4;     // + 2 tp + xi * 2 * SLABV
  const int zdst = zr * SLABU + zc * 4 + 2 * zp;
  f32x2 xv[2][3][3], zv[2][2];
  auto Vrow = [&](float* vb, int it, int R) {   // row R of B^T d B of item (tile pair) it
    f32x2 t0[3];
#pragma unroll
    for (int cc = 0; cc < 3; ++cc)
      t0[cc] = R == 0 ? pk_sub(xv[it][0][cc], xv[it][1][cc]) : R == 1 ? xv[it][1][cc] : pk_sub(xv[it][2][cc], xv[it][1][cc]);
    float* v = vb + 2 * it;
    *(f32x2*)(v + (R * 3 + 0) * 2 * SLABV) = pk_sub(t0[0], t0[1]);
    *(f32x2*)(v + (R * 3 + 1) * 2 * SLABV) = t0[1];
    *(f32x2*)(v + (R * 3 + 2) * 2 * SLABV) = pk_sub(t0[2], t0[1]);
  };
  auto Zrow = [&](float* zb, int R) {       // row R of G z G^T
    const f32x2 u0 = R == 0 ? zv[0][0] : R == 1 ? pk_add(zv[0][0], zv[1][0]) : zv[1][0];
    const f32x2 u1 = R == 0 ? zv[0][1] : R == 1 ? pk_add(zv[0][1], zv[1][1]) : zv[1][1];
    *(f32x2*)(zb + (R * 3 + 0) * 2 * SLABU) = u0;
    *(f32x2*)(zb + (R * 3 + 1) * 2 * SLABU) = pk_add(u0, u1);
    *(f32x2*)(zb + (R * 3 + 2) * 2 * SLABU) = u1;
  };
  auto Tall_plain = [&]() {                 // prologue only: plain loads (the compiler waits for them itself)
#pragma unroll
    for (int it = 0; it < 2; ++it)
#pragma unroll
      for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int cc = 0; cc < 3; ++cc)
          xv[it][r][cc] = f32x2{xraw[(r * 9 + 4 * it + cc) * WQ], xraw[(r * 9 + 4 * it + cc + 2) * WQ]};
#pragma unroll
    for (int r = 0; r < 2; ++r)
#pragma unroll
      for (int cc = 0; cc < 2; ++cc) zv[r][cc] = f32x2{zraw[(r * 8 + cc) * 64], zraw[(r * 8 + cc + 2) * 64]};
  };

  f32x16 acc[9][2];
#pragma unroll
  for (int xi = 0; xi < 9; ++xi)
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[xi][mi][r] = 0.f;

  // ---- prologue: chunk 0 transformed into buffer 0, chunk 1 raw in LDS, chunk 2 in registers
  Gall(0);
  Sall();
  __syncthreads();
  Tall_plain();
#pragma unroll
  for (int R = 0; R < 3; ++R) { Vrow(Vs + vdst, 0, R); Vrow(Vs + vdst, 1, R); Zrow(Zs + zdst, R); }
  Gall(nchunk > 1 ? 1 : 0);
  __syncthreads();
  Sall();
  Gall(nchunk > 2 ? 2 : nchunk - 1);
  __syncthreads();

  const int fragA = lh * SLABV + (wm * 64 + li) * 4;
  const int fragB = lh * SLABU + (wn * 32 + li) * 4;
  int ggy, ggx;                             // group of the chunk whose loads are in flight: chunk min(c + 3, nchunk - 1)
  {
    const int cg = cbeg + (nchunk > 2 ? 2 : nchunk - 1);
    ggy = cg / a.gw; ggx = cg - ggy * a.gw;
  }
  for (int c = 0; c < nchunk; ++c) {
    const float* vb_ = Vs + (c & 1) * V2_BUF + fragA;
    const float* ub_ = Zs + (c & 1) * U2_BUF + fragB;
    float* vw_ = Vs + ((c + 1) & 1) * V2_BUF + vdst;                // transforms of chunk c+1 (harmless after the last chunk)
    float* zw_ = Zs + ((c + 1) & 1) * U2_BUF + zdst;
    if (c + 3 < nchunk) {                                           // uniform; the tail re-loads the last chunk
      const bool wx = ggx + 1 == a.gw;
      ggx = wx ? 0 : ggx + 1;
      ggy += wx ? 1 : 0;
    }
    // No branch in the MFMA slots (winograd.hip, weight gradient): chunks inside the image load `uniform origin + lane
    // offset`, border chunks load a harmless valid address there and are re-loaded with bounds after the stream; LDS
    // stores are unconditional and a border chunk's padding is zeroed in LDS after the stream.
    const bool gf_next = is_fast(ggy, ggx);
    const char* xb_ = ximg + (gf_next ? (size_t)(((8 * ggy - 1) * a.W + (16 * ggx - 1)) * a.Cin) * 4 : (size_t)0);
    const char* zb_ = zimg + (gf_next ? (size_t)((4 * ggy * a.Wo + 8 * ggx) * a.Cout) * 4 : (size_t)0);
    f32x4 fa_[2][2], fb_[2];
    fa_[0][0] = *(const f32x4*)(vb_);
    fa_[0][1] = *(const f32x4*)(vb_ + 128);
    fb_[0] = *(const f32x4*)(ub_);
#define G2SLOT(SIDX)                                                                                \
    {                                                                                               \
      constexpr int sidx = (SIDX);                                                                  \
      constexpr int xi = sidx >> 3, w_ = sidx & 7, mi = w_ & 1, e = w_ >> 1;                        \
      if (w_ == 0) __builtin_amdgcn_s_waitcnt(0xC07F | (g2_younger(xi) << 8));                      \
      if (xi < 8) mfma_a(acc[xi][mi], fa_[xi & 1][mi][e], fb_[xi & 1][e]);                          \
      else mfma_v(acc[xi][mi], fa_[xi & 1][mi][e], fb_[xi & 1][e]);                                 \
      if (w_ < 3 && xi + 1 < 9) {                                                                   \
        constexpr int nx = (xi + 1 < 9) ? xi + 1 : 0;                                               \
        if (w_ == 0) fa_[nx & 1][0] = *(const f32x4*)(vb_ + nx * 2 * SLABV);                        \
        if (w_ == 1) fa_[nx & 1][1] = *(const f32x4*)(vb_ + nx * 2 * SLABV + 128);                  \
        if (w_ == 2) fb_[nx & 1] = *(const f32x4*)(ub_ + nx * 2 * SLABU);                           \
      }                                                                                             \
      constexpr int kind = g2_kind(sidx), k_ = g2_idx(sidx) < 0 ? 0 : g2_idx(sidx);                 \
      if (kind == 4) {                      /* raw patches of chunk c+1: two pairs */              \
        if (k_ < 2) {                                                                               \
          constexpr int r = k_ & 1;                                                                 \
          zv[r][0] = lds_pair_st64<r * 8 + 0, r * 8 + 2>(zr_a);                                     \
          zv[r][1] = lds_pair_st64<r * 8 + 1, r * 8 + 3>(zr_a);                                     \
        } else {                                                                                    \
          constexpr int i0 = 2 * ((k_ < 2 ? 0 : k_ - 2) % 9), i1 = i0 + 1;   /* 18 pairs: (item, row, col) */ \
          constexpr int p0 = ((i0 % 9) / 3) * 9 + 4 * (i0 / 9) + i0 % 3, p1 = ((i1 % 9) / 3) * 9 + 4 * (i1 / 9) + i1 % 3; \
          xv[i0 / 9][(i0 % 9) / 3][i0 % 3] = lds_pair_st64<2 * p0, 2 * p0 + 4>(xr_a);               \
          xv[i1 / 9][(i1 % 9) / 3][i1 % 3] = lds_pair_st64<2 * p1, 2 * p1 + 4>(xr_a);               \
        }                                                                                           \
      } else if (kind == 5) {               /* the dY reads (slots 3, 4) are >= 14 LDS operations old */ \
        if (k_ == 0) __builtin_amdgcn_s_waitcnt(0xC07F | (g2_between(G2_T[1], G2_Z[0]) << 8));      \
        Zrow(zw_, k_ % 3);                                                                          \
      } else if (kind == 6) {               /* all X' reads are older than the Z rows' 9 stores */ \
        if (k_ == 0) __builtin_amdgcn_s_waitcnt(0xC07F | (g2_between(G2_T[10], G2_V[0]) << 8));     \
        Vrow(vw_, (k_ % 6) / 3, (k_ % 6) % 3);                                                      \
      } else if (kind == 8) {               /* all waves are past their raw-patch reads */         \
        __builtin_amdgcn_s_barrier();                                                               \
      } else if (kind == 7) {               /* chunk c+2: one float4 of registers -> raw LDS */    \
        if (k_ < 6) *(f32x4*)(Rw + (xprow + 8 * (k_ % 6)) * WQ + xc4 * 4) =                         \
            affine ? affine_lrelu4(gx[k_ % 6], asc, ash, a.in_slope) : gx[k_ % 6];                  \
        else *(f32x4*)(Rw + XPWS * WQ + (zprow + 16 * (k_ & 1)) * 64 + zc4 * 4) = gz[k_ & 1];       \
      } else if (kind == 2) {               /* one global load of chunk c+3 */                     \
        if (k_ < 6) gx[k_ % 6] = *(const f32x4*)(xb_ + (gf_next ? voffx[k_ % 6] : (unsigned)(qc * 4)));      \
        else gz[k_ & 1] = *(const f32x4*)(zb_ + (gf_next ? voffz[k_ & 1] : (unsigned)(zc4 * 16)));  \
      }                                                                                             \
      __builtin_amdgcn_sched_barrier(0);                                                            \
    }
#define G2SLOT8(B) G2SLOT((B)) G2SLOT((B) + 1) G2SLOT((B) + 2) G2SLOT((B) + 3) G2SLOT((B) + 4) G2SLOT((B) + 5) G2SLOT((B) + 6) G2SLOT((B) + 7)
    G2SLOT8(0) G2SLOT8(8) G2SLOT8(16) G2SLOT8(24) G2SLOT8(32) G2SLOT8(40) G2SLOT8(48) G2SLOT8(56) G2SLOT8(64)
#undef G2SLOT8
#undef G2SLOT
    if (!(gfast && gf_next)) {              // uniform, border chunks only
      if (!gfast) {                         // the patch just stored: zero its padding (okm of its load)
#pragma unroll
        for (int q = 0; q < 6; ++q)
          if (!((okm >> q) & 1)) *(f32x4*)(Rw + (xprow + 8 * q) * WQ + xc4 * 4) = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int q = 0; q < 2; ++q)
          if (!((okm >> (6 + q)) & 1)) *(f32x4*)(Rw + XPWS * WQ + (zprow + 16 * q) * 64 + zc4 * 4) = f32x4{0.f, 0.f, 0.f, 0.f};
      }
      if (!gf_next) {                       // the patch just requested: load it again with bounds
#pragma unroll
        for (int q = 0; q < 6; ++q) Gx(q, ggy, ggx, false);
#pragma unroll
        for (int q = 0; q < 2; ++q) Gz(q, ggy, ggx, false);
      }
    }
    gfast = gf_next;
    __syncthreads();
  }

  // ---- per-image partial dU[xi][q][co] -> slab[b]
  const int Q = 4 * a.Cin;
  float* out = a.slab + ((long long)bs * 9) * Q * a.Cout;
  const int co = cob * 64 + wn * 32 + li;
#pragma unroll
  for (int xi = 0; xi < 9; ++xi)
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int q = qb * WQ + wm * 64 + mi * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
        const float v = xi < 8 ? acc_elem(acc[xi][mi][r]) : acc[8][mi][r];
        out[((long long)xi * Q + q) * a.Cout + co] = v;
      }
}

// dW[co][c][2a+py][2b+px] = (A^T m A)[a][b], m[xi] = sum_b slab[b][xi][q = (py*2+px)*Cin + c][co], A^T = [[1,1,0],[0,1,1]]
__global__ void wino2_wgrad_finish_kernel(const float* __restrict__ slab, float* __restrict__ dW, int nb, int Cin, int Cout) {
  const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const int Q = 4 * Cin;
  if (idx >= (long long)Q * Cout) return;
  const int co = (int)(idx % Cout), q = (int)(idx / Cout);
  float m[9];
#pragma unroll
  for (int xi = 0; xi < 9; ++xi) m[xi] = 0.f;
  for (int bb = 0; bb < nb; ++bb)
#pragma unroll
    for (int xi = 0; xi < 9; ++xi) m[xi] += slab[(((long long)bb * 9 + xi) * Q + q) * Cout + co];
  const int ph = q / Cin, c = q - ph * Cin, py = ph >> 1, px = ph & 1;
  float* o = dW + ((long long)co * Cin + c) * 16;
  o[(0 + py) * 4 + 0 + px] = (m[0] + m[1]) + (m[3] + m[4]);
  o[(0 + py) * 4 + 2 + px] = (m[1] + m[2]) + (m[4] + m[5]);
  o[(2 + py) * 4 + 0 + px] = (m[3] + m[4]) + (m[6] + m[7]);
  o[(2 + py) * 4 + 2 + px] = (m[4] + m[5]) + (m[7] + m[8]);
}

// input-gradient weights: U[chunk f = co/8][xi = i*3+j][kq][q (Np = 4 Cin)][e] = (G h G^T)[i][j] for co = f*8 + kq*4 + e,
// h[a'][b'] = W[co][c][2(1-a')+py][2(1-b')+px] with q = (py*2+px) * Cin + c
__global__ void wino2_pack_dgrad_kernel(const float* __restrict__ W, float* __restrict__ U, int Cout, int Cin, long long total) {
  const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= total) return;
  const int Np = 4 * Cin;
  const int e = (int)(idx & 3);
  long long r = idx >> 2;
  const int q = (int)(r % Np); r /= Np;
  const int kq = (int)(r & 1); r >>= 1;
  const int xi = (int)(r % 9); r /= 9;
  const int co = (int)r * 8 + kq * 4 + e;
  const int ph = q / Cin, c = q - ph * Cin, py = ph >> 1, px = ph & 1;
  float u = 0.f;
  if (co < Cout) {
    float h[2][2];
#pragma unroll
    for (int aa = 0; aa < 2; ++aa)
#pragma unroll
      for (int bb = 0; bb < 2; ++bb) h[aa][bb] = W[(((long long)co * Cin + c) * 4 + (2 * (1 - aa) + py)) * 4 + (2 * (1 - bb) + px)];
    const int i = xi / 3, j = xi % 3;
    const float r0 = i == 0 ? h[0][0] : i == 1 ? h[0][0] + h[1][0] : h[1][0];
    const float r1 = i == 0 ? h[0][1] : i == 1 ? h[0][1] + h[1][1] : h[1][1];
    u = j == 0 ? r0 : j == 1 ? r0 + r1 : r1;
  }
  U[idx] = u;
}

}  // namespace

extern "C" long long cy_wino2_packed_floats(int Cin, int N) {
  return (long long)(4 * (Cin / 8)) * 18 * ((N + 63) / 64 * 64) * 4;
}

extern "C" int cy_wino2_pack_weights(const float* W, float* U, int Cout, int Cin, void* stream) {
  CY_REQUIRE(W && U && Cout > 0 && Cin > 0 && Cin % 8 == 0, "cy_wino2_pack_weights: bad arguments (Cin %% 8 == 0)");
  const int Np = (Cout + 63) / 64 * 64;
  const long long total = cy_wino2_packed_floats(Cin, Cout);
  wino2_pack_kernel<<<(unsigned)cy_ceil_div(total, 256), 256, 0, (hipStream_t)stream>>>(W, U, Cout, Cin, Np, total);
  CY_LAUNCH_CHECK("cy_wino2_pack_weights");
  return 0;
}

// persistent grid: one block per CU (153 KB of LDS, 512 registers per lane)
static int wino2_persistent_blocks(long long tiles, long long* blocks, const char* who) {
  int dev = 0, ncu = 0;
  hipError_t he = hipGetDevice(&dev);
  if (he == hipSuccess) he = hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev);
  if (he != hipSuccess || ncu <= 0) return cy_set_error((int)he, "%s: cannot query the CU count: %s", who, hipGetErrorString(he));
  *blocks = tiles < ncu ? tiles : ncu;
  return 0;
}

extern "C" int cy_conv4x4s2_winograd(const float* X, const float* U, float* Y, const float* bias, double* stats,
                                     const float* in_scale, const float* in_shift, float in_slope, float out_slope, int B, int H,
                                     int W, int Cin, int Cout, void* stream) {
  CY_REQUIRE(out_slope >= 0.f && out_slope <= 1.f, "cy_conv4x4s2_winograd: out_slope=%g must be in [0, 1] (1 = no activation)", (double)out_slope);
  CY_REQUIRE(out_slope == 1.f || (stats == nullptr && in_scale == nullptr),
             "cy_conv4x4s2_winograd: the activation epilogue is for eval-mode forwards (no statistics, no fused input affine)");
  CY_REQUIRE((in_scale == nullptr) == (in_shift == nullptr), "cy_conv4x4s2_winograd: in_scale and in_shift go together");
  CY_REQUIRE(in_scale == nullptr || (in_slope > 0.f && in_slope <= 1.f), "cy_conv4x4s2_winograd: in_slope must be in (0, 1]");
  CY_REQUIRE(X && U && Y && B > 0 && H > 0 && W > 0 && Cout > 0, "cy_conv4x4s2_winograd: bad arguments");
  CY_REQUIRE(Cin % 8 == 0 && Cin >= 8, "cy_conv4x4s2_winograd: Cin=%d must be a multiple of 8", Cin);
  CY_REQUIRE(H % 2 == 0 && W % 2 == 0, "cy_conv4x4s2_winograd: H=%d, W=%d must be even", H, W);
  CY_REQUIRE((((uintptr_t)X | (uintptr_t)U) & 15) == 0, "cy_conv4x4s2_winograd: operands must be 16-byte aligned");
  CY_REQUIRE((long long)H * W * Cin < (1ll << 29), "cy_conv4x4s2_winograd: image too large for 32-bit offsets");
  Wino2Args a;
  a.X = X; a.U = U; a.Y = Y; a.bias = bias; a.stats = stats;
  a.in_scale = in_scale; a.in_shift = in_shift; a.in_slope = in_slope; a.out_slope = out_slope;
  a.B = B; a.H = H; a.W = W; a.Cin = Cin; a.Cout = Cout;
  a.Np = (Cout + 63) / 64 * 64;
  a.Ho = H / 2; a.Wo = W / 2;
  a.tbh = (a.Ho + 2 * TR2 - 1) / (2 * TR2); a.tbw = (a.Wo + 2 * TC2 - 1) / (2 * TC2);
  const long long tiles = (long long)B * a.tbh * a.tbw * (a.Np / 64);
  CY_REQUIRE(tiles < (1ll << 31), "cy_conv4x4s2_winograd: too many tiles");
  a.ntiles = (int)tiles;
  long long blocks = 0;
  int rcq = wino2_persistent_blocks(tiles, &blocks, "cy_conv4x4s2_winograd");
  if (rcq) return rcq;
  const size_t lds = (size_t)(2 * V2_BUF + 2 * U2_BUF + 2 * RAW2_BUF + (in_scale ? 2 * Cin : 0)) * 4;
  CY_REQUIRE(lds <= 160 * 1024, "cy_conv4x4s2_winograd: Cin=%d too large for the fused input affine", Cin);
  a.Hx = a.Wx = a.Cx = 0; a.bn_z = a.bn_scale = a.bn_shift = a.bn_mean = a.bn_invstd = nullptr; a.bn_red = nullptr; a.bn_slope = 0.f;
#ifdef W2_PROF
  if (const char* e = getenv("CY_W2_PROF")) a.bn_red = (double*)strtoull(e, nullptr, 0);
  if (const char* e = getenv("CY_W2_STAMPS")) a.bn_scale = (const float*)strtoull(e, nullptr, 0);
#endif
  int rc = in_scale ? cy_allow_lds(wino2_conv_kernel<0, true>, lds)
                    : out_slope != 1.f ? cy_allow_lds(wino2_conv_kernel<0, false, true>, lds) : cy_allow_lds(wino2_conv_kernel<0, false>, lds);
  if (rc) return rc;
  if (in_scale) wino2_conv_kernel<0, true><<<(unsigned)blocks, 256, lds, (hipStream_t)stream>>>(a);
  else if (out_slope != 1.f) wino2_conv_kernel<0, false, true><<<(unsigned)blocks, 256, lds, (hipStream_t)stream>>>(a);
  else wino2_conv_kernel<0, false><<<(unsigned)blocks, 256, lds, (hipStream_t)stream>>>(a);
  CY_LAUNCH_CHECK("cy_conv4x4s2_winograd");
  return 0;
}

// tile-group ranges per image: enough blocks to cover the chip (one block per CU) when B * 4Cin/128 * Cout/64 is small
static int wino2_wgrad_splits(int B, int Cin, int Cout) {
  const long long base = (long long)B * (4 * Cin / WQ) * (Cout / 64);
  int s = 1;
  while (base * s < 192 && s < 8) s *= 2;
  return s;
}
extern "C" long long cy_wino2_wgrad_ws_floats(int B, int Cin, int Cout) {
  return (long long)B * wino2_wgrad_splits(B, Cin, Cout) * 9 * 4 * Cin * Cout;
}

extern "C" int cy_conv4x4s2_winograd_wgrad(const float* X, const float* dZ, float* dW, float* ws, const float* in_scale,
                                           const float* in_shift, float in_slope, int B, int H, int W, int Cin,
                                           int Cout, void* stream) {
  CY_REQUIRE((in_scale == nullptr) == (in_shift == nullptr), "cy_conv4x4s2_winograd_wgrad: in_scale and in_shift go together");
  CY_REQUIRE(in_scale == nullptr || (in_slope > 0.f && in_slope <= 1.f), "cy_conv4x4s2_winograd_wgrad: in_slope must be in (0, 1]");
  CY_REQUIRE(X && dZ && dW && ws && B > 0 && H > 0 && W > 0, "cy_conv4x4s2_winograd_wgrad: bad arguments");
  CY_REQUIRE(Cin % 32 == 0 && Cout % 64 == 0, "cy_conv4x4s2_winograd_wgrad: Cin=%d must be a multiple of 32, Cout=%d of 64", Cin, Cout);
  CY_REQUIRE(H % 2 == 0 && W % 2 == 0, "cy_conv4x4s2_winograd_wgrad: H=%d, W=%d must be even", H, W);
  CY_REQUIRE((((uintptr_t)X | (uintptr_t)dZ) & 15) == 0, "cy_conv4x4s2_winograd_wgrad: operands must be 16-byte aligned");
  CY_REQUIRE((long long)H * W * Cin < (1ll << 29), "cy_conv4x4s2_winograd_wgrad: image too large for 32-bit offsets");
  Wino2WgradArgs a;
  a.X = X; a.dZ = dZ; a.slab = ws; a.B = B; a.H = H; a.W = W; a.Cin = Cin; a.Cout = Cout;
  a.in_scale = in_scale; a.in_shift = in_shift; a.in_slope = in_slope;
  a.Ho = H / 2; a.Wo = W / 2;
  a.gh = (a.Ho + 3) / 4; a.gw = (a.Wo + 7) / 8;
  a.nsplit = wino2_wgrad_splits(B, Cin, Cout);
  while (a.nsplit > a.gh * a.gw) a.nsplit >>= 1;         // (the workspace is sized for the larger count)
  const long long blocks = (long long)B * a.nsplit * (4 * Cin / WQ) * (Cout / 64);
  CY_REQUIRE(blocks < (1ll << 31), "cy_conv4x4s2_winograd_wgrad: grid too large");
  const size_t lds = (size_t)(2 * V2_BUF + 2 * U2_BUF + RAWG_BUF) * 4;
  int rc = in_scale ? cy_allow_lds(wino2_wgrad_kernel<true>, lds) : cy_allow_lds(wino2_wgrad_kernel<false>, lds);
  if (rc) return rc;
  hipStream_t s = (hipStream_t)stream;
  if (in_scale) wino2_wgrad_kernel<true><<<(unsigned)blocks, 256, lds, s>>>(a);
  else wino2_wgrad_kernel<false><<<(unsigned)blocks, 256, lds, s>>>(a);
  CY_LAUNCH_CHECK("cy_conv4x4s2_winograd_wgrad");
  const long long n = (long long)4 * Cin * Cout;
  wino2_wgrad_finish_kernel<<<(unsigned)cy_ceil_div(n, 256), 256, 0, s>>>(ws, dW, B * a.nsplit, Cin, Cout);
  CY_LAUNCH_CHECK("cy_conv4x4s2_winograd_wgrad(finish)");
  return 0;
}

/* ---- input gradient of the 4x4 / stride 2 / pad 1 layers */
extern "C" long long cy_wino2_dgrad_packed_floats(int Cin, int Cout) { return (long long)(Cout / 8) * 18 * (4 * Cin) * 4; }

extern "C" int cy_wino2_pack_dgrad_weights(const float* W, float* U, int Cout, int Cin, void* stream) {
  CY_REQUIRE(W && U && Cout > 0 && Cin > 0 && Cout % 8 == 0 && Cin % 16 == 0, "cy_wino2_pack_dgrad_weights: bad arguments");
  const long long total = cy_wino2_dgrad_packed_floats(Cin, Cout);
  wino2_pack_dgrad_kernel<<<(unsigned)cy_ceil_div(total, 256), 256, 0, (hipStream_t)stream>>>(W, U, Cout, Cin, total);
  CY_LAUNCH_CHECK("cy_wino2_pack_dgrad_weights");
  return 0;
}

extern "C" int cy_conv4x4s2_winograd_dgrad(const float* dZ, const float* U, float* dX, const float* bn_z, const float* bn_scale,
                                           const float* bn_shift, const float* bn_mean, const float* bn_invstd, float bn_slope,
                                           double* bn_red, int B, int H, int W, int Cin, int Cout, void* stream) {
  CY_REQUIRE(dZ && U && dX && B > 0 && H > 0 && W > 0, "cy_conv4x4s2_winograd_dgrad: bad arguments");
  CY_REQUIRE(Cin % 64 == 0 && Cout % 8 == 0, "cy_conv4x4s2_winograd_dgrad: Cin=%d must be a multiple of 64, Cout=%d of 8", Cin, Cout);
  CY_REQUIRE(H % 2 == 0 && W % 2 == 0, "cy_conv4x4s2_winograd_dgrad: H=%d, W=%d must be even", H, W);
  CY_REQUIRE((((uintptr_t)dZ | (uintptr_t)U | (uintptr_t)dX | (uintptr_t)bn_z) & 15) == 0, "cy_conv4x4s2_winograd_dgrad: operands must be 16-byte aligned");
  CY_REQUIRE(bn_red == nullptr || (bn_z && bn_scale && bn_shift && bn_mean && bn_invstd), "cy_conv4x4s2_winograd_dgrad: bn_red needs bn_z / scale / shift / mean / invstd");
  CY_REQUIRE((long long)H * W * Cin < (1ll << 29), "cy_conv4x4s2_winograd_dgrad: image too large for 32-bit offsets");
  Wino2Args a;
  a.X = dZ; a.U = U; a.Y = dX; a.bias = nullptr; a.stats = nullptr;
#ifdef W2_PROF
  if (const char* e = getenv("CY_W2_PROF")) a.bias = (const float*)strtoull(e, nullptr, 0);
#endif
  a.in_scale = a.in_shift = nullptr; a.in_slope = 1.f; a.out_slope = 1.f;
#ifdef W2_PROF
  if (getenv("CY_W2_NOSTORE")) a.in_slope = 2.f;
#endif
  a.B = B; a.H = H / 2; a.W = W / 2; a.Cin = Cout;          // the kernel's "input" is dY [B][H/2][W/2][Cout]
  a.Cout = 4 * Cin; a.Np = 4 * Cin;
  a.Ho = H / 2 + 1; a.Wo = W / 2 + 1;                       // grid of the space-to-depth view
  a.tbh = (a.Ho + 2 * TR2 - 1) / (2 * TR2); a.tbw = (a.Wo + 2 * TC2 - 1) / (2 * TC2);
  a.Hx = H; a.Wx = W; a.Cx = Cin;
  a.bn_z = bn_z; a.bn_scale = bn_scale; a.bn_shift = bn_shift; a.bn_mean = bn_mean; a.bn_invstd = bn_invstd;
  a.bn_red = bn_red; a.bn_slope = bn_slope;
  const long long tiles = (long long)B * a.tbh * a.tbw * (a.Np / 64);
  CY_REQUIRE(tiles < (1ll << 31), "cy_conv4x4s2_winograd_dgrad: too many tiles");
  a.ntiles = (int)tiles;
  long long blocks = 0;
  int rcq = wino2_persistent_blocks(tiles, &blocks, "cy_conv4x4s2_winograd_dgrad");
  if (rcq) return rcq;
  const size_t lds = (size_t)(2 * V2_BUF + 2 * U2_BUF + 2 * RAW2_BUF) * 4;
  int rc = cy_allow_lds(wino2_conv_kernel<1, false>, lds);
  if (rc) return rc;
  wino2_conv_kernel<1, false><<<(unsigned)blocks, 256, lds, (hipStream_t)stream>>>(a);
  CY_LAUNCH_CHECK("cy_conv4x4s2_winograd_dgrad");
  return 0;
}
