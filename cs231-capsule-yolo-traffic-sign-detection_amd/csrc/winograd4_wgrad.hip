// Weight gradient of the 3x3 / stride 1 / pad 1 layers through Winograd F(3x3, 4x4) on the fp32 matrix cores (gfx950):
// DarkCapsuleNet conv_2 (models.py:349-351), the largest launch of the training step.
//
//   dW = A^T [ sum_tiles (G dz G^T) (.) (B^T d B) ] A,   dz = 4x4 tile of the output gradient, d = the 6x6 input patch under it
// 36 multiplies per 16 x 9 multiply-adds: 4x fewer MFMAs than the direct form, 1.78x fewer than F(3x3, 2x2) (winograd.hip).
// B^T is the forward F(4x4,3x3) matrix (points 0, +-1, +-2), G the 6x4 Vandermonde matrix of the same points used UNSCALED
// (integer coefficients); its row scales (1/4, -1/6, -1/6, 1/24, 1/24, 1) and A^T (3x6) are applied by the finish kernel,
// which also adds the partial sums of the tile ranges in a fixed order (deterministic).  fp32 error of dW ~6e-6 relative
// (F(3x3,2x2): 1e-6; tools/probe/wino_f34_wgrad_numerics.py).
//
// The reduction runs over the TILES (the MFMA's k dimension, 4 tiles per v_mfma_f32_16x16x4_f32):
//   dU_xi[ci][co] += sum_tile V_xi[tile][ci] Z_xi[tile][co]      for the 36 positions xi.
// One block = 32 input channels x 64 output channels x all 36 positions for one range of tiles; 4 waves, one per SIMD; wave w
// owns output channels 16 w .. 16 w + 15 and both 16-row halves of the input channels: 36 x 2 accumulator tiles = 288 registers
// (positions 0..31 in AGPRs, 32..35 in VGPRs).  A chunk = 4 tiles in a row (4 x 16 output pixels, 6 x 18 input pixels):
//   * raw patches go global -> registers -> LDS through buffer descriptors of the image (the halo of the input patch that lies
//     outside the image gets an offset beyond the descriptor's range: zeros, one v_and_or_b32 per item),
//   * waves 0, 1 transform the input patch (the forward kernel's half items: 72 packed FMAs per thread), waves 2, 3 the dz
//     tiles (80 packed operations) -- two copies of the whole loop behind ONE branch at the top of the kernel,
//   * A operand image V[pos pair][tile][ci pair][pos][ci parity] (one ds_read_b128 per position pair), B operand image
//     Z[pos][tile][co] with the co bit 4 flipped for odd tiles (conflict-free ds_read_b32).
// Shapes: H % 4 == 0, W % 16 == 0, Cin % 32 == 0, Cout % 64 == 0, H W C 4 < 2^28 (the caller falls back to winograd.hip).
#include "common.h"

namespace {

// developer knob for timing experiments (results are wrong when set): 1 no dz stores, 2 no loads of z (BNF), 4 no BatchNorm arithmetic
#ifndef CY_G4_DBG
#define CY_G4_DBG 0
#endif

constexpr int G4_XPITCH = 36;                   // floats per raw input pixel: 32 channels + 4 pad; pixels stored COLUMN-major (col * 6 + row):
                                                // a column's rows are 144 B apart (ds_read2_b64 pairs), a tile's 4 columns 128 B mod 256
constexpr int G4_RAWX = 6 * 18 * G4_XPITCH;     // raw input patch
constexpr int G4_RAWZ = 64 * 64;                // raw dz patch [4 x 16 pixels][64 channels]
constexpr int G4_V_BUF = 18 * 4 * 16 * 4;       // [pos pair][tile][ci pair][pos in pair][ci parity]
constexpr int G4_Z_BUF = 36 * 4 * 64;           // [pos][tile][co (bit 4 ^ tile parity)]

typedef int i32x4g_ __attribute__((ext_vector_type(4)));
#ifdef CY_G4_PROF
// developer instrumentation (tools/g4prof.py): s_memtime stamps of one chunk per block and role
__device__ unsigned long long g4_prof_buf[256 * 2 * 16];
#define G4_STAMP(s_) { if ((s_) == 0) st_[0] = __builtin_amdgcn_s_memtime(); if ((s_) == 16) st_[1] = __builtin_amdgcn_s_memtime(); \
                       if ((s_) == 31) st_[2] = __builtin_amdgcn_s_memtime(); if ((s_) == 32) st_[3] = __builtin_amdgcn_s_memtime(); \
                       if ((s_) == 48) st_[4] = __builtin_amdgcn_s_memtime(); if ((s_) == 64) st_[5] = __builtin_amdgcn_s_memtime(); \
                       if ((s_) == 65) st_[6] = __builtin_amdgcn_s_memtime(); }
#else
#define G4_STAMP(s_)
#endif

struct Wino4WgradArgs {
  const float* X; const float* dZ; float* slab;
  int B, H, W, Cin, Cout, gh, gw, nrange;
  // BNF: dZ is d = dA * lrelu'(y) (premasked); dz = d * scale + (z - mean) * kb + kc is formed on the way in and written to dZout
  const float* Z; float* dZout;
  const float *scale, *mean, *invstd;
  const double* red; double inv_count;
};

__device__ __forceinline__ void g4_mfma_a(f32x4& c, float a, float b) {
  asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+a"(c) : "v"(a), "v"(b));
}
__device__ __forceinline__ void g4_mfma_v(f32x4& c, float a, float b) {
  asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(c) : "v"(a), "v"(b));
}
__device__ __forceinline__ f32x2 g4_fma(f32x2 x, f32x2 y, f32x2 z) { return __builtin_elementwise_fma(x, y, z); }
__device__ __forceinline__ f32x2 g4_fnma(f32x2 x, f32x2 y, f32x2 z) { return __builtin_elementwise_fma(-x, y, z); }
// loads hipcc does not count (winograd4.hip: a tracked load pending at the loop header draws vmcnt(0)); waited for by hand
__device__ __forceinline__ void g4_load(f32x4& dst, i32x4g_ desc, unsigned voff, unsigned soff) {
  asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen" : "=v"(dst) : "v"(voff), "s"(desc), "s"(soff));
}
__device__ __forceinline__ void g4_store(const f32x4& src, i32x4g_ desc, unsigned voff, unsigned soff) {
  asm volatile("buffer_store_dwordx4 %0, %1, %2, %3 offen\n\ts_nop 1" : : "v"(src), "v"(voff), "s"(desc), "s"(soff) : "memory");
}
template <int N> __device__ __forceinline__ void g4_vmwait(f32x4& x) { asm volatile("s_waitcnt vmcnt(%1)" : "+v"(x) : "n"(N)); }
__device__ __forceinline__ float g4_acc_elem(float a_elem) {
  float x;
  asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(x) : "a"(a_elem));
  return x;
}
__device__ __forceinline__ i32x4g_ g4_desc(const void* p, int bytes) {
  const unsigned long long b = (unsigned long long)(uintptr_t)p;
  return i32x4g_{(int)(unsigned)b, (int)(unsigned)((b >> 32) & 0xffffu), bytes, 0x00020000};
}

// ---- compile-time schedule of one chunk: 72 slots; slot s issues the MFMA of position s >> 1, input-channel half s & 1.
// ROLE 0 (waves 0, 1): 6 T_rd + 12 T_col + 12 T_row pieces of the input transform; ROLE 1 (waves 2, 3): 4 Z_rd + 8 Z_col +
// 12 Z_row pieces of the dz transform.  The raw reads (and the column passes that consume them) stand in front of the mid barrier
// (slot 18), behind it 8 S_raw (one float4 of chunk f + 2: registers -> raw LDS), each followed by its G (the same register's load
// of chunk f + 4), the row passes and the cursor step are spread over slots 19 .. 62 (packed into the first 31 slots the transform
// ran at 2.2x its slots' MFMA time); the end barrier (slot 64) stands behind the last fragment read.
constexpr int G4_MID = 18, G4_END = 64;
struct G4Sched { int kind[72]; int idx[72]; };   // 1 rd, 2 col, 3 row, 4 S_raw, 5 G, 6 ADV
constexpr G4Sched g4_make_sched(int role) {
  G4Sched s{};
  for (int i = 0; i < 72; ++i) { s.kind[i] = 0; s.idx[i] = 0; }
  int pk[64] = {}, n = 0, sl = 0;
  // in front of the mid barrier: every raw read (and what consumes it at once)
  if (role == 0) {                              // rd0 rd1 c0a c0b rd2 c1a c1b rd3 ... rd5 c4a c4b c5a c5b
    pk[n++] = 1; pk[n++] = 1;
    for (int c = 0; c < 6; ++c) { pk[n++] = 2; pk[n++] = 2; if (c < 4) pk[n++] = 1; }
  } else {                                      // rd0..3, the 8 column pieces, the first 6 row pieces
    for (int i = 0; i < 4; ++i) pk[n++] = 1;
    for (int i = 0; i < 8; ++i) pk[n++] = 2;
    for (int i = 0; i < 6; ++i) pk[n++] = 3;
  }
  for (int i = 0; i < n; ++i) s.kind[sl++] = pk[i];
  // behind it (slots 19 .. 62): 8 x (S_raw, its G, a row piece), the remaining row pieces, the cursor step -- spread evenly
  n = 0;
  const int rows_left = role == 0 ? 12 : 6;
  int r = 0;
  for (int k = 0; k < 8; ++k) { pk[n++] = 4; pk[n++] = 5; if (r < rows_left) { pk[n++] = 3; ++r; } }
  for (; r < rows_left; ++r) pk[n++] = 3;
  pk[n++] = 6;
  for (int i = 0; i < n; ++i) s.kind[G4_MID + 1 + (i * 44) / n] = pk[i];
  int cnt[8] = {};
  for (int i = 0; i < 72; ++i) { s.idx[i] = cnt[s.kind[i]]; cnt[s.kind[i]]++; }
  return s;
}
constexpr G4Sched G4S0 = g4_make_sched(0), G4S1 = g4_make_sched(1);
constexpr bool g4_sched_ok(const G4Sched& s, int role) {
  int cnt[8] = {};
  for (int i = 0; i < 72; ++i) cnt[s.kind[i]]++;
  for (int i = 0; i <= G4_MID; ++i) if (s.kind[i] >= 4) return false;          // raw LDS is rewritten only behind the mid barrier
  for (int i = G4_MID + 1; i < 72; ++i) if (s.kind[i] == 1) return false;       // ... and read only in front of it
  for (int i = G4_END; i < 72; ++i) if (s.kind[i] != 0) return false;
  return cnt[1] == (role == 0 ? 6 : 4) && cnt[2] == (role == 0 ? 12 : 8) && cnt[3] == 12 && cnt[4] == 8 && cnt[5] == 8 && cnt[6] == 1;
}
static_assert(g4_sched_ok(G4S0, 0) && g4_sched_ok(G4S1, 1), "winograd4_wgrad: chunk schedule incomplete");
constexpr int g4_kind(int role, int s) { return role == 0 ? G4S0.kind[s] : G4S1.kind[s]; }
constexpr int g4_idx(int role, int s) { return role == 0 ? G4S0.idx[s] : G4S1.idx[s]; }

template <int BNF, int ROLE>
__device__ __forceinline__ void g4_run(const Wino4WgradArgs& a, float* smem) {
  float* Vs = smem;                             // [2][G4_V_BUF]
  float* Zs = smem + 2 * G4_V_BUF;              // [2][G4_Z_BUF]
  float* Rx = Zs + 2 * G4_Z_BUF;                // raw input patch
  float* Rz = Rx + G4_RAWX;                     // raw dz patch

  const int t = threadIdx.x, lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int ncb = a.Cout / 64, nib = a.Cin / 32;
  int vid = blockIdx.x;
  if ((gridDim.x & 7) == 0) vid = (blockIdx.x & 7) * (gridDim.x >> 3) + (blockIdx.x >> 3);
  const int cob = vid % ncb;
  const int cib = (vid / ncb) % nib;
  const int jr = vid / (ncb * nib);
  const int ngroups = a.gh * a.gw;
  const long long gtot = (long long)a.B * ngroups;
  // ranges are cut in chunk PAIRS (the loop body holds two chunks: static register sets and LDS buffers); a range whose last pair
  // has no second chunk processes a phantom one whose input patch reads as zeros (V = 0: no contribution) and whose dz is not stored
  const long long ptot = (gtot + 1) / 2;
  const int pbeg = (int)(ptot * jr / a.nrange), pend = (int)(ptot * (jr + 1) / a.nrange);
  const int cbeg = 2 * pbeg;
  const int cend = 2 * pend < gtot ? 2 * pend : (int)gtot;
  const int nchunk = cend - cbeg;               // >= 1 real chunks
  const int npair = pend - pbeg;                // >= 1

  // ---- loaders.  Input patch: 864 float4 items (108 pixels x 8), item = t + 256 q: pixel = item >> 3, float4 = item & 7 (the 8
  // lanes of a pixel read its 128 contiguous bytes); the 4th round is partial (items of t >= 96 repeat their 3rd item).
  // dz patch: 1024 items, pixel = item >> 4, float4 = item & 15.
  const int ximg_bytes = a.H * a.W * a.Cin * 4, zimg_bytes = a.H * a.W * a.Cout * 4;
  const int xshift = (a.W + 1) * a.Cin * 4;     // the descriptor's base stands one row and one pixel in front of the image
  unsigned xvoff[4], xhfl[4], zvoff[4], zsto[4];   // zsto: BNF, store offset of dz item q -- out of the descriptor's range unless this block owns the item
  int xroff[4], zroff[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int item = (q < 3 || t < 96) ? t + 256 * q : t + 512;
    const int pix = item >> 3, c4 = item & 7, pr = pix / 18, pc = pix - pr * 18;
    xvoff[q] = (unsigned)(((pr * a.W + pc) * a.Cin + cib * 32 + c4 * 4) * 4);
    xhfl[q] = (pr == 0 ? 1u << 28 : 0u) | (pr == 5 ? 1u << 29 : 0u) | (pc == 0 ? 1u << 30 : 0u) | (pc == 17 ? 1u << 31 : 0u);
    xroff[q] = (pc * 6 + pr) * G4_XPITCH + c4 * 4;
    // BNF == 4 (four input-channel blocks share a dz tile): the items are rotated by the block's index, so that item 0 -- one pixel
    // row of the tile -- is the one this block writes back: one unconditional store per chunk, none for the other items
    const int zi = t + 256 * (BNF == 4 ? ((q + cib) & 3) : q), zp = zi >> 4, zc = zi & 15;
    zvoff[q] = (unsigned)((((zp >> 4) * a.W + (zp & 15)) * a.Cout + cob * 64 + zc * 4) * 4);
    zroff[q] = zp * 64 + zc * 4;
    // the nib blocks that share a dz tile all form it; each of (at most four of) them writes a share of it: with ONE writer its four
    // stores per chunk sat in front of its own prefetch loads (vector-memory operations retire in order): +3.5 ms on the launch
    const int nsh = nib < 4 ? nib : 4;
    zsto[q] = BNF == 4 ? zvoff[q] : (cib < nsh && (q % nsh) == cib) ? zvoff[q] : 0x80000000u;
  }
  // chunk cursor of the loads (uniform): image, tile row, chunk column
  int lgb = 0, lty = 0, lcx = 0;
  {
    const int gb = cbeg / ngroups, gr = cbeg - gb * ngroups;
    lgb = gb; lty = gr / a.gw; lcx = gr - lty * a.gw;
  }
  int lrem = nchunk - 1;                        // chunks the cursor may still advance (it stops at the block's last chunk)
  i32x4g_ xdesc, zdesc, zzdesc, odesc;
  unsigned xsoff = 0, zsoff = 0, xbt = 0;
  bool phantom = false;                         // the cursor stands behind the block's last real chunk
  // The chunk offsets RUN (one add per chunk) and the four descriptors are rebuilt only when the image changes: recomputed from (image,
  // row, column) per chunk they were ~170 scalar instructions -- 64-bit multiplies for every base -- in ONE slot of a stream in which
  // every instruction of the wave, scalar or not, takes an issue slot.  (No new uniform state: the kernel stands at 100 SGPRs, and
  // running image pointers pushed it into v_writelane spills.)
  auto set_desc = [&]() {
    xdesc = g4_desc((const char*)(a.X + (long long)lgb * a.H * a.W * a.Cin) - xshift, phantom ? 0 : ximg_bytes + xshift);
    zdesc = g4_desc(a.dZ + (long long)lgb * a.H * a.W * a.Cout, zimg_bytes);
    if constexpr (BNF) {
      zzdesc = g4_desc(a.Z + (long long)lgb * a.H * a.W * a.Cout, zimg_bytes);
      odesc = g4_desc(a.dZout + (long long)lgb * a.H * a.W * a.Cout, phantom ? 0 : zimg_bytes);
    }
  };
  auto set_edges = [&]() {
    xbt = (lty == 0 ? 1u << 28 : 0u) | (lty == a.gh - 1 ? 1u << 29 : 0u) | (lcx == 0 ? 1u << 30 : 0u) | (lcx == a.gw - 1 ? 1u << 31 : 0u);
  };
  auto set_row = [&]() {                        // offsets of the first chunk of tile row lty
    xsoff = (unsigned)((4 * lty) * a.W * a.Cin * 4);
    zsoff = (unsigned)((4 * lty) * a.W * a.Cout * 4);
  };
  auto set_cursor = [&]() {                     // (prologue only)
    set_desc();
    set_row();
    xsoff += (unsigned)(16 * lcx * a.Cin * 4);
    zsoff += (unsigned)(16 * lcx * a.Cout * 4);
    set_edges();
  };
  auto advance = [&]() {
    if (lrem <= 0) {
      if (!phantom) { phantom = true; set_desc(); }
      return;
    }
    --lrem;
    ++lcx; xsoff += (unsigned)(64 * a.Cin); zsoff += (unsigned)(64 * a.Cout);
    if (lcx == a.gw) {
      lcx = 0;
      if (++lty == a.gh) { lty = 0; ++lgb; set_desc(); }
      set_row();
    }
    set_edges();
  };
  // two register sets: at the top of iteration f set f & 1 holds chunk f + 2 and the other set chunk f + 3, both possibly still in
  // flight (a load has two chunks = 2 us to return: with one set -- 1 us -- the S_raw waits were the loop's largest stall)
  f32x4 gx[2][4], gz[2][4], zz[2][4];
  unsigned osoff[2] = {0u, 0u};                 // BNF: scalar offset / descriptor of the chunk held in set P (its dz goes there)
  i32x4g_ odesc_held[2];
  auto Gx = [&](int P, int q) { g4_load(gx[P][q], xdesc, (xhfl[q] & xbt) | xvoff[q], xsoff); };
  auto Gz = [&](int P, int q) {
    g4_load(gz[P][q], zdesc, zvoff[q], zsoff);
    if constexpr (BNF) { if (CY_G4_DBG & 2) zz[P][q] = gz[P][q]; else g4_load(zz[P][q], zzdesc, zvoff[q], zsoff); }
  };
  // BNF: per-channel constants of this thread's 4 dz channels: dz = d * sc + (z - mu) * kb + kc
  // dz = d * sc + z * kb + kc with kc = -sc * mean(d) - mean * kb folded (two packed FMAs per channel pair)
  f32x2 k2sc[2], k2b[2], k2c[2];
  if constexpr (BNF) {
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int ch = cob * 64 + (t & 15) * 4 + k;
      const float sc = a.scale[ch], is = a.invstd[ch];
      const float m1 = (float)(a.red[2 * ch] * a.inv_count), m2 = (float)(a.red[2 * ch + 1] * a.inv_count);
      const float kb = -sc * is * m2;
      k2sc[k >> 1][k & 1] = sc; k2b[k >> 1][k & 1] = kb; k2c[k >> 1][k & 1] = __builtin_fmaf(-a.mean[ch], kb, -sc * m1);
    }
  }
  auto Sx = [&](int P, int q) { *(f32x4*)(Rx + xroff[q]) = gx[P][q]; };
  auto Sz = [&](int P, int q) {
    if constexpr (BNF) {
      f32x4 o;
      if (CY_G4_DBG & 4) o = gz[P][q];
      else {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const f32x2 zv = {zz[P][q][2 * h], zz[P][q][2 * h + 1]}, dv = {gz[P][q][2 * h], gz[P][q][2 * h + 1]};
          const f32x2 r = g4_fma(dv, k2sc[h], g4_fma(zv, k2b[h], k2c[h]));
          o[2 * h] = r[0]; o[2 * h + 1] = r[1];
        }
      }
      *(f32x4*)(Rz + zroff[q]) = o;
      if (!(CY_G4_DBG & 1) && (BNF != 4 || q == 0)) g4_store(o, odesc_held[P], zsto[q], osoff[P]);
    } else {
      *(f32x4*)(Rz + zroff[q]) = gz[P][q];
    }
  };

  // ---- ROLE 0: input transform, thread (ci pair r, tile kg, row half hr): rows 3 hr .. 3 hr + 2 of V = B^T d B for channels
  // 2 r, 2 r + 1 as float2 (winograd4.hip: per-lane coefficients alpha / gamma and a row-shifted base make the halves one code)
  const int tt_ = t & 127;
  const int vr = tt_ & 15, vkg = (tt_ >> 4) & 3, hr = tt_ >> 6;
  const int tbase = (4 * vkg) * 6 * G4_XPITCH + 2 * vr;
  const int tbase_e = tbase + hr * G4_XPITCH;
  const int vdst = (vkg * 16 + vr) * 4;
  const int vd0 = vdst + (hr ? 5 : 0) * 768, vd1 = vdst + (hr ? 3 : 1) * 768, vd2 = vdst + (hr ? 4 : 2) * 768;
  const float alpha_ = hr ? -1.f : -4.f, gamma_ = hr ? 2.f : 1.f;
  const f32x2 kal = {alpha_, alpha_}, kga = {gamma_, gamma_};
  const f32x2 k4 = {4.f, 4.f}, km5 = {-5.f, -5.f}, km4 = {-4.f, -4.f}, k2 = {2.f, 2.f};
  f32x2 tt[3][6], dc[2][7], cx_, cy_, ci_, rt_[6];
  auto Trd = [&](int c) {
    f32x2* d = dc[c & 1];
    d[0] = *(const f32x2*)(Rx + tbase_e + (c * 6 + 0) * G4_XPITCH);
    d[1] = *(const f32x2*)(Rx + tbase_e + (c * 6 + 2) * G4_XPITCH);
    d[2] = *(const f32x2*)(Rx + tbase_e + (c * 6 + 4) * G4_XPITCH);
    d[3] = *(const f32x2*)(Rx + tbase + (c * 6 + 1) * G4_XPITCH);
    d[4] = *(const f32x2*)(Rx + tbase + (c * 6 + 2) * G4_XPITCH);
    d[5] = *(const f32x2*)(Rx + tbase + (c * 6 + 3) * G4_XPITCH);
    d[6] = *(const f32x2*)(Rx + tbase + (c * 6 + 4) * G4_XPITCH);
  };
  auto Tcol = [&](int c, int part) {
    const f32x2* d = dc[c & 1];
    if (part == 0) {
      cx_ = g4_fma(kal, d[4], d[6]);
      cy_ = g4_fma(kal, d[3], d[5]);
      ci_ = g4_fma(km5, d[1], d[2]);
    } else {
      tt[1][c] = g4_fma(kga, cy_, cx_);
      tt[2][c] = g4_fnma(kga, cy_, cx_);
      tt[0][c] = g4_fma(k4, d[0], ci_);
    }
  };
  auto Trow = [&](float* vb, int r, int part) {
    const f32x2* x = tt[r];
    float* v = vb + (r == 0 ? vd0 : r == 1 ? vd1 : vd2);   // + (j >> 1) * 256 + (j & 1) * 2
    if (part == 0) {
      rt_[0] = g4_fma(km5, x[2], x[4]);
      rt_[1] = g4_fma(km4, x[2], x[4]);
      rt_[2] = g4_fma(km4, x[1], x[3]);
    } else if (part == 1) {
      rt_[3] = x[4] - x[2];
      rt_[4] = x[3] - x[1];
      rt_[5] = g4_fma(km5, x[3], x[5]);
    } else if (part == 2) {
      *(f32x2*)(v + 0) = g4_fma(k4, x[0], rt_[0]);
      *(f32x2*)(v + 2) = rt_[1] + rt_[2];
      *(f32x2*)(v + 256) = rt_[1] - rt_[2];
    } else {
      *(f32x2*)(v + 256 + 2) = g4_fma(k2, rt_[4], rt_[3]);
      *(f32x2*)(v + 512) = g4_fnma(k2, rt_[4], rt_[3]);
      *(f32x2*)(v + 512 + 2) = g4_fma(k4, x[1], rt_[5]);
    }
  };
  // ---- ROLE 1: dz transform, thread (co pair cp, tile kg): U = G z G^T (6x6 from 4x4) for channels 2 cp, 2 cp + 1
  const int zcp = tt_ & 31, zkg = tt_ >> 5;
  const int zbase = (4 * zkg) * 64 + 2 * zcp;
  const int zdst = zkg * 64 + ((2 * zcp) ^ ((zkg & 1) << 4));
  f32x2 zin[4][4], zt[6][4], za_[4];            // zt[i][c]: column pass
  auto Zrd = [&](int pr) {
#pragma unroll
    for (int c = 0; c < 4; ++c) zin[pr][c] = *(const f32x2*)(Rz + zbase + (pr * 16 + c) * 64);
  };
  auto Zcol = [&](int c, int part) {            // G on the 4 rows of column c: rows 0 and 5 are copies
    if (part == 0) {
      za_[0] = zin[0][c] + zin[2][c];
      za_[1] = zin[1][c] + zin[3][c];
      za_[2] = g4_fma(k4, zin[2][c], zin[0][c]);
      za_[3] = g4_fma(k4, zin[3][c], zin[1][c]);
    } else {
      zt[1][c] = za_[0] + za_[1];
      zt[2][c] = za_[0] - za_[1];
      zt[3][c] = g4_fma(k2, za_[3], za_[2]);
      zt[4][c] = g4_fnma(k2, za_[3], za_[2]);
      zt[0][c] = zin[0][c];
      zt[5][c] = zin[3][c];
    }
  };
  auto Zrow = [&](float* zb, int i, int part) {
    const f32x2* x = zt[i];
    float* u = zb + zdst + (6 * i) * 256;
    if (part == 0) {
      za_[0] = x[0] + x[2];
      za_[1] = x[1] + x[3];
      za_[2] = g4_fma(k4, x[2], x[0]);
      za_[3] = g4_fma(k4, x[3], x[1]);
    } else {
      *(f32x2*)(u + 0 * 256) = x[0];
      *(f32x2*)(u + 1 * 256) = za_[0] + za_[1];
      *(f32x2*)(u + 2 * 256) = za_[0] - za_[1];
      *(f32x2*)(u + 3 * 256) = g4_fma(k2, za_[3], za_[2]);
      *(f32x2*)(u + 4 * 256) = g4_fnma(k2, za_[3], za_[2]);
      *(f32x2*)(u + 5 * 256) = x[3];
    }
  };
  auto Tall = [&](int buf) {
    if constexpr (ROLE == 0) {
      float* vb = Vs + buf * G4_V_BUF;
#pragma unroll
      for (int c = 0; c < 6; ++c) { Trd(c); Tcol(c, 0); Tcol(c, 1); }
#pragma unroll
      for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int part = 0; part < 4; ++part) Trow(vb, r, part);
    } else {
      float* zb = Zs + buf * G4_Z_BUF;
#pragma unroll
      for (int pr = 0; pr < 4; ++pr) Zrd(pr);
#pragma unroll
      for (int c = 0; c < 4; ++c) { Zcol(c, 0); Zcol(c, 1); }
#pragma unroll
      for (int i = 0; i < 6; ++i) { Zrow(zb, i, 0); Zrow(zb, i, 1); }
    }
  };

  // ---- prologue.  State at the top of iteration f (P = f & 1): V / Z[P] = chunk f, raw LDS = chunk f + 1, register set P = chunk
  // f + 2, set 1 - P = chunk f + 3, load cursor at f + 4.  Every round issues its loads in the loop's order (Gx0..3, Gz0..3): the
  // loop's hand-counted waits hold for its first iterations too.
  auto Gall = [&](int P) {
#pragma unroll
    for (int q = 0; q < 4; ++q) Gx(P, q);
#pragma unroll
    for (int q = 0; q < 4; ++q) Gz(P, q);
    if constexpr (BNF) { osoff[P] = zsoff; odesc_held[P] = odesc; }
    advance();
  };
  auto Wall = [&](int P) {
#pragma unroll
    for (int q = 0; q < 4; ++q) { g4_vmwait<0>(gx[P][q]); g4_vmwait<0>(gz[P][q]); if constexpr (BNF) g4_vmwait<0>(zz[P][q]); }
  };
  auto Sall = [&](int P) {
#pragma unroll
    for (int q = 0; q < 4; ++q) { Sx(P, q); Sz(P, q); }
  };
  set_cursor();
  Gall(0);                                      // chunk 0
  Wall(0);
  Sall(0);
  Gall(1);                                      // chunk 1
  __syncthreads();
  Tall(0);
  __syncthreads();
  Wall(1);
  Sall(1);
  Gall(0);                                      // chunk 2
  Gall(1);                                      // chunk 3
  __syncthreads();

  // fragments: A of position pair q: 16 bytes at ((q * 4 + kgl) * 16 + ml) * 4; B of position p: ((p * 4 + kgl) * 64 + col)
  const int kgl = lane >> 4, ml = lane & 15;
  const int fragA = (kgl * 16 + ml) * 4;
  const int fragB = kgl * 64 + ((16 * wave + ml) ^ ((kgl & 1) << 4));
  f32x4 fa[3];
  f32x2 fb[3];                                  // B fragments of position pairs (two dwords 1 KiB apart: one ds_read2st64_b32)
  auto rdB = [&](const float* zp) { return f32x2{zp[0], zp[256]}; };
  fa[0] = *(const f32x4*)(Vs + fragA);
  fa[1] = *(const f32x4*)(Vs + fragA + 256);
  fb[0] = rdB(Zs + fragB);
  fb[1] = rdB(Zs + fragB + 512);

  f32x4 accA[32][2], accV[4][2];
#pragma unroll
  for (int p = 0; p < 32; ++p)
#pragma unroll
    for (int h = 0; h < 2; ++h) accA[p][h] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int p = 0; p < 4; ++p)
#pragma unroll
    for (int h = 0; h < 2; ++h) accV[p][h] = f32x4{0.f, 0.f, 0.f, 0.f};

  // vector-memory operations per iteration, in issue order: Gx0..3, then per dz item [dz store (BNF)], Gz, [Gzz (BNF)].  S_raw(q) of
  // iteration f waits for the load of iteration f - 2 into the same register: exactly 15 younger operations without BNF; with BNF
  // at least 22 (iteration 0, whose loads the prologue issued without stores: 23 for the x items, 23 + q for d, 22 + q for z; later
  // 31 / 30 / 29) -- a lower bound of the younger operations is what a wait needs
  // (BNF == 4: one store per iteration: 13 operations per iteration, at least 22 younger ones as well)
  constexpr int VMW = BNF ? 22 : 15;
#ifdef CY_G4_PROF
  unsigned long long st_[8];
#endif
  for (int f = 0; f < npair; ++f) {
#define G4CHUNK(PAR)                                                                                  \
  {                                                                                                   \
    constexpr int P_ = (PAR);                                                                         \
    const float* va_ = Vs + P_ * G4_V_BUF + fragA;                                                    \
    const float* zb_ = Zs + P_ * G4_Z_BUF + fragB;                                                    \
    float* vw_ = Vs + (1 - P_) * G4_V_BUF;                                                            \
    float* zw_ = Zs + (1 - P_) * G4_Z_BUF;                                                            \
    G4SLOT8(0) G4SLOT8(8) G4SLOT8(16) G4SLOT8(24) G4SLOT8(32) G4SLOT8(40) G4SLOT8(48) G4SLOT8(56) G4SLOT8(64) \
  }
#define G4SLOT(SIDX)                                                                                  \
    {                                                                                                 \
      constexpr int s_ = (SIDX), p_ = s_ >> 1, mt_ = s_ & 1, q_ = p_ >> 1;                            \
      G4_STAMP(s_)                                                                                    \
      if (s_ == G4_MID || s_ == G4_END) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); \
      if (p_ < 32) g4_mfma_a(accA[p_ < 32 ? p_ : 0][mt_], fa[q_ % 3][2 * (p_ & 1) + mt_], fb[q_ % 3][p_ & 1]); \
      else g4_mfma_v(accV[p_ >= 32 ? p_ - 32 : 0][mt_], fa[q_ % 3][2 * (p_ & 1) + mt_], fb[q_ % 3][p_ & 1]);   \
      if (s_ < G4_END) {                                                                              \
        if ((s_ & 3) == 0 && q_ + 2 < 18) {                                                           \
          fa[(q_ + 2) % 3] = *(const f32x4*)(va_ + (q_ + 2) * 256);                                   \
          fb[(q_ + 2) % 3] = rdB(zb_ + (q_ + 2) * 512);                                               \
        }                                                                                             \
      } else {                                  /* behind the end barrier: the next chunk's first fragments (ring slot 0 is free at slot 64, slot 1 at 68) */ \
        if (s_ == 64) { fa[0] = *(const f32x4*)(vw_ + fragA); fb[0] = rdB(zw_ + fragB); }             \
        if (s_ == 68) { fa[1] = *(const f32x4*)(vw_ + fragA + 256); fb[1] = rdB(zw_ + fragB + 512); } \
      }                                                                                               \
      constexpr int kind = g4_kind(ROLE, s_), k_ = g4_idx(ROLE, s_);                                  \
      if (kind == 1) {                                                                                \
        if constexpr (ROLE == 0) Trd(k_ % 6); else Zrd(k_ % 4);                                       \
      } else if (kind == 2) {                                                                         \
        if constexpr (ROLE == 0) Tcol((k_ >> 1) % 6, k_ & 1); else Zcol((k_ >> 1) % 4, k_ & 1);       \
      } else if (kind == 3) {                                                                         \
        if constexpr (ROLE == 0) Trow(vw_, (k_ >> 2) % 3, k_ & 3); else Zrow(zw_, (k_ >> 1) % 6, k_ & 1); \
      } else if (kind == 4) {                   /* S_raw: x items 0..3, then dz items 0..3 */        \
        if (k_ < 4) { g4_vmwait<VMW>(gx[P_][k_ & 3]); Sx(P_, k_ & 3); }                               \
        else { g4_vmwait<VMW>(gz[P_][k_ & 3]); if constexpr (BNF) g4_vmwait<VMW>(zz[P_][k_ & 3]); Sz(P_, k_ & 3); } \
      } else if (kind == 5) {                                                                         \
        if (k_ < 4) Gx(P_, k_ & 3); else Gz(P_, k_ & 3);                                              \
      } else if (kind == 6) {                                                                         \
        if constexpr (BNF) { osoff[P_] = zsoff; odesc_held[P_] = odesc; }                             \
        advance();                                                                                    \
      }                                                                                               \
      __builtin_amdgcn_sched_barrier(0);                                                              \
    }
#define G4SLOT8(B) G4SLOT((B)) G4SLOT((B) + 1) G4SLOT((B) + 2) G4SLOT((B) + 3) G4SLOT((B) + 4) G4SLOT((B) + 5) G4SLOT((B) + 6) G4SLOT((B) + 7)
    G4CHUNK(0)
#ifdef CY_G4_PROF
    st_[7] = __builtin_amdgcn_s_memtime();
    if ((t & 127) == 0 && f == 7) {
      unsigned long long* pb = g4_prof_buf + (blockIdx.x * 2 + ROLE) * 16;
      for (int i = 0; i < 8; ++i) pb[i] = st_[i];
    }
#endif
    G4CHUNK(1)
#undef G4SLOT8
#undef G4SLOT
#undef G4CHUNK
  }

  // the loads of the chunks behind the block's last one are still in flight and hipcc does not know: their registers must not be
  // reused before they have landed
#pragma unroll
  for (int q = 0; q < 4; ++q)
#pragma unroll
    for (int P = 0; P < 2; ++P) { g4_vmwait<0>(gx[P][q]); g4_vmwait<0>(gz[P][q]); if constexpr (BNF) g4_vmwait<0>(zz[P][q]); }
  // ---- the block's partial sums: slab[range][pos][ci][co]; lane l, register e of accumulator (pos, mt): ci pair 4 (l >> 4) + e
  float* sl = a.slab + ((long long)jr * 36 * a.Cin + cib * 32) * a.Cout + cob * 64 + 16 * wave + ml;
#pragma unroll
  for (int p = 0; p < 36; ++p)
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float v = p < 32 ? g4_acc_elem(accA[p < 32 ? p : 0][mt][e]) : accV[p >= 32 ? p - 32 : 0][mt][e];
        sl[((long long)p * a.Cin + 2 * (4 * kgl + e) + mt) * a.Cout] = v;
      }
}

template <int BNF>
__global__ __launch_bounds__(256, 1) void wino4_wgrad_kernel(Wino4WgradArgs a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  if (threadIdx.x < 128) g4_run<BNF, 0>(a, smem);
  else g4_run<BNF, 1>(a, smem);
}

// sum of the ranges' partial sums, in range order (deterministic), into the slab of range 0: one thread per (pos, ci, co) -- a layer
// with few channels has few (ci, co) pairs but many ranges (DarkNet conv_2: 2048 pairs x 256 ranges), so the sum over the ranges
// must not sit in the per-pair finish kernel
__global__ void wino4_wgrad_reduce_kernel(float* __restrict__ slab, int nrange, long long per_range) {
  const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= per_range) return;
  float s = slab[idx];
  for (int r = 1; r < nrange; ++r) s += slab[(long long)r * per_range + idx];
  slab[idx] = s;
}

// dW[co][ci][3][3] = A^T [ S dU S ] A with S = diag(1/4, -1/6, -1/6, 1/24, 1/24, 1), A^T = [[1,1,1,1,1,0],[0,1,-1,2,-2,0],[0,1,1,4,4,1]]
__global__ void wino4_wgrad_finish_kernel(const float* __restrict__ slab, float* __restrict__ dW, int Cin, int Cout) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= Cin * Cout) return;
  const int co = idx % Cout, ci = idx / Cout;
  const float sc[6] = {0.25f, -1.f / 6, -1.f / 6, 1.f / 24, 1.f / 24, 1.f};
  float m[6][6];
#pragma unroll
  for (int i = 0; i < 6; ++i)
#pragma unroll
    for (int j = 0; j < 6; ++j) m[i][j] = slab[((long long)(6 * i + j) * Cin + ci) * Cout + co] * (sc[i] * sc[j]);
  float sr[6][3];
#pragma unroll
  for (int i = 0; i < 6; ++i) {
    sr[i][0] = m[i][0] + m[i][1] + m[i][2] + m[i][3] + m[i][4];
    sr[i][1] = (m[i][1] - m[i][2]) + 2.f * (m[i][3] - m[i][4]);
    sr[i][2] = (m[i][1] + m[i][2]) + 4.f * (m[i][3] + m[i][4]) + m[i][5];
  }
#pragma unroll
  for (int l = 0; l < 3; ++l) {
    float* o = dW + ((long long)co * Cin + ci) * 9 + l;
    o[0] = sr[0][l] + sr[1][l] + sr[2][l] + sr[3][l] + sr[4][l];
    o[3] = (sr[1][l] - sr[2][l]) + 2.f * (sr[3][l] - sr[4][l]);
    o[6] = (sr[1][l] + sr[2][l]) + 4.f * (sr[3][l] + sr[4][l]) + sr[5][l];
  }
}

int g4_nrange(int B, int H, int W, int Cin, int Cout) {
  int dev = 0, ncu = 256;
  if (hipGetDevice(&dev) == hipSuccess) hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev);
  const long long tiles = ((long long)B * (H / 4) * (W / 16) + 1) / 2;   // chunk pairs
  const int per = (Cin / 32) * (Cout / 64);
  long long n = ncu / per;
#ifdef CY_G4_NRANGE_ENV
  if (const char* e = getenv("CY_G4_NRANGE")) n = atoll(e);            // developer override (builds with -DCY_G4_NRANGE_ENV only)
#endif
  if (n < 1) n = 1;
  if (n > tiles) n = tiles;
  return (int)n;
}

}  // namespace

extern "C" int cy_wino4_wgrad_ok(int B, int H, int W, int Cin, int Cout) {
  return H > 0 && W > 0 && H % 4 == 0 && W % 16 == 0 && Cin % 32 == 0 && Cout % 64 == 0 &&
         (long long)H * W * Cin * 4 + (long long)(W + 1) * Cin * 4 < (1ll << 28) && (long long)H * W * Cout * 4 < (1ll << 28) && B > 0;
}

extern "C" long long cy_wino4_wgrad_ws_floats(int B, int H, int W, int Cin, int Cout) {
  return (long long)g4_nrange(B, H, W, Cin, Cout) * 36 * Cin * Cout;
}

static int g4_launch(Wino4WgradArgs& a, float* dW, bool bnf, hipStream_t s) {
  a.gh = a.H / 4; a.gw = a.W / 16;
  a.nrange = g4_nrange(a.B, a.H, a.W, a.Cin, a.Cout);
  const int blocks = a.nrange * (a.Cin / 32) * (a.Cout / 64);
  const size_t lds = (size_t)(2 * G4_V_BUF + 2 * G4_Z_BUF + G4_RAWX + G4_RAWZ) * 4;
  int rc = cy_allow_lds(wino4_wgrad_kernel<0>, lds);
  if (rc) return rc;
  rc = cy_allow_lds(wino4_wgrad_kernel<1>, lds);
  if (rc) return rc;
  rc = cy_allow_lds(wino4_wgrad_kernel<4>, lds);
  if (rc) return rc;
  if (bnf && a.Cin == 128) wino4_wgrad_kernel<4><<<(unsigned)blocks, 256, lds, s>>>(a);   // four input-channel blocks: one dz row each
  else if (bnf) wino4_wgrad_kernel<1><<<(unsigned)blocks, 256, lds, s>>>(a);
  else wino4_wgrad_kernel<0><<<(unsigned)blocks, 256, lds, s>>>(a);
  CY_LAUNCH_CHECK("cy_conv3x3_winograd4_wgrad");
  const int n = a.Cin * a.Cout;
  if (a.nrange > 1) {
    wino4_wgrad_reduce_kernel<<<(unsigned)cy_ceil_div(36ll * n, 256), 256, 0, s>>>(a.slab, a.nrange, 36ll * n);
    CY_LAUNCH_CHECK("cy_conv3x3_winograd4_wgrad (reduce)");
  }
  wino4_wgrad_finish_kernel<<<(unsigned)cy_ceil_div(n, 256), 256, 0, s>>>(a.slab, dW, a.Cin, a.Cout);
  CY_LAUNCH_CHECK("cy_conv3x3_winograd4_wgrad (finish)");
  return 0;
}

extern "C" int cy_conv3x3_winograd4_wgrad(const float* X, const float* dZ, float* dW, float* ws,
                                          int B, int H, int W, int Cin, int Cout, void* stream) {
  CY_REQUIRE(X && dZ && dW && ws, "cy_conv3x3_winograd4_wgrad: bad arguments");
  CY_REQUIRE(cy_wino4_wgrad_ok(B, H, W, Cin, Cout), "cy_conv3x3_winograd4_wgrad: shape B=%d H=%d W=%d Cin=%d Cout=%d not supported "
             "(H %% 4, W %% 16, Cin %% 32, Cout %% 64, image < 256 MB)", B, H, W, Cin, Cout);
  CY_REQUIRE((((uintptr_t)X | (uintptr_t)dZ) & 15) == 0, "cy_conv3x3_winograd4_wgrad: operands must be 16-byte aligned");
  Wino4WgradArgs a{};
  a.X = X; a.dZ = dZ; a.slab = ws; a.B = B; a.H = H; a.W = W; a.Cin = Cin; a.Cout = Cout;
  return g4_launch(a, dW, false, (hipStream_t)stream);
}

extern "C" int cy_conv3x3_winograd4_wgrad_bn(const float* X, const float* Z, const float* dA, float* dZ, const float* scale,
                                             const float* mean, const float* invstd, const double* red, long long count,
                                             float* dW, float* ws, int B, int H, int W, int Cin, int Cout, void* stream) {
  CY_REQUIRE(X && Z && dA && dZ && scale && mean && invstd && red && dW && ws && count > 0, "cy_conv3x3_winograd4_wgrad_bn: bad arguments");
  CY_REQUIRE(cy_wino4_wgrad_ok(B, H, W, Cin, Cout), "cy_conv3x3_winograd4_wgrad_bn: shape not supported");
  CY_REQUIRE((((uintptr_t)X | (uintptr_t)Z | (uintptr_t)dA | (uintptr_t)dZ) & 15) == 0, "cy_conv3x3_winograd4_wgrad_bn: operands must be 16-byte aligned");
  Wino4WgradArgs a{};
  a.X = X; a.dZ = dA; a.slab = ws; a.B = B; a.H = H; a.W = W; a.Cin = Cin; a.Cout = Cout;
  a.Z = Z; a.dZout = dZ; a.scale = scale; a.mean = mean; a.invstd = invstd; a.red = red; a.inv_count = 1.0 / (double)count;
  return g4_launch(a, dW, true, (hipStream_t)stream);
}

#ifdef CY_G4_PROF
extern "C" int cy_wino4_wgrad_read_prof(unsigned long long* host_dst) {
  return (int)hipMemcpyFromSymbol(host_dst, HIP_SYMBOL(g4_prof_buf), sizeof(unsigned long long) * 256 * 2 * 16);
}
#endif
