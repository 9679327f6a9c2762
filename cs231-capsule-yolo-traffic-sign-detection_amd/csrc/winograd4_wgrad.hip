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

constexpr int G4_XPITCH = 40;                   // floats per raw input pixel: 32 channels + 8 pad (tile stride = 128 B mod 256)
constexpr int G4_RAWX = 6 * 18 * G4_XPITCH;     // raw input patch
constexpr int G4_RAWZ = 64 * 64;                // raw dz patch [4 x 16 pixels][64 channels]
constexpr int G4_V_BUF = 18 * 4 * 16 * 4;       // [pos pair][tile][ci pair][pos in pair][ci parity]
constexpr int G4_Z_BUF = 36 * 4 * 64;           // [pos][tile][co (bit 4 ^ tile parity)]

typedef int i32x4g_ __attribute__((ext_vector_type(4)));

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
// 12 Z_row pieces of the dz transform; then, for both: the mid barrier (every raw read of the chunk has returned), 8 S_raw
// (one float4 of chunk f + 2: registers -> raw LDS) each followed by its G (the same register's load of chunk f + 3), the
// cursor step, and the end barrier (slot 64) behind the last fragment read.
constexpr int G4_MID = 31, G4_END = 64;
constexpr int g4_kind(int role, int s) {        // 1 rd, 2 col, 3 row, 4 S_raw, 5 G, 6 ADV
  const int nrd = role == 0 ? 6 : 4, ncol = role == 0 ? 12 : 8;
  if (s < nrd + ncol + 12) {
    if (role == 0) {                            // rd0 rd1 col0a col0b rd2 col1a col1b ... col5a col5b, then rows
      if (s < 30 - 12) {
        if (s < 2) return 1;
        const int k = s - 2;                    // (col a, col b, rd) triples
        if (k < 12) return (k % 3) == 2 ? 1 : 2;
        return 2;
      }
      return 3;
    }
    return s < nrd ? 1 : s < nrd + ncol ? 2 : 3;
  }
  if (s > G4_MID && s < G4_MID + 17) return ((s - G4_MID - 1) & 1) ? 5 : 4;
  if (s == G4_MID + 17) return 6;
  return 0;
}
constexpr int g4_idx(int role, int s) {
  const int k = g4_kind(role, s);
  int n = 0;
  for (int i = 0; i < s; ++i) n += g4_kind(role, i) == k ? 1 : 0;
  return n;
}

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
  const int cbeg = (int)(gtot * jr / a.nrange), cend = (int)(gtot * (jr + 1) / a.nrange);
  const int nchunk = cend - cbeg;               // >= 1

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
    xroff[q] = pix * G4_XPITCH + c4 * 4;
    const int zi = t + 256 * q, zp = zi >> 4, zc = zi & 15;
    zvoff[q] = (unsigned)((((zp >> 4) * a.W + (zp & 15)) * a.Cout + cob * 64 + zc * 4) * 4);
    zroff[q] = zp * 64 + zc * 4;
    // the nib blocks that share a dz tile all form it; each of (at most four of) them writes a share of it: with ONE writer its four
    // stores per chunk sat in front of its own prefetch loads (vector-memory operations retire in order): +3.5 ms on the launch
    const int nsh = nib < 4 ? nib : 4;
    zsto[q] = (cib < nsh && (q % nsh) == cib) ? zvoff[q] : 0x80000000u;
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
  auto set_cursor = [&]() {
    xdesc = g4_desc((const char*)(a.X + (long long)lgb * a.H * a.W * a.Cin) - xshift, ximg_bytes + xshift);
    zdesc = g4_desc(a.dZ + (long long)lgb * a.H * a.W * a.Cout, zimg_bytes);
    if constexpr (BNF) {
      zzdesc = g4_desc(a.Z + (long long)lgb * a.H * a.W * a.Cout, zimg_bytes);
      odesc = g4_desc(a.dZout + (long long)lgb * a.H * a.W * a.Cout, zimg_bytes);
    }
    xsoff = (unsigned)(((4 * lty) * a.W + 16 * lcx) * a.Cin * 4);
    zsoff = (unsigned)(((4 * lty) * a.W + 16 * lcx) * a.Cout * 4);
    xbt = (lty == 0 ? 1u << 28 : 0u) | (lty == a.gh - 1 ? 1u << 29 : 0u) | (lcx == 0 ? 1u << 30 : 0u) | (lcx == a.gw - 1 ? 1u << 31 : 0u);
  };
  auto advance = [&]() {
    if (lrem <= 0) return;
    --lrem;
    if (++lcx == a.gw) { lcx = 0; if (++lty == a.gh) { lty = 0; ++lgb; } }
    set_cursor();
  };
  f32x4 gx[4], gz[4], zz[4];
  unsigned osoff = 0;                           // BNF: scalar offset of the chunk held in gz / zz (its dz goes there)
  i32x4g_ odesc_held;
  auto Gx = [&](int q) { g4_load(gx[q], xdesc, (xhfl[q] & xbt) | xvoff[q], xsoff); };
  auto Gz = [&](int q) {
    g4_load(gz[q], zdesc, zvoff[q], zsoff);
    if constexpr (BNF) { if (CY_G4_DBG & 2) zz[q] = gz[q]; else g4_load(zz[q], zzdesc, zvoff[q], zsoff); }
  };
  // BNF: per-channel constants of this thread's 4 dz channels: dz = d * sc + (z - mu) * kb + kc
  f32x4 k_sc = {0.f, 0.f, 0.f, 0.f}, k_nmu = k_sc, k_b = k_sc, k_c = k_sc;
  if constexpr (BNF) {
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int ch = cob * 64 + (t & 15) * 4 + k;
      const float sc = a.scale[ch], is = a.invstd[ch];
      const float m1 = (float)(a.red[2 * ch] * a.inv_count), m2 = (float)(a.red[2 * ch + 1] * a.inv_count);
      k_sc[k] = sc; k_nmu[k] = -a.mean[ch]; k_b[k] = -sc * is * m2; k_c[k] = -sc * m1;
    }
  }
  auto Sx = [&](int q) { *(f32x4*)(Rx + xroff[q]) = gx[q]; };
  auto Sz = [&](int q) {
    if constexpr (BNF) {
      f32x4 o;
#pragma unroll
      for (int k = 0; k < 4; ++k) o[k] = (CY_G4_DBG & 4) ? gz[q][k] : __builtin_fmaf(gz[q][k], k_sc[k], __builtin_fmaf(zz[q][k] + k_nmu[k], k_b[k], k_c[k]));
      *(f32x4*)(Rz + zroff[q]) = o;
      if (!(CY_G4_DBG & 1)) g4_store(o, odesc_held, zsto[q], osoff);
    } else {
      *(f32x4*)(Rz + zroff[q]) = gz[q];
    }
  };

  // ---- ROLE 0: input transform, thread (ci pair r, tile kg, row half hr): rows 3 hr .. 3 hr + 2 of V = B^T d B for channels
  // 2 r, 2 r + 1 as float2 (winograd4.hip: per-lane coefficients alpha / gamma and a row-shifted base make the halves one code)
  const int tt_ = t & 127;
  const int vr = tt_ & 15, vkg = (tt_ >> 4) & 3, hr = tt_ >> 6;
  const int tbase = (4 * vkg) * G4_XPITCH + 2 * vr;
  const int tbase_e = tbase + hr * 18 * G4_XPITCH;
  const int vdst = (vkg * 16 + vr) * 4;
  const int vd0 = vdst + (hr ? 5 : 0) * 768, vd1 = vdst + (hr ? 3 : 1) * 768, vd2 = vdst + (hr ? 4 : 2) * 768;
  const float alpha_ = hr ? -1.f : -4.f, gamma_ = hr ? 2.f : 1.f;
  const f32x2 kal = {alpha_, alpha_}, kga = {gamma_, gamma_};
  const f32x2 k4 = {4.f, 4.f}, km5 = {-5.f, -5.f}, km4 = {-4.f, -4.f}, k2 = {2.f, 2.f};
  f32x2 tt[3][6], dc[2][7], cx_, cy_, ci_, rt_[6];
  auto Trd = [&](int c) {
    f32x2* d = dc[c & 1];
    d[0] = *(const f32x2*)(Rx + tbase_e + (0 * 18 + c) * G4_XPITCH);
    d[1] = *(const f32x2*)(Rx + tbase_e + (2 * 18 + c) * G4_XPITCH);
    d[2] = *(const f32x2*)(Rx + tbase_e + (4 * 18 + c) * G4_XPITCH);
    d[3] = *(const f32x2*)(Rx + tbase + (1 * 18 + c) * G4_XPITCH);
    d[4] = *(const f32x2*)(Rx + tbase + (2 * 18 + c) * G4_XPITCH);
    d[5] = *(const f32x2*)(Rx + tbase + (3 * 18 + c) * G4_XPITCH);
    d[6] = *(const f32x2*)(Rx + tbase + (4 * 18 + c) * G4_XPITCH);
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

  // ---- prologue.  State at the top of iteration f: V / Z[f&1] = chunk f, raw LDS = chunk f + 1, registers = chunk f + 2 (in
  // flight), load cursor at f + 3.
  set_cursor();
#pragma unroll
  for (int q = 0; q < 4; ++q) Gx(q);
#pragma unroll
  for (int q = 0; q < 4; ++q) Gz(q);
  if constexpr (BNF) { osoff = zsoff; odesc_held = odesc; }
  advance();
#pragma unroll
  for (int q = 0; q < 4; ++q) { g4_vmwait<0>(gx[q]); g4_vmwait<0>(gz[q]); if constexpr (BNF) g4_vmwait<0>(zz[q]); }
#pragma unroll
  for (int q = 0; q < 4; ++q) { Sx(q); Sz(q); }
#pragma unroll
  for (int q = 0; q < 4; ++q) Gx(q);
#pragma unroll
  for (int q = 0; q < 4; ++q) Gz(q);
  const unsigned osoff1 = zsoff;
  const i32x4g_ odesc1 = odesc;
  advance();
  __syncthreads();
  Tall(0);
  __syncthreads();
#pragma unroll
  for (int q = 0; q < 4; ++q) { g4_vmwait<0>(gx[q]); g4_vmwait<0>(gz[q]); if constexpr (BNF) g4_vmwait<0>(zz[q]); }
  if constexpr (BNF) { osoff = osoff1; odesc_held = odesc1; }
#pragma unroll
  for (int q = 0; q < 4; ++q) { Sx(q); Sz(q); }
#pragma unroll
  for (int q = 0; q < 4; ++q) Gx(q);                 // (the loop's issue order: its hand-counted waits hold for iteration 0 too)
#pragma unroll
  for (int q = 0; q < 4; ++q) Gz(q);
  if constexpr (BNF) { osoff = zsoff; odesc_held = odesc; }
  advance();
  __syncthreads();

  // fragments: A of position pair q: 16 bytes at ((q * 4 + kgl) * 16 + ml) * 4; B of position p: ((p * 4 + kgl) * 64 + col)
  const int kgl = lane >> 4, ml = lane & 15;
  const int fragA = (kgl * 16 + ml) * 4;
  const int fragB = kgl * 64 + ((16 * wave + ml) ^ ((kgl & 1) << 4));
  f32x4 fa[3];
  float fb[6];
  fa[0] = *(const f32x4*)(Vs + fragA);
  fa[1] = *(const f32x4*)(Vs + fragA + 256);
#pragma unroll
  for (int p = 0; p < 4; ++p) fb[p] = Zs[fragB + p * 256];

  f32x4 accA[32][2], accV[4][2];
#pragma unroll
  for (int p = 0; p < 32; ++p)
#pragma unroll
    for (int h = 0; h < 2; ++h) accA[p][h] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int p = 0; p < 4; ++p)
#pragma unroll
    for (int h = 0; h < 2; ++h) accV[p][h] = f32x4{0.f, 0.f, 0.f, 0.f};

  // vector-memory operations per iteration, in issue order: Gx0..3, then per dz item [dz store (BNF)], Gz, [Gzz (BNF)]; the prologue
  // issues its last round in the same order (without the stores).  S_raw(q) of iteration f + 1 waits for the load of iteration f
  // into the same register: exactly 7 younger operations without BNF; with BNF at least 10 (iteration 0: 11 for the x items,
  // 11 + q for d, 10 + q for z; later iterations 15 / 14 / 13) -- a lower bound of the younger operations is what a wait needs
  constexpr int VMW = BNF ? 10 : 7;
  for (int f = 0; f < nchunk; ++f) {
    const float* va_ = Vs + (f & 1) * G4_V_BUF + fragA;
    const float* zb_ = Zs + (f & 1) * G4_Z_BUF + fragB;
    float* vw_ = Vs + ((f + 1) & 1) * G4_V_BUF;
    float* zw_ = Zs + ((f + 1) & 1) * G4_Z_BUF;
#define G4SLOT(SIDX)                                                                                  \
    {                                                                                                 \
      constexpr int s_ = (SIDX), p_ = s_ >> 1, mt_ = s_ & 1, q_ = p_ >> 1;                            \
      if (s_ == G4_MID || s_ == G4_END) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); \
      if (p_ < 32) g4_mfma_a(accA[p_ < 32 ? p_ : 0][mt_], fa[q_ % 3][2 * (p_ & 1) + mt_], fb[p_ % 6]); \
      else g4_mfma_v(accV[p_ >= 32 ? p_ - 32 : 0][mt_], fa[q_ % 3][2 * (p_ & 1) + mt_], fb[p_ % 6]);   \
      if (s_ < G4_END) {                                                                              \
        if ((s_ & 3) == 0 && q_ + 2 < 18) fa[(q_ + 2) % 3] = *(const f32x4*)(va_ + (q_ + 2) * 256);   \
        if (mt_ == 0 && p_ + 4 < 36) fb[(p_ + 4) % 6] = zb_[(p_ + 4) * 256];                          \
      } else {                                  /* behind the end barrier: the next chunk's first fragments */ \
        if (s_ == 64) fa[0] = *(const f32x4*)(vw_ + fragA);                                           \
        if (s_ == 68) fa[1] = *(const f32x4*)(vw_ + fragA + 256);                                     \
        if (s_ == 64) fb[0] = zw_[fragB];                                                             \
        if (s_ == 65) fb[1] = zw_[fragB + 256];                                                       \
        if (s_ == 66) fb[2] = zw_[fragB + 512];                                                       \
        if (s_ == 68) fb[3] = zw_[fragB + 768];                                                       \
      }                                                                                               \
      constexpr int kind = g4_kind(ROLE, s_), k_ = g4_idx(ROLE, s_);                                  \
      if (kind == 1) {                                                                                \
        if constexpr (ROLE == 0) Trd(k_ % 6); else Zrd(k_ % 4);                                       \
      } else if (kind == 2) {                                                                         \
        if constexpr (ROLE == 0) Tcol((k_ >> 1) % 6, k_ & 1); else Zcol((k_ >> 1) % 4, k_ & 1);       \
      } else if (kind == 3) {                                                                         \
        if constexpr (ROLE == 0) Trow(vw_, (k_ >> 2) % 3, k_ & 3); else Zrow(zw_, (k_ >> 1) % 6, k_ & 1); \
      } else if (kind == 4) {                   /* S_raw: x items 0..3, then dz items 0..3 */        \
        if (k_ < 4) { g4_vmwait<VMW>(gx[k_ & 3]); Sx(k_ & 3); }                                       \
        else { g4_vmwait<VMW>(gz[k_ & 3]); if constexpr (BNF) g4_vmwait<VMW>(zz[k_ & 3]); Sz(k_ & 3); } \
      } else if (kind == 5) {                                                                         \
        if (k_ < 4) Gx(k_ & 3); else Gz(k_ & 3);                                                      \
      } else if (kind == 6) {                                                                         \
        if constexpr (BNF) { osoff = zsoff; odesc_held = odesc; }                                     \
        advance();                                                                                    \
      }                                                                                               \
      __builtin_amdgcn_sched_barrier(0);                                                              \
    }
#define G4SLOT8(B) G4SLOT((B)) G4SLOT((B) + 1) G4SLOT((B) + 2) G4SLOT((B) + 3) G4SLOT((B) + 4) G4SLOT((B) + 5) G4SLOT((B) + 6) G4SLOT((B) + 7)
    G4SLOT8(0) G4SLOT8(8) G4SLOT8(16) G4SLOT8(24) G4SLOT8(32) G4SLOT8(40) G4SLOT8(48) G4SLOT8(56) G4SLOT8(64)
#undef G4SLOT8
#undef G4SLOT
  }

  // the loads of the chunks behind the block's last one are still in flight and hipcc does not know: their registers must not be
  // reused before they have landed
#pragma unroll
  for (int q = 0; q < 4; ++q) { g4_vmwait<0>(gx[q]); g4_vmwait<0>(gz[q]); if constexpr (BNF) g4_vmwait<0>(zz[q]); }
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

// dW[co][ci][3][3] = A^T [ S (sum_ranges dU) S ] A with S = diag(1/4, -1/6, -1/6, 1/24, 1/24, 1), A^T = [[1,1,1,1,1,0],[0,1,-1,2,-2,0],[0,1,1,4,4,1]]
__global__ void wino4_wgrad_finish_kernel(const float* __restrict__ slab, float* __restrict__ dW, int nrange, int Cin, int Cout) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= Cin * Cout) return;
  const int co = idx % Cout, ci = idx / Cout;
  const float sc[6] = {0.25f, -1.f / 6, -1.f / 6, 1.f / 24, 1.f / 24, 1.f};
  float m[6][6];
#pragma unroll
  for (int i = 0; i < 6; ++i)
#pragma unroll
    for (int j = 0; j < 6; ++j) {
      float s = 0.f;
      for (int r = 0; r < nrange; ++r) s += slab[(((long long)r * 36 + (6 * i + j)) * Cin + ci) * Cout + co];
      m[i][j] = s * (sc[i] * sc[j]);
    }
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
  const long long tiles = (long long)B * (H / 4) * (W / 16);
  const int per = (Cin / 32) * (Cout / 64);
  long long n = ncu / per;
  if (const char* e = getenv("CY_G4_NRANGE")) n = atoll(e);            // developer override (tests of the range logic)
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
  if (bnf) wino4_wgrad_kernel<1><<<(unsigned)blocks, 256, lds, s>>>(a);
  else wino4_wgrad_kernel<0><<<(unsigned)blocks, 256, lds, s>>>(a);
  CY_LAUNCH_CHECK("cy_conv3x3_winograd4_wgrad");
  const int n = a.Cin * a.Cout;
  wino4_wgrad_finish_kernel<<<(unsigned)cy_ceil_div(n, 256), 256, 0, s>>>(a.slab, dW, a.nrange, a.Cin, a.Cout);
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
