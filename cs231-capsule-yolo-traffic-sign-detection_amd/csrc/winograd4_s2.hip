// Fused Winograd F(4x4, 2x2) forward for 4x4 / stride 2 / pad 1 convolutions on NHWC fp32 (DarkCapsuleNet conv_3..5,
// models.py:352-363), gfx950.
//
// As in winograd_s2.hip the layer is a 2x2 stride-1 "valid" convolution of the space-to-depth view shifted by one pixel,
//   X'(Y, X, (py, px, c)) = in(2Y - 1 + py, 2X - 1 + px, c),   out(y, x, co) = sum_{a,b} sum_q X'(y + a, x + b, q) g'(co, q, a, b),
//   g'(co, (py, px, c), a, b) = w[co][c][2a + py][2b + px],
// but with the minimal-filtering form F(4x4, 2x2): 25 multiplies per 4x4 outputs = 1.5625 per output against 2.25 for
// F(2x2, 2x2) and 4 for the direct form -- 1.44x fewer MFMAs than winograd_s2.hip.  Interpolation points 0, 1, -1, -2:
//   B^T = [[2,1,-2,-1,0],[0,2,3,1,0],[0,-2,1,1,0],[0,1,0,-1,0],[0,-2,-1,2,1]],  G = [[1/2,0],[1/6,1/6],[1/2,-1/2],[1/6,-1/3],[0,1]],
//   A^T = [[1,1,1,1,0],[0,1,-1,-2,0],[0,1,1,4,0],[0,1,-1,-8,1]]      (fp32 error 3e-6 against 1e-6 for F(2x2,2x2)).
//
// Structure = winograd4.hip (see there): block = 4 x 8 tiles (16 x 32 output pixels) x 64 output channels, 4 waves, one per SIMD;
// a wave owns all 25 positions of the 32 tiles for 16 output channels (25 x 2 tiles of v_mfma_f32_16x16x4_f32 = 200 AGPRs), the
// output transform is lane-local, the transformed weights stream L2 -> registers (13 position pairs resident, each reloaded for
// the next chunk right behind its last MFMA), loads are hand-waited inline asm, one barrier per chunk inside the MFMA stream.
// A chunk of the reduction is one (py, px) class and 8 input channels (100 MFMAs per wave); its 17 x 33 patch of X' is a
// stride-2 sampling of the input, read through a buffer descriptor whose base stands one row and one pixel in front of the
// image (padding = an offset beyond the descriptor's range: zeros).  Waves 0, 1 transform rows 0 and 4 of V = B^T d B (52 packed
// operations per thread and chunk), waves 2, 3 rows 1..3 (58): two copies of the whole loop behind one branch at the top.
// AFFINE: the input is lrelu(X * in_scale[c] + in_shift[c]) (the producer's BatchNorm + LeakyReLU, applied on the way into LDS;
// padding stays exactly 0 through a select on the item's saved validity).
#include <type_traits>
#include "common.h"

namespace {

constexpr int H4_PC = 33;                       // patch columns of X' (8 tiles x 4 + 1)
constexpr int H4_NPIX = 17 * 33;                // 561
constexpr int H4_RAWP = 577;                    // >= 561, = 1 (mod 16)
constexpr int H4_RAW_BUF = 2 * H4_RAWP * 4;     // floats: [kq][pixel][4]
constexpr int H4_V_BUF = 25 * 256;              // floats: [pos][kg][16 tile slots][tile half][2 k-steps]
constexpr int H4_NQ = 5;                        // patch float4 items per thread (1122 over 256 threads)
constexpr int H4_NP = 13;                       // position pairs (the 26th position does not exist: its B operand is zero, never used)
constexpr int H4_OG = 272, H4_OSTEP = 4 * H4_OG;
constexpr int H4_BAR = 92;                      // slot of the chunk's barrier (every fragment read of the chunk was issued at slot 88)

typedef int i32x4h_ __attribute__((ext_vector_type(4)));

struct Wino4S2Args {
  const float* X; const float* U; float* Y; const float* bias; double* stats;
  const float* in_scale; const float* in_shift; float in_slope;
  int B, H, W, Cin, Cout, Np, Ho, Wo, tbh, tbw;
  int ntiles;
  int sgroup;                                   // input gradient: grid tiles per group of 4 channel blocks (see tile_pos)
  float out_slope;
  // input-gradient mode (MODE 1): X = dY [B][H][W][Cin] (H, W, Cin = the layer's output map and channels), Y = dX [B][Hx][Wx][Cx] of the
  // layer, scattered from the (Ho x Wo = Hx/2+1 x Wx/2+1) grid of the space-to-depth view; Cout = Np = 4 Cx; optional BatchNorm-
  // backward sums of the producer block (cy_conv_gemm_t.bn_*): then d = dX * lrelu'(z * scale + shift) is what is stored
  int Hx, Wx, Cx;
  const float* bn_z; const float* bn_scale; const float* bn_shift; const float* bn_mean; const float* bn_invstd;
  double* bn_red; float bn_slope;
};

__device__ __forceinline__ void h4_mfma(f32x4& c, float a, float b) {
  asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+a"(c) : "v"(a), "v"(b));
}
__device__ __forceinline__ f32x2 h4_fma(f32x2 x, f32x2 y, f32x2 z) { return __builtin_elementwise_fma(x, y, z); }
template <int OFF> __device__ __forceinline__ void h4_bload(f32x4& dst, const char* base, unsigned voff) {
  asm volatile("global_load_dwordx4 %0, %1, %2 offset:%3" : "=v"(dst) : "v"(voff), "s"(base), "n"(OFF));
}
__device__ __forceinline__ void h4_rload(f32x4& dst, i32x4h_ desc, unsigned voff, unsigned soff) {
  asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen" : "=v"(dst) : "v"(voff), "s"(desc), "s"(soff));
}
template <int N> __device__ __forceinline__ void h4_vmwait(f32x4& x) { asm volatile("s_waitcnt vmcnt(%1)" : "+v"(x) : "n"(N)); }
__device__ __forceinline__ float h4_acc_elem(float a_elem) {
  float x;
  asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(x) : "a"(a_elem));
  return x;
}

// ---- compile-time schedule of one chunk: 100 slots; slot s issues the MFMA of position s >> 2, tile half s & 1, k-step (s >> 1) & 1.
// The A fragment of position p + 2 is fetched in slot 4p.  Pieces (one per slot):
//   1 RG    S_raw(k) + G_raw(k): one float4 of chunk f + 2 registers -> LDS, then the same register's load of chunk f + 3 (k = 0..4,
//           slots 1..5: in front of every B load, so that the hand-counted waits are the same in the first chunk of a block)
//   2 GB    B operand of position pair q of the NEXT chunk, right behind the pair's last MFMA
//   3 rd / 4 col / 5 row   the role's transform pieces        6 ADV  patch cursor to f + 4
struct H4Sched { int kind[100]; int idx[100]; };
constexpr H4Sched h4_make_sched(int role) {
  H4Sched s{};
  for (int i = 0; i < 100; ++i) { s.kind[i] = 0; s.idx[i] = 0; }
  for (int k = 0; k < H4_NQ; ++k) s.kind[1 + k] = 1;
  s.kind[6] = 6;
  for (int q = 0; q < H4_NP; ++q) s.kind[8 * q + 8 > 99 ? 99 : 8 * q + 8] = 2;
  // transform pieces: rd0 rd1, then per column its phases and the read of column + 2; then the rows
  int pk[48] = {}, n = 0;
  const int ncp = role == 0 ? 3 : 2, nrows = role == 0 ? 2 : 3;
  pk[n++] = 3; pk[n++] = 3;
  for (int c = 0; c < 5; ++c) {
    for (int p = 0; p < ncp; ++p) pk[n++] = 4;
    if (c < 3) pk[n++] = 3;
  }
  for (int r = 0; r < nrows * 3; ++r) pk[n++] = 5;
  int sl = 7;
  for (int i = 0; i < n; ++i) {
    while (sl < 100 && s.kind[sl] != 0) ++sl;
    s.kind[sl] = pk[i];
    sl += 2;                                    // every other slot: the pieces reach slot ~75, in front of the barrier
  }
  int cnt[8] = {};
  for (int i = 0; i < 100; ++i) { s.idx[i] = cnt[s.kind[i]]; cnt[s.kind[i]]++; }
  return s;
}
constexpr H4Sched H4S0 = h4_make_sched(0), H4S1 = h4_make_sched(1);
constexpr bool h4_sched_ok(const H4Sched& s, int role) {
  int cnt[8] = {};
  for (int i = 0; i < 100; ++i) cnt[s.kind[i]]++;
  for (int i = H4_BAR; i < 100; ++i) if (s.kind[i] >= 3 && s.kind[i] <= 5) return false;
  return cnt[1] == H4_NQ && cnt[2] == H4_NP && cnt[3] == 5 && cnt[4] == (role == 0 ? 15 : 10) && cnt[5] == (role == 0 ? 6 : 9) && cnt[6] == 1;
}
static_assert(h4_sched_ok(H4S0, 0) && h4_sched_ok(H4S1, 1), "winograd4_s2: chunk schedule incomplete");
constexpr int h4_kind(int role, int s) { return role == 0 ? H4S0.kind[s] : H4S1.kind[s]; }
constexpr int h4_idx(int role, int s) { return role == 0 ? H4S0.idx[s] : H4S1.idx[s]; }
constexpr int h4_vm_between(int role, int lo, int hi) {      // vector-memory operations issued in slots [lo, hi)
  int n = 0;
  for (int i = lo < 0 ? 0 : lo; i < hi && i < 100; ++i) n += (h4_kind(role, i) == 1 || h4_kind(role, i) == 2) ? 1 : 0;
  return n;
}
constexpr int h4_slot_of_gb(int q) { return 8 * q + 8 > 99 ? 99 : 8 * q + 8; }
// operations younger than the load of pair q (issued behind the pair's last MFMA of the previous chunk, or by the prologue in
// pair order behind its patch loads) when the pair's first MFMA (slot 8 q) issues
constexpr int h4_younger_b(int role, int q) { return (H4_NP - 1 - q) + h4_vm_between(role, 0, 8 * q); }
// ... and than the load of patch item k when RG(k) stores it one chunk later: 4 - k patch loads, 13 B loads, k patch loads
constexpr int h4_younger_r(int k) { return (H4_NQ - 1 - k) + H4_NP + k; }

// MODE 0: forward.  MODE 1 (2: with the producer's BatchNorm-backward sums): input gradient: dX'(Y, X, q) = sum_{a',b'} D(Y + a', X + b', co) h(q, co, a', b') with D(Y, X) = dY(Y - 1, X - 1)
// (zero outside) and h(q, co, a', b') = g'(co, q, 1 - a', 1 - b'): the same 2x2 "valid" convolution over the plain NHWC tensor dY shifted by one
// pixel, K = Cout of the layer (chunks of 8 output channels, one class), N = 4 Cin; the drain scatters (Y, X, q = (py, px, c)) to
// dX(2Y - 1 + py, 2X - 1 + px, c).
template <int EPI, bool AFFINE, int ROLE, int MODE>
__device__ __forceinline__ void h4_run(const Wino4S2Args& a, float* smem) {
  float* Vs = smem;                             // [2][H4_V_BUF]
  float* Rs = smem + 2 * H4_V_BUF;              // [2][H4_RAW_BUF]
  float* Os = Rs + 2 * H4_RAW_BUF;              // [4 waves][2 H4_OSTEP] drain scratch
  float* Aff = Os + 4 * 2 * H4_OSTEP;           // [2][Cin] (AFFINE)

  const int t = threadIdx.x, lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  if constexpr (AFFINE) {
    for (int i = t; i < a.Cin; i += 256) { Aff[i] = a.in_scale[i]; Aff[a.Cin + i] = a.in_shift[i]; }
    __syncthreads();
  }
  unsigned vid = blockIdx.x;
  if ((gridDim.x & 7u) == 0) vid = (blockIdx.x & 7u) * (gridDim.x >> 3) + (blockIdx.x >> 3);
  const int nblk = a.Np / 64;
  const int ntile_mine = (a.ntiles - (int)vid + (int)gridDim.x - 1) / (int)gridDim.x;
  const int cpp = a.Cin / 8;                    // chunks per (py, px) class
  const int nchunk = MODE == 0 ? 4 * cpp : cpp;
  struct TilePos { int nb, b, Y0, X0; };
  auto tile_pos = [&](int k) {
    const int id = (int)vid + k * (int)gridDim.x;
    TilePos p;
    int rest;
    if constexpr (MODE == 0) {
      p.nb = id % nblk;
      rest = id / nblk;
    } else {
      // input gradient: N = 4 Cin is 16 channel blocks for conv_3 and each streams its own 426 KB slice of U per tile -- with all 16 on
      // one XCD (32 consecutive ids = 2 grid tiles x 16 channel blocks) that is 6.8 MB through a 4 MB L2 (measured: 9.6 GB of L2 misses
      // per launch).  32 consecutive ids are therefore 4 channel blocks x sgroup (8 when it divides the grid tiles) tiles: 1.7 MB of U
      // and 8 patches of dY per XCD.
      const int nl = id & 3;
      int r = id >> 2;
      const int si = r % a.sgroup; r /= a.sgroup;
      const int nq = r % (nblk >> 2);
      p.nb = nq * 4 + nl;
      rest = (r / (nblk >> 2)) * a.sgroup + si;
    }
    const int tbx = rest % a.tbw; rest /= a.tbw;
    const int tby = rest % a.tbh;
    p.b = rest / a.tbh; p.Y0 = tby * 16; p.X0 = tbx * 32;
    return p;
  };

  // ---- patch loader (winograd_s2.hip's item order): item = t + 256 q, pixel = pix0 + 128 q, k-quad = (t >> 4) & 1
  const int kq_of_thread = (t >> 4) & 1;
  const int pix0 = (t >> 5) * 16 + (t & 15);
  const int roff0 = (kq_of_thread * H4_RAWP + pix0) * 4;
  const bool rlast_ok = pix0 + 128 * (H4_NQ - 1) < H4_NPIX;
  const int roff4 = roff0 + 512 * (rlast_ok ? H4_NQ - 1 : H4_NQ - 2);
  unsigned gvoff[H4_NQ], hfl[H4_NQ];            // offset of class (0, 0) from the shifted base; border flags (bits 28..31)
  i32x4h_ xdesc = {0, 0, 0, 0};
  const int xshift = (a.W + 1) * a.Cin * 4;
  const int img_bytes = a.H * a.W * a.Cin * 4;
  auto set_raw_tile = [&](int k) {
    const TilePos p = tile_pos(k);
    const unsigned long long xb = (unsigned long long)(uintptr_t)((const char*)(a.X + (long long)p.b * a.H * a.W * a.Cin) - xshift);
    // (readfirstlane: an "s" asm operand must be provably uniform, and the image index comes out of vector-float divisions)
    xdesc = i32x4h_{__builtin_amdgcn_readfirstlane((int)(unsigned)xb), __builtin_amdgcn_readfirstlane((int)(unsigned)((xb >> 32) & 0xffffu)),
                    __builtin_amdgcn_readfirstlane(img_bytes + xshift), 0x00020000};
#pragma unroll
    for (int q = 0; q < H4_NQ; ++q) {
      const int pix = pix0 + 128 * ((q < H4_NQ - 1 || rlast_ok) ? q : q - 1);
      const int pr = pix / H4_PC, pc = pix - pr * H4_PC;
      const int Yg = p.Y0 + pr, Xg = p.X0 + pc;      // grid point of X': input rows 2 Yg - 1 + py, columns 2 Xg - 1 + px
      if constexpr (MODE == 0) {
        gvoff[q] = (unsigned)(((2 * Yg * a.W + 2 * Xg) * a.Cin + kq_of_thread * 4) * 4);
        // py = 0 needs Yg >= 1 (and 2 Yg - 1 < H), py = 1 needs 2 Yg < H; the same for the columns; grid points behind Ho / Wo: never valid
        hfl[q] = ((Yg < 1 || 2 * Yg - 1 >= a.H) ? 1u << 28 : 0u) | ((2 * Yg >= a.H) ? 1u << 29 : 0u) |
                 ((Xg < 1 || 2 * Xg - 1 >= a.W) ? 1u << 30 : 0u) | ((2 * Xg >= a.W) ? 1u << 31 : 0u);
      } else {                                  // D(Yg, Xg) = dY(Yg - 1, Xg - 1): the shifted base takes the -1s
        gvoff[q] = (unsigned)(((Yg * a.W + Xg) * a.Cin + kq_of_thread * 4) * 4);
        hfl[q] = ((Yg < 1 || Yg > a.H) ? 1u << 28 : 0u) | ((Xg < 1 || Xg > a.W) ? 1u << 30 : 0u);
      }
    }
  };
  auto advance = [&](int& k, int& c) {
    if (c + 1 < nchunk) { ++c; return false; }
    if (k + 1 < ntile_mine) { ++k; c = 0; return true; }
    return false;
  };
  int kr = 0, cr = 0;                           // patch cursor (tile of this block, chunk = class * cpp + channel chunk)
  unsigned rsoff = 0, rbt = 0;                  // uniform: scalar offset and border mask of the cursor's chunk
  int rc0 = 0;                                  // first channel of the cursor's chunk
  auto set_raw_chunk = [&]() {
    if constexpr (MODE != 0) {
      rc0 = cr * 8;
      rsoff = (unsigned)__builtin_amdgcn_readfirstlane(cr * 32);
      rbt = (1u << 28) | (1u << 30);
      return;
    }
    const int cls = cr / cpp, cc = cr - cls * cpp;
    rc0 = cc * 8;
    // (readfirstlane: the quotient cr / cpp comes out of a vector-float sequence, and an "s" asm operand must be provably uniform)
    rsoff = (unsigned)__builtin_amdgcn_readfirstlane((((cls >> 1) * a.W + (cls & 1)) * a.Cin + cc * 8) * 4);
    rbt = (unsigned)__builtin_amdgcn_readfirstlane((int)(((cls >> 1) ? 1u << 29 : 1u << 28) | ((cls & 1) ? 1u << 31 : 1u << 30)));
  };
  f32x4 graw[H4_NQ];
  unsigned hv[H4_NQ];                           // AFFINE: (flags & class mask) of the item held in graw: nonzero = padding
  int sc0 = 0;                                  // AFFINE: first channel of the chunk held in graw
  auto Graw1 = [&](int q, f32x4& dst, unsigned& hvq) {
    if constexpr (AFFINE) { hvq = hfl[q] & rbt; h4_rload(dst, xdesc, hvq | gvoff[q], rsoff); }
    else h4_rload(dst, xdesc, (hfl[q] & rbt) | gvoff[q], rsoff);
  };
  f32x4 asc = {1.f, 1.f, 1.f, 1.f}, ash = {0.f, 0.f, 0.f, 0.f};
  auto set_affine = [&](int c0) {
    if constexpr (AFFINE) { asc = *(const f32x4*)(Aff + c0 + kq_of_thread * 4); ash = *(const f32x4*)(Aff + a.Cin + c0 + kq_of_thread * 4); }
  };
  auto Sraw1 = [&](float* rb, int q, const f32x4& src, unsigned hvq) {
    f32x4 v = src;
    if constexpr (AFFINE) {
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const float y = __builtin_fmaf(src[k], asc[k], ash[k]);
        v[k] = hvq == 0u ? fmaxf(y, y * a.in_slope) : 0.f;
      }
    }
    *(f32x4*)(rb + (q < H4_NQ - 1 ? roff0 + 512 * q : roff4)) = v;
  };

  // ---- B operand stream: U[nb][chunk][wave][pair 13][lane][4]
  const long long u_wave = H4_NP * 1024;
  unsigned ulane[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) ulane[k] = (unsigned)lane * 16u + 4096u * k;
  auto u_ptr = [&](int k, int c) { return (const char*)a.U + (((long long)tile_pos(k).nb * nchunk + c) * 4 + wave) * u_wave; };
  int ku = 0, cu = 0;
  f32x4 bq[H4_NP];

  // ---- transform item: tile (ty, tx), channel pair kg; ROLE 0: rows 0 and 4 of V, ROLE 1: rows 1, 2, 3
  const int tt_ = t & 127;
  // Thread -> item and the slot swizzle follow the LDS lane groups of MI355X_MICROARCH.md (tools/probe/lds_conflicts_wino4s2.py computes all
  // of this kernel's patterns).  ds_write_b64 / ds_write2st64_b64 and the ds_read2_b64 hipcc merges the transform's reads into serve 16
  // CONTIGUOUS lanes per cycle on 32 banks:
  //   * a lane stores one half of a 16-byte slot, so 16 contiguous lanes must be 8 slots (mod 8) x both tile halves: lane bits 0..1 = kg,
  //     2 = ttx & 1, 3 = tile half, slot XOR 2 kg (bits 1, 2 of the slot from kg, bit 0 from ttx);
  //   * the raw reads (16-float tile pitch, k-quads 4 floats apart mod 32) reach 16 of 32 banks whatever the map: 2-way, as little as
  //     possible, needs kg and ttx & 1 inside the 16 lanes;
  //   * the fragments' ds_read_b128 pairs lanes {0-3, 12-15} of k-group 2j with lanes {4-11} of k-group 2j + 1 on 64 banks: XOR 2 maps
  //     {4..11} onto itself and XOR 4 / 6 swap the two sets for the pair (2, 3): conflict-free.
  // (lane = kg + 4 ttx + 32 tty with XOR 4 kg, the first version, had the stores and, by the table, the fragment reads at 2-way:
  // SQ_LDS_BANK_CONFLICT 45 % of the LDS cycles; time was the same -- the LDS array is 25 % busy here.)
  const int kg = tt_ & 3, ttx = ((tt_ >> 2) & 1) | (((tt_ >> 4) & 3) << 1), tty = (((tt_ >> 3) & 1) << 1) | ((tt_ >> 6) & 1);
  const int tbase = ((kg >> 1) * H4_RAWP + (4 * tty) * H4_PC + 4 * ttx) * 4 + 2 * (kg & 1);
  const int tslot = ((tty & 1) * 8 + ttx) ^ (2 * kg);
  const int vdst = (kg * 16 + tslot) * 4 + (tty >> 1) * 2;      // + pos * 256
  const f32x2 k2 = {2.f, 2.f}, km2 = {-2.f, -2.f}, k3 = {3.f, 3.f};
  constexpr int NR = ROLE == 0 ? 2 : 3;         // rows of this role; row index of its k-th row:
  auto row_of = [](int k) { return ROLE == 0 ? (k == 0 ? 0 : 4) : k + 1; };
  f32x2 tt[NR][5], dc[2][5], ca_[3], cb_[2];
  auto Trd = [&](const float* rb, int c) {
    f32x2* d = dc[c & 1];
    if constexpr (ROLE == 0) {
#pragma unroll
      for (int r = 0; r < 5; ++r) d[r] = *(const f32x2*)(rb + tbase + (r * H4_PC + c) * 4);
    } else {
#pragma unroll
      for (int r = 1; r < 4; ++r) d[r] = *(const f32x2*)(rb + tbase + (r * H4_PC + c) * 4);
    }
  };
  auto Tcol = [&](int c, int part) {
    const f32x2* d = dc[c & 1];
    if constexpr (ROLE == 0) {                  // t0 = 2 d0 + d1 - 2 d2 - d3,  t4 = -2 d1 - d2 + 2 d3 + d4
      if (part == 0) { ca_[0] = h4_fma(k2, d[0], d[1]); ca_[1] = h4_fma(km2, d[1], d[4]); }
      else if (part == 1) { cb_[0] = h4_fma(km2, d[2], ca_[0]); cb_[1] = h4_fma(k2, d[3], ca_[1]); }
      else { tt[0][c] = cb_[0] - d[3]; tt[1][c] = cb_[1] - d[2]; }
    } else {                                    // t1 = 2 d1 + 3 d2 + d3,  t2 = -2 d1 + d2 + d3,  t3 = d1 - d3
      if (part == 0) { ca_[0] = h4_fma(k2, d[1], d[3]); ca_[1] = h4_fma(km2, d[1], d[3]); tt[2][c] = d[1] - d[3]; }
      else { tt[0][c] = h4_fma(k3, d[2], ca_[0]); tt[1][c] = ca_[1] + d[2]; }
    }
  };
  f32x2 rp_[4], rq_[2];
  auto Trow = [&](float* vb, int k, int part) { // the role's k-th row: v_j from x = tt[k][0..4]
    const f32x2* x = tt[k];
    float* v = vb + vdst + (5 * row_of(k)) * 256;
    if (part == 0) {
      rp_[0] = h4_fma(k2, x[0], x[1]);
      rp_[1] = h4_fma(k2, x[1], x[3]);
      rp_[2] = h4_fma(km2, x[1], x[3]);
      rp_[3] = h4_fma(km2, x[1], x[4]);
      *(f32x2*)(v + 3 * 256) = x[1] - x[3];
    } else if (part == 1) {
      rq_[0] = h4_fma(km2, x[2], rp_[0]);
      rq_[1] = h4_fma(k2, x[3], rp_[3]);
      *(f32x2*)(v + 1 * 256) = h4_fma(k3, x[2], rp_[1]);
      *(f32x2*)(v + 2 * 256) = rp_[2] + x[2];
    } else {
      *(f32x2*)(v + 0 * 256) = rq_[0] - x[3];
      *(f32x2*)(v + 4 * 256) = rq_[1] - x[2];
    }
  };
  auto Tall = [&](int buf_raw, int buf_v) {
    const float* rb = Rs + buf_raw * H4_RAW_BUF;
    float* vb = Vs + buf_v * H4_V_BUF;
    constexpr int NCP = ROLE == 0 ? 3 : 2;
#pragma unroll
    for (int c = 0; c < 5; ++c) {
      Trd(rb, c);
#pragma unroll
      for (int p = 0; p < NCP; ++p) Tcol(c, p);
    }
#pragma unroll
    for (int k = 0; k < NR; ++k)
#pragma unroll
      for (int p = 0; p < 3; ++p) Trow(vb, k, p);
  };

  // ---- prologue.  State at the top of stream position f: V[f&1] = position f, raw[(f+1)&1] = patch of f+1, graw = patch of f+2,
  // patch cursor at f+3; bq = the 13 pairs of f (loaded in pair order BEHIND the patch loads: the loop's wait counts hold)
  {
    f32x4 graw1[H4_NQ];
    unsigned hv1[H4_NQ];
    set_raw_tile(0);
    set_raw_chunk();
#pragma unroll
    for (int q = 0; q < H4_NQ; ++q) Graw1(q, graw[q], hv[q]);
    set_affine(rc0);
#pragma unroll
    for (int q = 0; q < H4_NQ; ++q) h4_vmwait<0>(graw[q]);
#pragma unroll
    for (int q = 0; q < H4_NQ; ++q) Sraw1(Rs, q, graw[q], hv[q]);
    if (advance(kr, cr)) set_raw_tile(kr);
    set_raw_chunk();
    const int c1 = rc0;
#pragma unroll
    for (int q = 0; q < H4_NQ; ++q) Graw1(q, graw1[q], hv1[q]);
    __syncthreads();
    Tall(0, 0);
    set_affine(c1);
#pragma unroll
    for (int q = 0; q < H4_NQ; ++q) h4_vmwait<0>(graw1[q]);
#pragma unroll
    for (int q = 0; q < H4_NQ; ++q) Sraw1(Rs + H4_RAW_BUF, q, graw1[q], hv1[q]);
    if (advance(kr, cr)) set_raw_tile(kr);
    set_raw_chunk();
    sc0 = rc0;
#pragma unroll
    for (int q = 0; q < H4_NQ; ++q) Graw1(q, graw[q], hv[q]);
    __syncthreads();
    if (advance(kr, cr)) set_raw_tile(kr);
    set_raw_chunk();
  }
  const char* up_cur = u_ptr(0, 0);
  if (advance(ku, cu)) {}
  const char* up_nxt = u_ptr(ku, cu);
#pragma unroll
  for (int q = 0; q < H4_NP; ++q) h4_bload<0>(bq[q], up_cur + (q & 3) * 1024, ulane[q >> 2]);

  const int kgl = lane >> 4, ml = lane & 15;
  const int fragA = (kgl * 16 + (ml ^ (2 * kgl))) * 4;
  f32x4 fa[5];                                  // A fragments of positions p % 5 (25 positions: the ring's phase is the same in every chunk)
  fa[0] = *(const f32x4*)(Vs + fragA);
  fa[1] = *(const f32x4*)(Vs + fragA + 256);
  int c_next = 0;
  const f32x2 kd2 = {2.f, 2.f}, kd4 = {4.f, 4.f}, kd8 = {8.f, 8.f};
  // input-gradient mode: the producer's BatchNorm-backward sums stay in the lanes' registers over all tiles of the block's channel
  // block (lane = 4 channels of one of 16 pixels) and go out through wave shuffles and double atomics when it changes / at the end
  f32x4 b1 = {0.f, 0.f, 0.f, 0.f}, b2 = b1;
  int bn_nb = -1;
  auto flush_bn = [&](int nbx) {
    const int q0x = nbx * 64, cb0x = q0x - (q0x / a.Cx) * a.Cx;
#pragma unroll
    for (int k = 0; k < 4; ++k)
#pragma unroll
      for (int msk = 4; msk < 64; msk <<= 1) { b1[k] += __shfl_xor(b1[k], msk, 64); b2[k] += __shfl_xor(b2[k], msk, 64); }
    if (lane < 4) {
      double* rd = a.bn_red + (size_t)(blockIdx.x % CY_STATS_COPIES) * a.Cx * 2;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int ch = cb0x + wave * 16 + lane * 4 + k;
        atomicAdd(rd + 2 * ch, (double)b1[k]);
        atomicAdd(rd + 2 * ch + 1, (double)b2[k]);
      }
    }
    b1 = f32x4{0.f, 0.f, 0.f, 0.f}; b2 = b1;
  };
  for (int km = 0; km < ntile_mine; ++km) {
    f32x4 acc[25][2];
#pragma unroll
    for (int p = 0; p < 25; ++p)
#pragma unroll
      for (int h = 0; h < 2; ++h) acc[p][h] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int cm = 0; cm < nchunk; ++cm, ++c_next) {
      const int c = c_next;
      const float* va_ = Vs + (c & 1) * H4_V_BUF + fragA;
      const float* rb_ = Rs + ((c + 1) & 1) * H4_RAW_BUF;
      float* vw_ = Vs + ((c + 1) & 1) * H4_V_BUF;
      float* rw_ = Rs + (c & 1) * H4_RAW_BUF;
#define H4SLOT(SIDX)                                                                                  \
      {                                                                                               \
        constexpr int s_ = (SIDX), p_ = s_ >> 2, w_ = s_ & 3, h_ = w_ & 1, ks_ = w_ >> 1, q_ = p_ >> 1; \
        if (s_ == H4_BAR) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");             \
        if (w_ == 0 && (p_ & 1) == 0) h4_vmwait<h4_younger_b(ROLE, q_)>(bq[q_]);                      \
        h4_mfma(acc[p_][h_], fa[p_ % 5][2 * h_ + ks_], bq[q_][2 * (p_ & 1) + ks_]);                   \
        if (w_ == 0 && p_ + 2 < 25) fa[(p_ + 2) % 5] = *(const f32x4*)(va_ + (p_ + 2) * 256);         \
        if (s_ == H4_BAR) fa[0] = *(const f32x4*)(vw_ + fragA);                                       \
        if (s_ == H4_BAR + 4) fa[1] = *(const f32x4*)(vw_ + fragA + 256);                             \
        constexpr int kind = h4_kind(ROLE, s_), k_ = h4_idx(ROLE, s_);                                \
        if (kind == 1) {                                                                              \
          if (k_ == 0) set_affine(sc0);                                                               \
          h4_vmwait<h4_younger_r(k_ % H4_NQ)>(graw[k_ % H4_NQ]);                                      \
          Sraw1(rw_, k_ % H4_NQ, graw[k_ % H4_NQ], hv[k_ % H4_NQ]);                                   \
          Graw1(k_ % H4_NQ, graw[k_ % H4_NQ], hv[k_ % H4_NQ]);                                        \
        } else if (kind == 2) {                 /* pair k_ of the next chunk into the register pair k_ just left */ \
          h4_bload<(k_ & 3) * 1024>(bq[k_ % H4_NP], up_nxt, ulane[(k_ % H4_NP) >> 2]);                \
        } else if (kind == 3) {                                                                       \
          Trd(rb_, k_ % 5);                                                                           \
        } else if (kind == 4) {                                                                       \
          Tcol((k_ / (ROLE == 0 ? 3 : 2)) % 5, k_ % (ROLE == 0 ? 3 : 2));                             \
        } else if (kind == 5) {                                                                       \
          Trow(vw_, (k_ / 3) % NR, k_ % 3);                                                           \
        } else if (kind == 6) {                 /* graw now holds f+3 (its channel: rc0); patch cursor -> f+4 */ \
          sc0 = rc0;                                                                                  \
          if (advance(kr, cr)) set_raw_tile(kr);                                                      \
          set_raw_chunk();                                                                            \
        }                                                                                             \
        __builtin_amdgcn_sched_barrier(0);                                                            \
      }
#define H4SLOT10(B) H4SLOT((B)) H4SLOT((B) + 1) H4SLOT((B) + 2) H4SLOT((B) + 3) H4SLOT((B) + 4) H4SLOT((B) + 5) H4SLOT((B) + 6) H4SLOT((B) + 7) H4SLOT((B) + 8) H4SLOT((B) + 9)
      H4SLOT10(0) H4SLOT10(10) H4SLOT10(20) H4SLOT10(30) H4SLOT10(40) H4SLOT10(50) H4SLOT10(60) H4SLOT10(70) H4SLOT10(80) H4SLOT10(90)
#undef H4SLOT10
#undef H4SLOT
      up_cur = up_nxt;
      if (advance(ku, cu)) {}
      up_nxt = u_ptr(ku, cu);
    }
    if constexpr (MODE != 0) {
      // ======== input gradient: the same A^T M A; grid point (Yg, Xg) of q block nb = one (py, px) class and 64 channels of it
      const TilePos tp = tile_pos(km);
      float* ow = Os + wave * (2 * H4_OSTEP);
      const int g_ = lane >> 4, co16 = lane & 15;
      const int q0 = tp.nb * 64, ph = q0 / a.Cx, cb0 = q0 - ph * a.Cx, py = ph >> 1, px = ph & 1;
      const int pxl = lane >> 2, cq = lane & 3, ly = pxl >> 2, lx = pxl & 3;
      const int cch = cb0 + wave * 16 + cq * 4;                 // first of this lane's 4 channels in the read-back
      constexpr bool bnb = MODE == 2;        // (a template parameter: the z loads and their waits are then on the same paths)
      if (bnb && bn_nb >= 0 && bn_nb != tp.nb) flush_bn(bn_nb);   // (uniform, rare) another channel block: hand over the sums so far
      bn_nb = tp.nb;
      f32x4 bsc = {0.f, 0.f, 0.f, 0.f}, bsh = bsc, bnm = bsc, bis = bsc;
      if constexpr (bnb) {                                // (tracked loads: they are waited for right here, before the chunk loop can see them pending)
        bsc = *(const f32x4*)(a.bn_scale + cch); bsh = *(const f32x4*)(a.bn_shift + cch);
        bis = *(const f32x4*)(a.bn_invstd + cch);
        bnm = -*(const f32x4*)(a.bn_mean + cch) * bis;
        asm volatile("s_waitcnt vmcnt(0)" : "+v"(bsc), "+v"(bsh), "+v"(bis), "+v"(bnm));
      }
      const int rowstride = 2 * a.Wx * a.Cx, colstride = 2 * a.Cx;          // floats per grid row / column
      const unsigned lane_off = (unsigned)(ly * rowstride + lx * colstride + cq * 4);
      const long long tile_off = ((long long)tp.b * a.Hx + (2 * tp.Y0 - 1 + py)) * a.Wx * a.Cx + (long long)(2 * tp.X0 - 1 + px) * a.Cx + cb0 + wave * 16;
      float* ytile = a.Y + tile_off;
      const float* ztile = a.bn_z + tile_off;
      // valid grid rows / columns: iy = 2 Yg - 1 + py in [0, Hx) <=> 1 - py <= Yg < Ho - py
      const bool full = tp.Y0 >= 1 - py && tp.Y0 + 16 <= a.Ho - py && tp.X0 >= 1 - px && tp.X0 + 32 <= a.Wo - px;
#pragma unroll
      for (int sp = 0; sp < 4; ++sp) {
        const int h = sp >> 1, r0 = 2 * (sp & 1);
        f32x4 zq[2][4];
        if constexpr (bnb) {                              // the tile pair's z values: requested before the transform runs
#pragma unroll
          for (int rr = 0; rr < 2; ++rr)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              const int Tj = 16 * h + 4 * j + r0 + rr;
              const int Yg = tp.Y0 + 4 * (Tj >> 3) + ly, Xg = tp.X0 + 4 * (Tj & 7) + lx;
              const bool ok = full || (Yg >= 1 - py && Yg < a.Ho - py && Xg >= 1 - px && Xg < a.Wo - px);
              // (no branch around the load: lanes outside the image read the tensor's first values and never use them)
              const float* zp = ok ? ztile + (4 * (Tj >> 3)) * rowstride + (4 * (Tj & 7)) * colstride + lane_off : a.bn_z;
              // (no `nt`: a wave reads 64 of a line's 128 bytes, the wave next to it the other 64 a moment later -- as a streaming
              // access the line had left L2 by then: 9.4 GB fetched for 5.67 GB of z, and 0.4 ms of conv_3's 5.6)
              asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(zq[rr][j]) : "v"(zp));
            }
        }
        f32x2 S[5][4];
#pragma unroll
        for (int i = 0; i < 5; ++i) {
          f32x2 m[5];
#pragma unroll
          for (int j = 0; j < 5; ++j) { m[j][0] = h4_acc_elem(acc[5 * i + j][h][r0]); m[j][1] = h4_acc_elem(acc[5 * i + j][h][r0 + 1]); }
          const f32x2 s12 = m[1] + m[2], d12 = m[1] - m[2];
          S[i][0] = (m[0] + s12) + m[3];
          S[i][1] = h4_fma(-kd2, m[3], d12);
          S[i][2] = h4_fma(kd4, m[3], s12);
          S[i][3] = h4_fma(-kd8, m[3], d12) + m[4];
        }
#pragma unroll
        for (int x = 0; x < 4; ++x) {
          const f32x2 s12 = S[1][x] + S[2][x], d12 = S[1][x] - S[2][x];
          f32x2 y[4];
          y[0] = (S[0][x] + s12) + S[3][x];
          y[1] = h4_fma(-kd2, S[3][x], d12);
          y[2] = h4_fma(kd4, S[3][x], s12);
          y[3] = h4_fma(-kd8, S[3][x], d12) + S[4][x];
#pragma unroll
          for (int yy = 0; yy < 4; ++yy) {
            ow[g_ * H4_OG + (yy * 4 + x) * 16 + co16] = y[yy][0];
            ow[H4_OSTEP + g_ * H4_OG + (yy * 4 + x) * 16 + co16] = y[yy][1];
          }
        }
        if constexpr (bnb) {
#pragma unroll
          for (int rr = 0; rr < 2; ++rr) asm volatile("s_waitcnt vmcnt(0)" : "+v"(zq[rr][0]), "+v"(zq[rr][1]), "+v"(zq[rr][2]), "+v"(zq[rr][3]));
        }
#pragma unroll
        for (int rr = 0; rr < 2; ++rr)
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const int Tj = 16 * h + 4 * j + r0 + rr;
            const f32x4 v = *(const f32x4*)(ow + rr * H4_OSTEP + j * H4_OG + pxl * 16 + cq * 4);
            const int Yg = tp.Y0 + 4 * (Tj >> 3) + ly, Xg = tp.X0 + 4 * (Tj & 7) + lx;
            const bool ok = full || (Yg >= 1 - py && Yg < a.Ho - py && Xg >= 1 - px && Xg < a.Wo - px);
            float* yp = ytile + (4 * (Tj >> 3)) * rowstride + (4 * (Tj & 7)) * colstride + lane_off;
            if (ok) {
              f32x4 sv = v;
              if constexpr (bnb) {
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                  const float z = zq[rr][j][k];
                  const float yv = __builtin_fmaf(z, bsc[k], bsh[k]);
                  const float d = yv > 0.f ? v[k] : v[k] * a.bn_slope;
                  sv[k] = d;
                  b1[k] += d;
                  b2[k] = __builtin_fmaf(d, __builtin_fmaf(z, bis[k], bnm[k]), b2[k]);
                }
              }
              __builtin_nontemporal_store(sv, (f32x4*)yp);
            }
          }
      }
    } else {
    // ======== tile km is complete: lane-local output transform (25 -> 16 per tile and channel), two accumulator rows at a time
    const TilePos tp = tile_pos(km);
    float* ow = Os + wave * (2 * H4_OSTEP);
    const int g_ = lane >> 4, co16 = lane & 15;
    const int co = tp.nb * 64 + wave * 16 + co16;
    float bv = 0.f;
    if (a.bias != nullptr) {
      const float* bp = a.bias + (co < a.Cout ? co : 0);
      asm volatile("global_load_dword %0, %1, off\n\ts_waitcnt vmcnt(0)" : "=&v"(bv) : "v"(bp));
      if (co >= a.Cout) bv = 0.f;
    }
    const bool full = tp.Y0 + 16 <= a.Ho && tp.X0 + 32 <= a.Wo && tp.nb * 64 + 64 <= a.Cout && (a.Cout & 3) == 0;
    const int pxl = lane >> 2, cq = lane & 3;
    const int cbase = tp.nb * 64 + wave * 16 + cq * 4;
    const unsigned lane_off = (unsigned)(((pxl >> 2) * a.Wo + (pxl & 3)) * a.Cout + cq * 4);
    float* ybase = a.Y + (((long long)tp.b * a.Ho + tp.Y0) * a.Wo + tp.X0) * a.Cout + tp.nb * 64 + wave * 16;
    const f32x2 bv2 = {bv, bv};
    f32x2 ssum2 = {0.f, 0.f}, ssq2 = {0.f, 0.f};
    float esum = 0.f, esq = 0.f;                // statistics of blocks that reach over the map's edge
#pragma unroll
    for (int sp = 0; sp < 4; ++sp) {
      const int h = sp >> 1, r0 = 2 * (sp & 1);
      f32x2 S[5][4];
#pragma unroll
      for (int i = 0; i < 5; ++i) {
        f32x2 m[5];
#pragma unroll
        for (int j = 0; j < 5; ++j) { m[j][0] = h4_acc_elem(acc[5 * i + j][h][r0]); m[j][1] = h4_acc_elem(acc[5 * i + j][h][r0 + 1]); }
        const f32x2 s12 = m[1] + m[2], d12 = m[1] - m[2];
        S[i][0] = (m[0] + s12) + m[3];
        S[i][1] = h4_fma(-kd2, m[3], d12);
        S[i][2] = h4_fma(kd4, m[3], s12);
        S[i][3] = h4_fma(-kd8, m[3], d12) + m[4];
      }
#pragma unroll
      for (int x = 0; x < 4; ++x) {
        const f32x2 s12 = (S[1][x] + S[2][x]) + bv2, d12 = (S[1][x] - S[2][x]) + bv2;
        f32x2 y[4];
        y[0] = (S[0][x] + s12) + S[3][x];
        y[1] = h4_fma(-kd2, S[3][x], d12);
        y[2] = h4_fma(kd4, S[3][x], s12);
        y[3] = h4_fma(-kd8, S[3][x], d12) + S[4][x];
#pragma unroll
        for (int yy = 0; yy < 4; ++yy) {
          f32x2 v = y[yy];
          if constexpr (EPI == 2) { v[0] = fmaxf(v[0], v[0] * a.out_slope); v[1] = fmaxf(v[1], v[1] * a.out_slope); }
          ow[g_ * H4_OG + (yy * 4 + x) * 16 + co16] = v[0];
          ow[H4_OSTEP + g_ * H4_OG + (yy * 4 + x) * 16 + co16] = v[1];
          if constexpr (EPI == 1) {
            if (full) { ssum2 = ssum2 + v; ssq2 = h4_fma(v, v, ssq2); }
            else {                             // (scalar sums: conditional updates of vector elements sent hipcc's InstCombine into a loop)
              const int T0 = 16 * h + 4 * g_ + r0, T1 = T0 + 1;
              const float v0 = v[0], v1 = v[1];
              if (co < a.Cout && tp.Y0 + 4 * (T0 >> 3) + yy < a.Ho && tp.X0 + 4 * (T0 & 7) + x < a.Wo) { esum += v0; esq = __builtin_fmaf(v0, v0, esq); }
              if (co < a.Cout && tp.Y0 + 4 * (T1 >> 3) + yy < a.Ho && tp.X0 + 4 * (T1 & 7) + x < a.Wo) { esum += v1; esq = __builtin_fmaf(v1, v1, esq); }
            }
          }
        }
      }
#pragma unroll
      for (int rr = 0; rr < 2; ++rr)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int Tj = 16 * h + 4 * j + r0 + rr;
          const f32x4 v = *(const f32x4*)(ow + rr * H4_OSTEP + j * H4_OG + pxl * 16 + cq * 4);
          float* yp = ybase + ((long long)(4 * (Tj >> 3)) * a.Wo + 4 * (Tj & 7)) * a.Cout + lane_off;
          if (full) {
            __builtin_nontemporal_store(v, (f32x4*)yp);
          } else {
            const int oy = tp.Y0 + 4 * (Tj >> 3) + (pxl >> 2), ox = tp.X0 + 4 * (Tj & 7) + (pxl & 3);
            if (oy < a.Ho && ox < a.Wo) {
              if ((a.Cout & 3) == 0 && cbase + 3 < a.Cout) *(f32x4*)yp = v;
              else {
#pragma unroll
                for (int k = 0; k < 4; ++k) if (cbase + k < a.Cout) yp[k] = v[k];
              }
            }
          }
        }
    }
    if constexpr (EPI == 1) {
      float ssum = (ssum2[0] + ssum2[1]) + esum, ssq = (ssq2[0] + ssq2[1]) + esq;
      ssum += __shfl_xor(ssum, 16, 64); ssq += __shfl_xor(ssq, 16, 64);
      ssum += __shfl_xor(ssum, 32, 64); ssq += __shfl_xor(ssq, 32, 64);
      if (lane < 16 && co < a.Cout) {
        double* st = a.stats + (size_t)(blockIdx.x % CY_STATS_COPIES) * a.Cout * 2;
        atomicAdd(st + 2 * co, (double)ssum);
        atomicAdd(st + 2 * co + 1, (double)ssq);
      }
    }
    }
  }
  if constexpr (MODE == 2) { if (bn_nb >= 0) flush_bn(bn_nb); }
  // the prefetch behind the block's last chunk is still in flight: its registers must not be reused before it has landed
#pragma unroll
  for (int q = 0; q < H4_NQ; ++q) h4_vmwait<0>(graw[q]);
#pragma unroll
  for (int q = 0; q < H4_NP; ++q) h4_vmwait<0>(bq[q]);
}

template <int EPI, bool AFFINE, int MODE = 0>
__global__ __launch_bounds__(256, 1) void wino4s2_conv_kernel(Wino4S2Args a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  if (threadIdx.x < 128) h4_run<EPI, AFFINE, 0, MODE>(a, smem);
  else h4_run<EPI, AFFINE, 1, MODE>(a, smem);
}

// input-gradient operand: U[nb (q block)][chunk (co / 8)][wave][pair][lane][e]: q = 64 nb + 16 wave + (l & 15) = (py, px, c),
// co = 8 chunk + 2 (l >> 4) + s; value (G h G^T)[i][j], h[a'][b'] = W[co][c][2 (1 - a') + py][2 (1 - b') + px]
__global__ void wino4s2_pack_dgrad_kernel(const float* __restrict__ W, float* __restrict__ U, int Cout, int Cin, long long total) {
  const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;   // one (co, q) per thread
  if (idx >= total) return;
  const int Np = 4 * Cin;
  const int q = (int)(idx % Np), co = (int)(idx / Np);
  const int ph = q / Cin, c = q - ph * Cin, py = ph >> 1, px = ph & 1;
  float g[2][2];
#pragma unroll
  for (int aa = 0; aa < 2; ++aa)
#pragma unroll
    for (int bb = 0; bb < 2; ++bb) g[aa][bb] = W[(((long long)co * Cin + c) * 4 + (2 * (1 - aa) + py)) * 4 + (2 * (1 - bb) + px)];
  const float Gm[5][2] = {{0.5f, 0.f}, {1.f / 6, 1.f / 6}, {0.5f, -0.5f}, {1.f / 6, -1.f / 3}, {0.f, 1.f}};
  const int nb = q >> 6, wv = (q >> 4) & 3, c16 = q & 15;
  const int nchunk = Cout / 8, cc = co >> 3, kgp = (co >> 1) & 3, s = co & 1;
  float* out = U + ((((long long)nb * nchunk + cc) * 4 + wv) * H4_NP) * 256 + (kgp * 16 + c16) * 4;
#pragma unroll
  for (int i = 0; i < 5; ++i)
#pragma unroll
    for (int j = 0; j < 5; ++j) {
      const int pos = 5 * i + j;
      float u = 0.f;
#pragma unroll
      for (int p = 0; p < 2; ++p)
#pragma unroll
        for (int r = 0; r < 2; ++r) u += Gm[i][p] * g[p][r] * Gm[j][r];
      out[(pos >> 1) * 256 + 2 * (pos & 1) + s] = u;
    }
  out[12 * 256 + 2 + s] = 0.f;
}

// U[nb][f = cls * Cin/8 + cc][wave][pair q][lane l][e]: pos = 2 q + (e >> 1) = 5 i + j, s = e & 1, co = 64 nb + 16 wave + (l & 15),
// c = 8 cc + 2 (l >> 4) + s; value (G g' G^T)[i][j] with g'[a][b] = W[co][c][2a + py][2b + px], cls = 2 py + px (0 for pos 25)
__global__ void wino4s2_pack_kernel(const float* __restrict__ W, float* __restrict__ U, int Cout, int Cin, int Np, long long total) {
  const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;   // one (cls, c, co) per thread
  if (idx >= total) return;
  const int co = (int)(idx % Np);
  long long r = idx / Np;
  const int c = (int)(r % Cin), cls = (int)(r / Cin);
  const int py = cls >> 1, px = cls & 1;
  float g[2][2] = {{0.f, 0.f}, {0.f, 0.f}};
  if (co < Cout) {
#pragma unroll
    for (int aa = 0; aa < 2; ++aa)
#pragma unroll
      for (int bb = 0; bb < 2; ++bb) g[aa][bb] = W[(((long long)co * Cin + c) * 4 + (2 * aa + py)) * 4 + (2 * bb + px)];
  }
  const float Gm[5][2] = {{0.5f, 0.f}, {1.f / 6, 1.f / 6}, {0.5f, -0.5f}, {1.f / 6, -1.f / 3}, {0.f, 1.f}};
  const int nb = co >> 6, wv = (co >> 4) & 3, c16 = co & 15;
  const int cpp = Cin / 8, cc = c >> 3, kgp = (c >> 1) & 3, s = c & 1;
  float* out = U + ((((long long)nb * (4 * cpp) + cls * cpp + cc) * 4 + wv) * H4_NP) * 256 + (kgp * 16 + c16) * 4;
#pragma unroll
  for (int i = 0; i < 5; ++i)
#pragma unroll
    for (int j = 0; j < 5; ++j) {
      const int pos = 5 * i + j;
      float u = 0.f;
#pragma unroll
      for (int p = 0; p < 2; ++p)
#pragma unroll
        for (int q = 0; q < 2; ++q) u += Gm[i][p] * g[p][q] * Gm[j][q];
      out[(pos >> 1) * 256 + 2 * (pos & 1) + s] = u;
    }
  out[12 * 256 + 2 + s] = 0.f;                  // position 25
}

}  // namespace

extern "C" int cy_wino4s2_ok(int B, int H, int W, int Cin, int Cout) {
  return B > 0 && H > 0 && W > 0 && Cout > 0 && (H & 1) == 0 && (W & 1) == 0 && Cin % 8 == 0 && Cin >= 8 &&
         (long long)H * W * Cin * 4 + (long long)(W + 1) * Cin * 4 < (1ll << 28);
}

extern "C" long long cy_wino4s2_packed_floats(int Cin, int N) { return (long long)4 * Cin * 26 * ((N + 63) / 64 * 64); }

extern "C" int cy_wino4s2_pack_weights(const float* W, float* U, int Cout, int Cin, void* stream) {
  CY_REQUIRE(W && U && Cout > 0 && Cin > 0 && Cin % 8 == 0, "cy_wino4s2_pack_weights: bad arguments");
  const int Np = (Cout + 63) / 64 * 64;
  const long long total = 4ll * Cin * Np;
  wino4s2_pack_kernel<<<(unsigned)cy_ceil_div(total, 256), 256, 0, (hipStream_t)stream>>>(W, U, Cout, Cin, Np, total);
  CY_LAUNCH_CHECK("cy_wino4s2_pack_weights");
  return 0;
}

extern "C" int cy_conv4x4s2_winograd4(const float* X, const float* U, float* Y, const float* bias, double* stats,
                                      const float* in_scale, const float* in_shift, float in_slope, float out_slope,
                                      int B, int H, int W, int Cin, int Cout, void* stream) {
  CY_REQUIRE(X && U && Y, "cy_conv4x4s2_winograd4: bad arguments");
  CY_REQUIRE(cy_wino4s2_ok(B, H, W, Cin, Cout), "cy_conv4x4s2_winograd4: shape B=%d H=%d W=%d Cin=%d Cout=%d not supported", B, H, W, Cin, Cout);
  CY_REQUIRE(out_slope >= 0.f && out_slope <= 1.f, "cy_conv4x4s2_winograd4: out_slope=%g must be in [0, 1]", (double)out_slope);
  CY_REQUIRE(out_slope == 1.f || (stats == nullptr && in_scale == nullptr), "cy_conv4x4s2_winograd4: the activation epilogue is for eval-mode forwards");
  CY_REQUIRE((in_scale == nullptr) == (in_shift == nullptr), "cy_conv4x4s2_winograd4: in_scale and in_shift come together");
  CY_REQUIRE(in_scale == nullptr || (in_slope > 0.f && in_slope <= 1.f), "cy_conv4x4s2_winograd4: in_slope=%g must be in (0, 1]", (double)in_slope);
  CY_REQUIRE((((uintptr_t)X | (uintptr_t)U | (uintptr_t)Y) & 15) == 0, "cy_conv4x4s2_winograd4: operands must be 16-byte aligned");
  CY_REQUIRE((long long)(H / 2) * (W / 2) * Cout < (1ll << 29), "cy_conv4x4s2_winograd4: output too large for 32-bit offsets");
  Wino4S2Args a{};
  a.X = X; a.U = U; a.Y = Y; a.bias = bias; a.stats = stats; a.in_scale = in_scale; a.in_shift = in_shift; a.in_slope = in_slope;
  a.out_slope = out_slope;
  a.B = B; a.H = H; a.W = W; a.Cin = Cin; a.Cout = Cout; a.Np = (Cout + 63) / 64 * 64; a.Ho = H / 2; a.Wo = W / 2;
  a.tbh = (a.Ho + 15) / 16; a.tbw = (a.Wo + 31) / 32;
  const long long tiles = (long long)B * a.tbh * a.tbw * (a.Np / 64);
  CY_REQUIRE(tiles < (1ll << 31), "cy_conv4x4s2_winograd4: too many tiles");
  a.ntiles = (int)tiles;
  int dev = 0, ncu = 0;
  hipError_t he = hipGetDevice(&dev);
  if (he == hipSuccess) he = hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev);
  if (he != hipSuccess || ncu <= 0) return cy_set_error((int)he, "cy_conv4x4s2_winograd4: cannot query the CU count: %s", hipGetErrorString(he));
  const long long blocks = tiles < ncu ? tiles : ncu;
  const size_t lds = (size_t)(2 * H4_V_BUF + 2 * H4_RAW_BUF + 8 * H4_OSTEP + (in_scale ? 2 * Cin : 0)) * 4;
  hipStream_t s = (hipStream_t)stream;
#define H4_LAUNCH(EPI, AFF)                                                   \
  {                                                                           \
    int rc = cy_allow_lds(wino4s2_conv_kernel<EPI, AFF>, lds);                \
    if (rc) return rc;                                                        \
    wino4s2_conv_kernel<EPI, AFF><<<(unsigned)blocks, 256, lds, s>>>(a);      \
  }
  if (in_scale != nullptr) { if (stats != nullptr) H4_LAUNCH(1, true) else H4_LAUNCH(0, true) }
  else if (stats != nullptr) H4_LAUNCH(1, false)
  else if (out_slope != 1.f) H4_LAUNCH(2, false)
  else H4_LAUNCH(0, false)
#undef H4_LAUNCH
  CY_LAUNCH_CHECK("cy_conv4x4s2_winograd4");
  return 0;
}

extern "C" int cy_wino4s2_dgrad_ok(int B, int H, int W, int Cin, int Cout) {     // H, W, Cin, Cout of the LAYER (dX is [B][H][W][Cin])
  return B > 0 && H > 0 && W > 0 && (H & 1) == 0 && (W & 1) == 0 && Cin % 64 == 0 && Cout % 8 == 0 && Cout >= 8 &&
         (long long)(H / 2) * (W / 2) * Cout * 4 + (long long)(W / 2 + 1) * Cout * 4 < (1ll << 28) && (long long)H * W * Cin < (1ll << 29);
}

extern "C" long long cy_wino4s2_dgrad_packed_floats(int Cin, int Cout) { return (long long)Cout * 26 * 4 * Cin; }

extern "C" int cy_wino4s2_pack_dgrad_weights(const float* W, float* U, int Cout, int Cin, void* stream) {
  CY_REQUIRE(W && U && Cout > 0 && Cin > 0 && Cout % 8 == 0 && Cin % 16 == 0, "cy_wino4s2_pack_dgrad_weights: bad arguments");
  const long long total = 4ll * Cin * Cout;
  wino4s2_pack_dgrad_kernel<<<(unsigned)cy_ceil_div(total, 256), 256, 0, (hipStream_t)stream>>>(W, U, Cout, Cin, total);
  CY_LAUNCH_CHECK("cy_wino4s2_pack_dgrad_weights");
  return 0;
}

extern "C" int cy_conv4x4s2_winograd4_dgrad(const float* dZ, const float* U, float* dX, const float* bn_z, const float* bn_scale,
                                            const float* bn_shift, const float* bn_mean, const float* bn_invstd, float bn_slope,
                                            double* bn_red, int B, int H, int W, int Cin, int Cout, void* stream) {
  CY_REQUIRE(dZ && U && dX, "cy_conv4x4s2_winograd4_dgrad: bad arguments");
  CY_REQUIRE(cy_wino4s2_dgrad_ok(B, H, W, Cin, Cout), "cy_conv4x4s2_winograd4_dgrad: shape B=%d H=%d W=%d Cin=%d Cout=%d not supported", B, H, W, Cin, Cout);
  CY_REQUIRE((((uintptr_t)dZ | (uintptr_t)U | (uintptr_t)dX | (uintptr_t)bn_z) & 15) == 0, "cy_conv4x4s2_winograd4_dgrad: operands must be 16-byte aligned");
  CY_REQUIRE(bn_red == nullptr || (bn_z && bn_scale && bn_shift && bn_mean && bn_invstd), "cy_conv4x4s2_winograd4_dgrad: bn_red needs bn_z / scale / shift / mean / invstd");
  Wino4S2Args a{};
  a.X = dZ; a.U = U; a.Y = dX; a.in_slope = 1.f; a.out_slope = 1.f;
  a.B = B; a.H = H / 2; a.W = W / 2; a.Cin = Cout;          // the kernel's "input" is dY [B][H/2][W/2][Cout]
  a.Cout = 4 * Cin; a.Np = 4 * Cin;
  a.Ho = H / 2 + 1; a.Wo = W / 2 + 1;                       // grid of the space-to-depth view
  a.tbh = (a.Ho + 15) / 16; a.tbw = (a.Wo + 31) / 32;
  a.Hx = H; a.Wx = W; a.Cx = Cin;
  a.bn_z = bn_z; a.bn_scale = bn_scale; a.bn_shift = bn_shift; a.bn_mean = bn_mean; a.bn_invstd = bn_invstd;
  a.bn_red = bn_red; a.bn_slope = bn_slope;
  const long long tiles = (long long)B * a.tbh * a.tbw * (a.Np / 64);
  CY_REQUIRE(tiles < (1ll << 31), "cy_conv4x4s2_winograd4_dgrad: too many tiles");
  a.ntiles = (int)tiles;
  const long long sp_tiles = (long long)B * a.tbh * a.tbw;
  a.sgroup = sp_tiles % 8 == 0 ? 8 : sp_tiles % 4 == 0 ? 4 : sp_tiles % 2 == 0 ? 2 : 1;
  int dev = 0, ncu = 0;
  hipError_t he = hipGetDevice(&dev);
  if (he == hipSuccess) he = hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev);
  if (he != hipSuccess || ncu <= 0) return cy_set_error((int)he, "cy_conv4x4s2_winograd4_dgrad: cannot query the CU count: %s", hipGetErrorString(he));
  long long blocks = tiles < ncu ? tiles : ncu;
  // a grid that is a multiple of the number of channel blocks keeps every block on ONE channel block (its BatchNorm sums leave once)
  const int nblk = a.Np / 64;
  if (blocks > nblk) blocks -= blocks % nblk;
  const size_t lds = (size_t)(2 * H4_V_BUF + 2 * H4_RAW_BUF + 8 * H4_OSTEP) * 4;
  int rc = cy_allow_lds(wino4s2_conv_kernel<0, false, 1>, lds);
  if (!rc) rc = cy_allow_lds(wino4s2_conv_kernel<0, false, 2>, lds);
  if (rc) return rc;
  if (bn_red != nullptr) wino4s2_conv_kernel<0, false, 2><<<(unsigned)blocks, 256, lds, (hipStream_t)stream>>>(a);
  else wino4s2_conv_kernel<0, false, 1><<<(unsigned)blocks, 256, lds, (hipStream_t)stream>>>(a);
  CY_LAUNCH_CHECK("cy_conv4x4s2_winograd4_dgrad");
  return 0;
}
