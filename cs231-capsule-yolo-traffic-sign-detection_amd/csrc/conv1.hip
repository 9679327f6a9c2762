// First layer of the backbones (models.py:347-349 DarkCapsuleNet conv_1: 3 -> 128, models.py:132-136 DarkNet conv_1:
// 3 -> 32; 3x3, stride 1, pad 1, NCHW image in): 1 % of the model's FLOPs but 2.8 GB of output at the headline shape,
// i.e. a STORE-bound layer.  The implicit-GEMM kernel spends its time in block prologues (43 k blocks with one K tile
// each, two per CU); here every wave is persistent and needs neither LDS nor barriers:
//   * a wave owns tiles of 32 consecutive pixels of one image row x all output channels (NT = Cout / 32 MFMA tiles),
//   * K = 27 taps -> 14 steps of mfma_f32_32x32x2f32; the B operand (weights, 14 NT registers) stays in registers for
//     the whole launch, the A operand is read straight from the image (22 MB, L2-resident): lane (pixel i, k half) needs
//     x[c][y + kh - 1][x0 + i + kw - 1], 128 contiguous bytes per (tap, half) -- fetched one tile ahead,
//   * MFMA tile nt holds the channels NT * column + nt: in the accumulator layout (lane = column, 16 pixel rows) a lane
//     owns NT consecutive channels of a pixel -- one 16-byte store per pixel and lane, 512 contiguous bytes per pixel,
//     full cache lines, no transposition,
//   * BatchNorm statistics are kept per lane over ALL tiles of the wave: one pair of double atomics per channel and wave.
#include "common.h"
// The activation (2.8 GB at the headline shape) is stored nontemporal -- it would only push the 22 MB image, which every
// pixel's 27 taps re-read, out of L2: conv1_fwd_act 0.65 -> 0.49 ms (5.7 TB/s); the gradient loads of the backward
// passes likewise (no measurable change there).
#ifndef CY_C1_NT
#define CY_C1_NT 1
#endif
#if CY_C1_NT
#define CY_C1_ST(v, p) __builtin_nontemporal_store((v), (p))
#define CY_C1_LD(p) __builtin_nontemporal_load((p))
#else
#define CY_C1_ST(v, p) (*(p) = (v))
#define CY_C1_LD(p) (*(p))
#endif


namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

struct Conv1Args {
  const float* X; const float* W; const float* bias; float* Y; double* stats;
  const float* scale; const float* shift; float slope;   // forward only, optional: Y = lrelu((conv + bias) * scale + shift)
  int B, H, Wd, Cout;
  long long ntiles;                         // B * H * Wd / 32
};

// OBF: the activation is written as bf16 (the bf16 path's second block reads it as such: no fp32 copy, no cast pass)
template <int NT, bool AFF, bool STORE, bool OBF = false>
__global__ __launch_bounds__(256, 1) void conv1_fwd_kernel(Conv1Args a) {
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int li = lane & 31, lh = lane >> 5;
  const long long gw = (long long)blockIdx.x * 4 + wave, nw = (long long)gridDim.x * 4;
  const int segs = a.Wd / 32;
  const size_t plane = (size_t)a.H * a.Wd;

  // ---- per-lane tap geometry: step s uses k = 2 s + lh -> (c, kh, kw); k = 27 (s = 13, upper half) is padding
  int toff[14];                             // offset of the tap from the pixel, in floats, inside the image
  int tky[14], tkx[14];
  float wreg[NT][14];
#pragma unroll
  for (int s = 0; s < 14; ++s) {
    const int k = 2 * s + lh;
    const int kk = k < 27 ? k : 0;
    const int c = kk / 9, kh = (kk % 9) / 3, kw = kk % 3;
    tky[s] = k < 27 ? kh - 1 : (1 << 20);   // padding step: never inside the image -> operand 0
    tkx[s] = kw - 1;
    toff[s] = (int)(c * plane) + (kh - 1) * a.Wd + (kw - 1);
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
      wreg[nt][s] = k < 27 ? a.W[((size_t)(NT * li + nt) * 3 + c) * 9 + kh * 3 + kw] : 0.f;
  }
  float bv[NT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) bv[nt] = a.bias != nullptr ? a.bias[NT * li + nt] : 0.f;
  float ssum[NT], ssq[NT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) { ssum[nt] = 0.f; ssq[nt] = 0.f; }
  // optional BatchNorm affine + LeakyReLU on the way out (second pass of a conv -> BN -> LeakyReLU block: the layer is
  // store-bound and its input tiny, so recomputing the convolution is cheaper than reading z back): folded into bias
  constexpr bool aff = AFF;                 // compile time: a uniform branch per element would cost more than the layer
  float asc[NT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) {
    asc[nt] = aff ? a.scale[NT * li + nt] : 1.f;
    if (aff) bv[nt] = bv[nt] * asc[nt] + a.shift[NT * li + nt];
  }

  auto load_a = [&](long long tile, float (&av)[14]) {
    const int seg = (int)(tile % segs);
    const long long row = tile / segs;
    const int y = (int)(row % a.H), b = (int)(row / a.H);
    const int x = seg * 32 + li;
    const float* px = a.X + (size_t)b * 3 * plane + (size_t)y * a.Wd + x;
    const bool inner = y >= 1 && y + 1 < a.H && seg >= 1 && seg + 1 < segs;    // uniform
    if (inner) {
#pragma unroll
      for (int s = 0; s < 14; ++s) av[s] = (s < 13 || lh == 0) ? px[toff[s]] : 0.f;
    } else {
#pragma unroll
      for (int s = 0; s < 14; ++s) {
        const bool ok = (unsigned)(y + tky[s]) < (unsigned)a.H && (unsigned)(x + tkx[s]) < (unsigned)a.Wd;
        const float v = px[ok ? toff[s] : 0];
        av[s] = ok ? v : 0.f;
      }
    }
  };

  float acur[14], anext[14];
  long long tile = gw;
  if (tile < a.ntiles) load_a(tile, acur);
  for (; tile < a.ntiles; tile += nw) {
    const long long tn = tile + nw;
    if (tn < a.ntiles) load_a(tn, anext);   // one tile of latency cover
    f32x16 acc[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[nt][r] = 0.f;
#pragma unroll
    for (int s = 0; s < 14; ++s)
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) acc[nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(acur[s], wreg[nt][s], acc[nt], 0, 0, 0);
    // ---- epilogue: + bias, statistics; MFMA tile nt holds the channels NT * column + nt, so a lane owns NT
    // consecutive channels of each of its 16 pixels: one 4 NT-byte store per pixel, a pixel's Cout channels are contiguous
    float* yp = a.Y + (size_t)tile * 32 * a.Cout + NT * li;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      typedef float vecn __attribute__((ext_vector_type(NT)));
      const int p = (r & 3) + 8 * (r >> 2) + 4 * lh;
      float v[NT];
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        if (aff) {
          const float y = __builtin_fmaf(acc[nt][r], asc[nt], bv[nt]);
          const float ys = y * a.slope;                         // slope in [0, 1]: lrelu(y) = max(y, slope y)
          asm("v_max_f32 %0, %1, %2" : "=v"(v[nt]) : "v"(y), "v"(ys));   // fmaxf() would canonicalise both operands first
        } else {
          v[nt] = acc[nt][r] + bv[nt];
          ssum[nt] += v[nt];
          ssq[nt] = __builtin_fmaf(v[nt], v[nt], ssq[nt]);
        }
      }
      if constexpr (!STORE) {
        (void)p;                            // statistics only: z is recomputed wherever it is needed
      } else if constexpr (OBF) {
        unsigned short* yb = (unsigned short*)a.Y + (size_t)tile * 32 * a.Cout + NT * li + (size_t)p * a.Cout;
        unsigned short h[NT];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) { const __bf16 b = (__bf16)v[nt]; h[nt] = *(const unsigned short*)&b; }
        if constexpr (NT == 4) {
          typedef unsigned u32x2_t __attribute__((ext_vector_type(2)));
          CY_C1_ST((u32x2_t{(unsigned)h[0] | ((unsigned)h[1] << 16), (unsigned)h[2] | ((unsigned)h[3] << 16)}), (u32x2_t*)yb);
        } else if constexpr (NT == 2) {
          *(unsigned*)yb = (unsigned)h[0] | ((unsigned)h[1] << 16);
        } else {
          yb[0] = h[0];
        }
      } else if constexpr (NT == 1) {
        yp[(size_t)p * a.Cout] = v[0];
      } else {
        vecn o;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) o[nt] = v[nt];
        CY_C1_ST(o, (vecn*)(yp + (size_t)p * a.Cout));
      }
    }
#pragma unroll
    for (int s = 0; s < 14; ++s) acur[s] = anext[s];
  }
  if (a.stats != nullptr) {
    double* st = a.stats + (size_t)((blockIdx.x * 4 + wave) % CY_STATS_COPIES) * a.Cout * 2;
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      const float s1 = ssum[nt] + __shfl_xor(ssum[nt], 32, 64), s2 = ssq[nt] + __shfl_xor(ssq[nt], 32, 64);
      if (lh == 0) {
        atomicAdd(st + 2 * (NT * li + nt), (double)s1);
        atomicAdd(st + 2 * (NT * li + nt) + 1, (double)s2);
      }
    }
  }
}

// ---- the activation pass of the bf16 path ("precision": "bf16") on the bf16 matrix cores: K = 27 taps + 5 zeros = two steps of
// v_mfma_f32_32x32x16_bf16 per channel tile instead of 14 of v_mfma_f32_32x32x2f32 -- 8 MFMAs of 32 cycles per 32-pixel tile against
// 56 of 64 cycles: the fp32 form is MFMA-bound (1.1 ms at 608 x 608, batch 32, for 3 GB of bf16 output), this one store-bound.
// Lane (pixel li, k half lh) holds the taps k = 16 s + 8 lh + e, e = 0..7, of step s: image values and weights are rounded to bf16
// (8 significant bits: the u8 pixels behind the centred image lose nothing but the centring's fraction) and accumulate in fp32; the
// BatchNorm constants come from the block's fp32 statistics as before.
typedef __bf16 c1_bf16x8 __attribute__((ext_vector_type(8)));
template <int NT>
__global__ __launch_bounds__(256, 1) void conv1_fwd_bf16_kernel(Conv1Args a) {
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int li = lane & 31, lh = lane >> 5;
  const long long gw = (long long)blockIdx.x * 4 + wave, nw = (long long)gridDim.x * 4;
  const int segs = a.Wd / 32;
  const size_t plane = (size_t)a.H * a.Wd;
  int toff[16], tky[16], tkx[16];
  c1_bf16x8 wreg[NT][2];
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const int k = 16 * (i >> 3) + 8 * lh + (i & 7);
    const int kk = k < 27 ? k : 0;
    const int c = kk / 9, kh = (kk % 9) / 3, kw = kk % 3;
    tky[i] = k < 27 ? kh - 1 : (1 << 20);   // padding tap: never inside the image -> operand 0
    tkx[i] = kw - 1;
    toff[i] = (int)(c * plane) + (kh - 1) * a.Wd + (kw - 1);
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
      wreg[nt][i >> 3][i & 7] = (__bf16)(k < 27 ? a.W[((size_t)(NT * li + nt) * 3 + c) * 9 + kh * 3 + kw] : 0.f);
  }
  float bv[NT], asc[NT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) {
    asc[nt] = a.scale[NT * li + nt];
    bv[nt] = (a.bias != nullptr ? a.bias[NT * li + nt] : 0.f) * asc[nt] + a.shift[NT * li + nt];
  }
  auto load_a = [&](long long tile, float (&av)[16]) {
    const int seg = (int)(tile % segs);
    const long long row = tile / segs;
    const int y = (int)(row % a.H), b = (int)(row / a.H);
    const int x = seg * 32 + li;
    const float* px = a.X + (size_t)b * 3 * plane + (size_t)y * a.Wd + x;
    const bool inner = y >= 1 && y + 1 < a.H && seg >= 1 && seg + 1 < segs;    // uniform
    if (inner) {
#pragma unroll
      for (int i = 0; i < 16; ++i) av[i] = (i < 8 || lh == 0 || (i & 7) < 3) ? px[toff[i]] : 0.f;
    } else {
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const bool ok = (unsigned)(y + tky[i]) < (unsigned)a.H && (unsigned)(x + tkx[i]) < (unsigned)a.Wd;
        const float v = px[ok ? toff[i] : 0];
        av[i] = ok ? v : 0.f;
      }
    }
  };
  float acur[16], anext[16];
  long long tile = gw;
  if (tile < a.ntiles) load_a(tile, acur);
  for (; tile < a.ntiles; tile += nw) {
    const long long tn = tile + nw;
    if (tn < a.ntiles) load_a(tn, anext);   // one tile of latency cover
    c1_bf16x8 a8[2];
#pragma unroll
    for (int i = 0; i < 16; ++i) a8[i >> 3][i & 7] = (__bf16)acur[i];
    f32x16 acc[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[nt][r] = 0.f;
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) acc[nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a8[s], wreg[nt][s], acc[nt], 0, 0, 0);
    // epilogue: as conv1_fwd_kernel<NT, true, true, true> (a lane owns NT consecutive channels of each of its 16 pixels)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int p = (r & 3) + 8 * (r >> 2) + 4 * lh;
      unsigned short h[NT];
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        const float y = __builtin_fmaf(acc[nt][r], asc[nt], bv[nt]);
        const float ys = y * a.slope;
        float v;
        asm("v_max_f32 %0, %1, %2" : "=v"(v) : "v"(y), "v"(ys));
        const __bf16 b = (__bf16)v;
        h[nt] = *(const unsigned short*)&b;
      }
      unsigned short* yb = (unsigned short*)a.Y + (size_t)tile * 32 * a.Cout + NT * li + (size_t)p * a.Cout;
      if constexpr (NT == 4) {
        typedef unsigned u32x2_t __attribute__((ext_vector_type(2)));
        CY_C1_ST((u32x2_t{(unsigned)h[0] | ((unsigned)h[1] << 16), (unsigned)h[2] | ((unsigned)h[3] << 16)}), (u32x2_t*)yb);
      } else if constexpr (NT == 2) {
        *(unsigned*)yb = (unsigned)h[0] | ((unsigned)h[1] << 16);
      } else {
        yb[0] = h[0];
      }
    }
#pragma unroll
    for (int i = 0; i < 16; ++i) acur[i] = anext[i];
  }
}

// ---- weight gradient of the same layer: dW[co][c][kh][kw] = sum over pixels of dZ[pixel][co] * x[c][y + kh - 1][x + kw - 1].
// The pixels are the MFMA k dimension: D[co (32 NT rows)][tap (27 of 32 columns)] += A[co][pixel] B[pixel][tap] with
// A = dZ (lane = channel, k half = pixel parity: 128 contiguous bytes per pixel) and B = the image patch (lane = tap:
// every lane walks ITS OWN row of the image, consecutive pixels = consecutive addresses).  Columns 27..31 accumulate
// finite garbage and are never stored.  Every wave is persistent over tiles of 32 pixels (16 MFMA steps x NT), loads
// one tile ahead, and leaves its partial sums in a slab; `conv1_wgrad_finish_kernel` adds the slabs in a fixed order.
template <int NT>
__global__ __launch_bounds__(256, 1) void conv1_wgrad_kernel(Conv1Args a, float* __restrict__ slabs) {
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int li = lane & 31, lh = lane >> 5;
  const long long gw = (long long)blockIdx.x * 4 + wave, nw = (long long)gridDim.x * 4;
  const int segs = a.Wd / 32;
  const size_t plane = (size_t)a.H * a.Wd;
  const int kk = li < 27 ? li : 0;          // this lane's tap (columns 27..31 read tap 0's data: never stored)
  const int tc = kk / 9, tkh = (kk % 9) / 3, tkw = kk % 3;
  const int toff = (int)(tc * plane) + (tkh - 1) * a.Wd + (tkw - 1) + lh;    // + 2 s: pixel 2 s + lh of the tile

  auto load_tile = [&](long long tile, float (&av)[NT][16], float (&bv)[16]) {
    const int seg = (int)(tile % segs);
    const long long row = tile / segs;
    const int y = (int)(row % a.H), b = (int)(row / a.H);
    // MFMA tile nt holds the channels NT * row + nt: a lane's NT operands of one pixel are NT consecutive floats
    const float* pz = a.Y + (size_t)tile * 32 * a.Cout + (size_t)lh * a.Cout + NT * li;  // a.Y = dZ here
#pragma unroll
    for (int s = 0; s < 16; ++s) {
      typedef float vecn __attribute__((ext_vector_type(NT)));
      if constexpr (NT == 1) {
        av[0][s] = pz[(size_t)(2 * s) * a.Cout];
      } else {
        const vecn v = *(const vecn*)(pz + (size_t)(2 * s) * a.Cout);
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) av[nt][s] = v[nt];
      }
    }
    const float* px = a.X + (size_t)b * 3 * plane + (size_t)y * a.Wd + seg * 32;
    const bool inner = y >= 1 && y + 1 < a.H && seg >= 1 && seg + 1 < segs;              // uniform
    if (inner) {
#pragma unroll
      for (int s = 0; s < 16; ++s) bv[s] = px[toff + 2 * s];
    } else {
      const bool rowok = (unsigned)(y + tkh - 1) < (unsigned)a.H;
#pragma unroll
      for (int s = 0; s < 16; ++s) {
        const bool ok = rowok && (unsigned)(seg * 32 + 2 * s + lh + tkw - 1) < (unsigned)a.Wd;
        const float v = px[ok ? toff + 2 * s : 0];
        bv[s] = ok ? v : 0.f;
      }
    }
  };

  f32x16 acc[NT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[nt][r] = 0.f;
  float acur[NT][16], bcur[16], anext[NT][16], bnext[16];
  long long tile = gw;
  if (tile < a.ntiles) load_tile(tile, acur, bcur);
  for (; tile < a.ntiles; tile += nw) {
    const long long tn = tile + nw;
    if (tn < a.ntiles) load_tile(tn, anext, bnext);
#pragma unroll
    for (int s = 0; s < 16; ++s)
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) acc[nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(acur[nt][s], bcur[s], acc[nt], 0, 0, 0);
#pragma unroll
    for (int s = 0; s < 16; ++s) {
      bcur[s] = bnext[s];
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) acur[nt][s] = anext[nt][s];
    }
  }
  // slab[wave][co][32 taps]: lane li = tap (contiguous), accumulator row = channel
  float* out = slabs + (size_t)gw * a.Cout * 32;
#pragma unroll
  for (int nt = 0; nt < NT; ++nt)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int co = NT * ((r & 3) + 8 * (r >> 2) + 4 * lh) + nt;
      out[(size_t)co * 32 + li] = acc[nt][r];
    }
}

// ---- backward of the whole first block (conv -> BatchNorm -> LeakyReLU) without z and without dz in memory.
// The block's z (2.8 GB) would be read twice (BatchNorm-backward sums, then the apply pass) and dz written and read once
// more by the weight gradient; instead both passes RECOMPUTE z = conv(x) + bias per 32-pixel tile (56 MFMAs, the image
// is L2-resident) and read only dA, the gradient with respect to the activation:
//   PASS 0: red[c] += (sum d, sum d * xhat), d = dA * lrelu'(z * scale + shift), xhat = (z - mean) * invstd
//   PASS 1: dz = scale * (d - m1 - xhat * m2) (m = red / count) is formed in registers IN THE LAYOUT the weight-gradient
//           MFMA wants: the accumulator of the recomputed tile has lane = channel column, 16 pixel rows p(r, half), and the
//           weight gradient's k dimension (pixels) may be enumerated in any order, so step s takes pixel p(s, half) --
//           the lane's own accumulator row s -- and the image operand is fetched for the same pixel.
struct Conv1BnArgs {
  const float* X; const float* W; const float* bias; const float* dA;
  const float* scale; const float* shift; const float* mean; const float* invstd; float slope;
  double* red_out;                          // PASS 0: [CY_STATS_COPIES][Cout][2], zeroed by the caller
  const double* red_in; double inv_count;   // PASS 1: [Cout][2] summed (and, for SyncBN, averaged) by the caller
  float* slabs;                             // PASS 1: per-wave partial dW
  int B, H, Wd, Cout;
  long long ntiles;
};

// GBF: dA is bf16 (it comes from the bf16 path's second block)
template <int NT, int PASS, bool GBF = false>
__global__ __launch_bounds__(256, 1) void conv1_bn_bwd_kernel(Conv1BnArgs a) {
  typedef float vecn __attribute__((ext_vector_type(NT)));
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int li = lane & 31, lh = lane >> 5;
  const long long gw = (long long)blockIdx.x * 4 + wave, nw = (long long)gridDim.x * 4;
  const int segs = a.Wd / 32;
  const size_t plane = (size_t)a.H * a.Wd;
  // forward operand geometry (conv1_fwd_kernel): lane = pixel li, step s = taps 2 s + lh
  int toff[14], tky[14], tkx[14];
  float wreg[NT][14];
#pragma unroll
  for (int s = 0; s < 14; ++s) {
    const int k = 2 * s + lh;
    const int kk = k < 27 ? k : 0;
    const int c = kk / 9, kh = (kk % 9) / 3, kw = kk % 3;
    tky[s] = k < 27 ? kh - 1 : (1 << 20);
    tkx[s] = kw - 1;
    toff[s] = (int)(c * plane) + (kh - 1) * a.Wd + (kw - 1);
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
      wreg[nt][s] = k < 27 ? a.W[((size_t)(NT * li + nt) * 3 + c) * 9 + kh * 3 + kw] : 0.f;
  }
  // weight-gradient operand geometry (conv1_wgrad_kernel): lane = tap li, step s = pixel p(s, lh)
  const int kt = li < 27 ? li : 0;
  const int wc = kt / 9, wkh = (kt % 9) / 3, wkw = kt % 3;
  const int woff = (int)(wc * plane) + (wkh - 1) * a.Wd + (wkw - 1) + 4 * lh;    // + (s & 3) + 8 (s >> 2)
  // per-channel constants of this lane's NT channels
  float bv[NT], sc[NT], sh[NT], is[NT], nm[NT], m1[NT], m2[NT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) {
    const int c = NT * li + nt;
    bv[nt] = a.bias != nullptr ? a.bias[c] : 0.f;
    sc[nt] = a.scale[c]; sh[nt] = a.shift[c]; is[nt] = a.invstd[c]; nm[nt] = -a.mean[c] * is[nt];
    m1[nt] = PASS == 1 ? (float)(a.red_in[2 * c] * a.inv_count) : 0.f;
    m2[nt] = PASS == 1 ? (float)(a.red_in[2 * c + 1] * a.inv_count) : 0.f;
  }
  float b1[NT], b2[NT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) { b1[nt] = 0.f; b2[nt] = 0.f; }

  auto load_img = [&](long long tile, float (&av)[14], float (&bw)[16]) {
    const int seg = (int)(tile % segs);
    const long long row = tile / segs;
    const int y = (int)(row % a.H), b = (int)(row / a.H);
    const float* p0 = a.X + (size_t)b * 3 * plane + (size_t)y * a.Wd + seg * 32;
    const float* px = p0 + li;
    const bool inner = y >= 1 && y + 1 < a.H && seg >= 1 && seg + 1 < segs;    // uniform
    if (inner) {
#pragma unroll
      for (int s = 0; s < 14; ++s) av[s] = (s < 13 || lh == 0) ? px[toff[s]] : 0.f;
      if (PASS >= 1) {
#pragma unroll
        for (int s = 0; s < 16; ++s) bw[s] = p0[woff + (s & 3) + 8 * (s >> 2)];
      }
    } else {
#pragma unroll
      for (int s = 0; s < 14; ++s) {
        const bool ok = (unsigned)(y + tky[s]) < (unsigned)a.H && (unsigned)(seg * 32 + li + tkx[s]) < (unsigned)a.Wd;
        const float v = px[ok ? toff[s] : 0];
        av[s] = ok ? v : 0.f;
      }
      if (PASS >= 1) {
        const bool rowok = (unsigned)(y + wkh - 1) < (unsigned)a.H;
#pragma unroll
        for (int s = 0; s < 16; ++s) {
          const int pcol = (s & 3) + 8 * (s >> 2) + 4 * lh;
          const bool ok = rowok && (unsigned)(seg * 32 + pcol + wkw - 1) < (unsigned)a.Wd;
          const float v = p0[ok ? woff + (s & 3) + 8 * (s >> 2) : 0];
          bw[s] = ok ? v : 0.f;
        }
      }
    }
  };

  f32x16 accw[NT];                          // PASS 1: dW[channel NT row + nt][tap]
#pragma unroll
  for (int nt = 0; nt < NT; ++nt)
#pragma unroll
    for (int r = 0; r < 16; ++r) accw[nt][r] = 0.f;
  float acur[14], anext[14], bwcur[16], bwnext[16];
  long long tile = gw;
  if (tile < a.ntiles) load_img(tile, acur, bwcur);
  for (; tile < a.ntiles; tile += nw) {
    // dA of this tile in the accumulator layout: requested first, used after the 14 NT recompute MFMAs
    vecn g[16];
    if constexpr (GBF) {
      const unsigned short* pg = (const unsigned short*)a.dA + (size_t)tile * 32 * a.Cout + NT * li;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const unsigned short* q = pg + (size_t)((r & 3) + 8 * (r >> 2) + 4 * lh) * a.Cout;
        if constexpr (NT == 4) {
          typedef unsigned u32x2_t __attribute__((ext_vector_type(2)));
          const u32x2_t w = CY_C1_LD((const u32x2_t*)q);
          g[r][0] = __uint_as_float(w[0] << 16); g[r][1] = __uint_as_float(w[0] & 0xffff0000u);
          g[r][2] = __uint_as_float(w[1] << 16); g[r][3] = __uint_as_float(w[1] & 0xffff0000u);
        } else if constexpr (NT == 2) {
          const unsigned w = *(const unsigned*)q;
          g[r][0] = __uint_as_float(w << 16); g[r][1] = __uint_as_float(w & 0xffff0000u);
        } else {
          g[r] = vecn(__uint_as_float((unsigned)q[0] << 16));
        }
      }
    } else {
      const float* pg = a.dA + (size_t)tile * 32 * a.Cout + NT * li;
#pragma unroll
      for (int r = 0; r < 16; ++r) g[r] = CY_C1_LD((const vecn*)(pg + (size_t)((r & 3) + 8 * (r >> 2) + 4 * lh) * a.Cout));
    }
    const long long tn = tile + nw;
    if (tn < a.ntiles) load_img(tn, anext, bwnext);
    f32x16 acc[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[nt][r] = 0.f;
#pragma unroll
    for (int s = 0; s < 14; ++s)
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) acc[nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(acur[s], wreg[nt][s], acc[nt], 0, 0, 0);
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      float dzr[NT];
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        const float z = acc[nt][r] + bv[nt];
        const float y = __builtin_fmaf(z, sc[nt], sh[nt]);
        const float gv = ((const float*)&g[r])[nt];
        const float d = y > 0.f ? gv : gv * a.slope;
        const float xh = __builtin_fmaf(z, is[nt], nm[nt]);
        if (PASS == 0) {
          b1[nt] += d;
          b2[nt] = __builtin_fmaf(d, xh, b2[nt]);
        } else if (PASS == 1) {
          dzr[nt] = sc[nt] * (d - m1[nt] - xh * m2[nt]);
        } else {                            // PASS 2: sum d and the weight gradient OF d; conv1_bn_bwd_finish_kernel does the rest
          b1[nt] += d;
          dzr[nt] = d;
        }
      }
      if (PASS >= 1) {
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) accw[nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(dzr[nt], bwcur[r], accw[nt], 0, 0, 0);
      }
    }
#pragma unroll
    for (int s = 0; s < 14; ++s) acur[s] = anext[s];
    if (PASS >= 1) {
#pragma unroll
      for (int s = 0; s < 16; ++s) bwcur[s] = bwnext[s];
    }
  }
  if (PASS == 0 || PASS == 2) {
    double* rd = a.red_out + (size_t)((blockIdx.x * 4 + wave) % CY_STATS_COPIES) * a.Cout * 2;
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      const float s1 = b1[nt] + __shfl_xor(b1[nt], 32, 64), s2 = b2[nt] + __shfl_xor(b2[nt], 32, 64);
      if (lh == 0) {
        atomicAdd(rd + 2 * (NT * li + nt), (double)s1);
        if (PASS == 0) atomicAdd(rd + 2 * (NT * li + nt) + 1, (double)s2);
      }
    }
  }
  if (PASS >= 1) {
    float* out = a.slabs + (size_t)gw * a.Cout * 32;
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int co = NT * ((r & 3) + 8 * (r >> 2) + 4 * lh) + nt;
        out[(size_t)co * 32 + li] = accw[nt][r];
      }
  }
}

// ---- PASS 2 of the bf16 path on the bf16 matrix cores (cy_conv1_bn_bwd_onepass_bf16): the recomputation of z is the forward's own
// two v_mfma_f32_32x32x16_bf16 steps (conv1_fwd_bf16_kernel: the same operands in the same order, so the mask lrelu'(y) is the one
// the stored activation was formed with), and the weight gradient of d takes the pixels as its k dimension in the order that needs no
// data movement at all: step s of channel tile nt multiplies the lane's OWN accumulator rows r = 8 s .. 8 s + 7 (pixels p(r, lh),
// its 8 k values) with the patch values of the same pixels that the lane of tap li has fetched (bw[8 s .. 8 s + 7], as in the fp32
// kernel).  16 bf16 MFMAs of 32 cycles per 32-pixel tile against 120 fp32 MFMAs of 64 (2.37 ms at 608 x 608, batch 32: the launch
// was bound by them); what is left is the read of dA (3 GB) and ~500 vector instructions per tile.
// (PASS as in conv1_bn_bwd_kernel: 0 the sums, 1 the weight gradient of dz, 2 the one-pass form)
template <int NT, int PASS>
__global__ __launch_bounds__(256, 1) void conv1_bn_bwd_bf16mm_kernel(Conv1BnArgs a) {
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int li = lane & 31, lh = lane >> 5;
  const long long gw = (long long)blockIdx.x * 4 + wave, nw = (long long)gridDim.x * 4;
  const int segs = a.Wd / 32;
  const size_t plane = (size_t)a.H * a.Wd;
  // forward operand geometry (conv1_fwd_bf16_kernel): lane = pixel li, taps k = 16 s + 8 lh + e
  int toff[16], tky[16], tkx[16];
  c1_bf16x8 wreg[NT][2];
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const int k = 16 * (i >> 3) + 8 * lh + (i & 7);
    const int kk = k < 27 ? k : 0;
    const int c = kk / 9, kh = (kk % 9) / 3, kw = kk % 3;
    tky[i] = k < 27 ? kh - 1 : (1 << 20);
    tkx[i] = kw - 1;
    toff[i] = (int)(c * plane) + (kh - 1) * a.Wd + (kw - 1);
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
      wreg[nt][i >> 3][i & 7] = (__bf16)(k < 27 ? a.W[((size_t)(NT * li + nt) * 3 + c) * 9 + kh * 3 + kw] : 0.f);
  }
  // weight-gradient operand geometry (conv1_wgrad_kernel): lane = tap li, k value (s, e) = pixel p(8 s + e, lh)
  const int kt = li < 27 ? li : 0;
  const int wc = kt / 9, wkh = (kt % 9) / 3, wkw = kt % 3;
  const int woff = (int)(wc * plane) + (wkh - 1) * a.Wd + (wkw - 1) + 4 * lh;    // + (r & 3) + 8 (r >> 2)
  float bv[NT], sc[NT], sh[NT], is[NT], nm[NT], m1[NT], m2[NT], b1[NT], b2[NT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) {
    const int c = NT * li + nt;
    bv[nt] = a.bias != nullptr ? a.bias[c] : 0.f;
    sc[nt] = a.scale[c]; sh[nt] = a.shift[c];
    sh[nt] = __builtin_fmaf(bv[nt], sc[nt], sh[nt]);          // y = acc * scale + (bias * scale + shift)
    is[nt] = PASS < 2 ? a.invstd[c] : 0.f;                    // xhat = acc * invstd + (bias - mean) * invstd
    nm[nt] = PASS < 2 ? (bv[nt] - a.mean[c]) * is[nt] : 0.f;
    m1[nt] = PASS == 1 ? (float)(a.red_in[2 * c] * a.inv_count) : 0.f;
    m2[nt] = PASS == 1 ? (float)(a.red_in[2 * c + 1] * a.inv_count) : 0.f;
    b1[nt] = 0.f; b2[nt] = 0.f;
  }
  auto load_img = [&](long long tile, float (&av)[16], float (&bw)[16]) {
    const int seg = (int)(tile % segs);
    const long long row = tile / segs;
    const int y = (int)(row % a.H), b = (int)(row / a.H);
    const float* p0 = a.X + (size_t)b * 3 * plane + (size_t)y * a.Wd + seg * 32;
    const float* px = p0 + li;
    const bool inner = y >= 1 && y + 1 < a.H && seg >= 1 && seg + 1 < segs;    // uniform
    if (inner) {
#pragma unroll
      for (int i = 0; i < 16; ++i) av[i] = (i < 8 || lh == 0 || (i & 7) < 3) ? px[toff[i]] : 0.f;
      if (PASS >= 1) {
#pragma unroll
        for (int r = 0; r < 16; ++r) bw[r] = p0[woff + (r & 3) + 8 * (r >> 2)];
      }
    } else {
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const bool ok = (unsigned)(y + tky[i]) < (unsigned)a.H && (unsigned)(seg * 32 + li + tkx[i]) < (unsigned)a.Wd;
        const float v = px[ok ? toff[i] : 0];
        av[i] = ok ? v : 0.f;
      }
      if (PASS >= 1) {
        const bool rowok = (unsigned)(y + wkh - 1) < (unsigned)a.H;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int pcol = (r & 3) + 8 * (r >> 2) + 4 * lh;
          const bool ok = rowok && (unsigned)(seg * 32 + pcol + wkw - 1) < (unsigned)a.Wd;
          const float v = p0[ok ? woff + (r & 3) + 8 * (r >> 2) : 0];
          bw[r] = ok ? v : 0.f;
        }
      }
    }
  };
  f32x16 accw[NT];                          // PASS >= 1: dW of dz (1) / of d (2): [channel NT row + nt][tap]
#pragma unroll
  for (int nt = 0; nt < NT; ++nt)
#pragma unroll
    for (int r = 0; r < 16; ++r) accw[nt][r] = 0.f;
  float acur[16], anext[16], bwcur[16], bwnext[16];
  long long tile = gw;
  if (tile < a.ntiles) load_img(tile, acur, bwcur);
  for (; tile < a.ntiles; tile += nw) {
    // dA (bf16) of this tile in the accumulator layout: requested first, used after the recompute MFMAs
    float g[16][NT];
    {
      const unsigned short* pg = (const unsigned short*)a.dA + (size_t)tile * 32 * a.Cout + NT * li;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const unsigned short* q = pg + (size_t)((r & 3) + 8 * (r >> 2) + 4 * lh) * a.Cout;
        if constexpr (NT == 4) {
          typedef unsigned u32x2_t __attribute__((ext_vector_type(2)));
          const u32x2_t w = CY_C1_LD((const u32x2_t*)q);
          g[r][0] = __uint_as_float(w[0] << 16); g[r][1] = __uint_as_float(w[0] & 0xffff0000u);
          g[r][2] = __uint_as_float(w[1] << 16); g[r][3] = __uint_as_float(w[1] & 0xffff0000u);
        } else if constexpr (NT == 2) {
          const unsigned w = *(const unsigned*)q;
          g[r][0] = __uint_as_float(w << 16); g[r][1] = __uint_as_float(w & 0xffff0000u);
        } else {
          g[r][0] = __uint_as_float((unsigned)q[0] << 16);
        }
      }
    }
    const long long tn = tile + nw;
    if (tn < a.ntiles) load_img(tn, anext, bwnext);
    c1_bf16x8 a8[2], b8[2];
#pragma unroll
    for (int i = 0; i < 16; ++i) { a8[i >> 3][i & 7] = (__bf16)acur[i]; if (PASS >= 1) b8[i >> 3][i & 7] = (__bf16)bwcur[i]; }
    f32x16 acc[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[nt][r] = 0.f;
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) acc[nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a8[s], wreg[nt][s], acc[nt], 0, 0, 0);
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      c1_bf16x8 d8[2];
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float y = __builtin_fmaf(acc[nt][r], sc[nt], sh[nt]);
        const float gv = g[r][nt];
        const float d = y > 0.f ? gv : gv * a.slope;
        if (PASS == 0) {
          const float xh = __builtin_fmaf(acc[nt][r], is[nt], nm[nt]);
          b1[nt] += d;
          b2[nt] = __builtin_fmaf(d, xh, b2[nt]);
        } else if (PASS == 1) {
          const float xh = __builtin_fmaf(acc[nt][r], is[nt], nm[nt]);
          d8[r >> 3][r & 7] = (__bf16)(sc[nt] * (d - m1[nt] - xh * m2[nt]));
        } else {
          b1[nt] += d;
          d8[r >> 3][r & 7] = (__bf16)d;
        }
      }
      if (PASS >= 1) {
#pragma unroll
        for (int s = 0; s < 2; ++s) accw[nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(d8[s], b8[s], accw[nt], 0, 0, 0);
      }
    }
#pragma unroll
    for (int i = 0; i < 16; ++i) { acur[i] = anext[i]; if (PASS >= 1) bwcur[i] = bwnext[i]; }
  }
  if (PASS == 0 || PASS == 2) {
    double* rd = a.red_out + (size_t)((blockIdx.x * 4 + wave) % CY_STATS_COPIES) * a.Cout * 2;
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      const float s1 = b1[nt] + __shfl_xor(b1[nt], 32, 64), s2 = b2[nt] + __shfl_xor(b2[nt], 32, 64);
      if (lh == 0) {
        atomicAdd(rd + 2 * (NT * li + nt), (double)s1);
        if (PASS == 0) atomicAdd(rd + 2 * (NT * li + nt) + 1, (double)s2);
      }
    }
  }
  if (PASS >= 1) {
    float* out = a.slabs + (size_t)gw * a.Cout * 32;
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int co = NT * ((r & 3) + 8 * (r >> 2) + 4 * lh) + nt;
        out[(size_t)co * 32 + li] = accw[nt][r];
      }
  }
}

// dW[co][c][kh][kw] (= [co][27]) = sum over the waves' slabs in a fixed order: one block per channel, thread = (tap,
// slab group): 128 contiguous bytes per slab row, 8 partial sums per tap combined through LDS
__global__ __launch_bounds__(256) void conv1_wgrad_finish_kernel(const float* __restrict__ slabs, float* __restrict__ dW,
                                                                 int nslab, int Cout) {
  __shared__ float part[8][32];
  const int co = blockIdx.x, tap = threadIdx.x & 31, grp = threadIdx.x >> 5;
  float s = 0.f;
  for (int w = grp; w < nslab; w += 8) s += slabs[((size_t)w * Cout + co) * 32 + tap];
  part[grp][tap] = s;
  __syncthreads();
  if (threadIdx.x < 27) {
    float v = 0.f;
#pragma unroll
    for (int g = 0; g < 8; ++g) v += part[g][threadIdx.x];
    dW[co * 27 + threadIdx.x] = v;
  }
}

// One-pass backward of the first block, second half.  PASS 2 left G[co][t] = sum_p d_p patch_p[t] (per-wave slabs) and
// Sd[co] = sum_p d_p (d = dA * lrelu'(y)); with the patch moments of the forward (csrc/conv1_moments.hip: M2 = sum patch patch^T,
// mv = sum patch = row 27, P = M2[27][27]) everything else is algebra, because z_p = w . patch_p + b is linear in the patch:
//   sum d z      = w . G + b Sd                                   -> sum d xhat = invstd (sum d z - mean Sd)
//   sum xhat patch[t] = invstd ((M2 w)[t] + (b - mean) mv[t])
//   dW[t] = scale (G[t] - m1 mv[t] - m2 sum xhat patch[t]),   m1 = Sd / P, m2 = sum d xhat / P;   dbeta = Sd, dgamma = sum d xhat
// One block per output channel; thread = (tap, slab group) for the slab sum (fixed order), then 27 threads finish in double.
__global__ __launch_bounds__(256) void conv1_bn_bwd_finish_kernel(const float* __restrict__ slabs, int nslab, const double* __restrict__ redc,
                                                                  const double* __restrict__ M2, const float* __restrict__ W,
                                                                  const float* __restrict__ bias, const float* __restrict__ scale,
                                                                  const float* __restrict__ mean, const float* __restrict__ invstd,
                                                                  float* __restrict__ dW, float* __restrict__ dgamma,
                                                                  float* __restrict__ dbeta, double* __restrict__ red_out, int Cout) {
  __shared__ float part[8][32];
  __shared__ double G[27], wv[27], sdz;
  const int co = blockIdx.x, tap = threadIdx.x & 31, grp = threadIdx.x >> 5;
  float s = 0.f;
  for (int w = grp; w < nslab; w += 8) s += slabs[((size_t)w * Cout + co) * 32 + tap];
  part[grp][tap] = s;
  __syncthreads();
  if (threadIdx.x < 27) {
    float v = 0.f;
#pragma unroll
    for (int g = 0; g < 8; ++g) v += part[g][threadIdx.x];
    G[threadIdx.x] = (double)v;
    wv[threadIdx.x] = (double)W[co * 27 + threadIdx.x];
  }
  __syncthreads();
  double Sd = 0.0;
  for (int c = 0; c < CY_STATS_COPIES; ++c) Sd += redc[((size_t)c * Cout + co) * 2];
  const double b = bias != nullptr ? (double)bias[co] : 0.0, mu = (double)mean[co], is = (double)invstd[co];
  if (threadIdx.x == 0) {
    double wg = 0.0;
    for (int t = 0; t < 27; ++t) wg += wv[t] * G[t];
    sdz = wg + b * Sd;
  }
  __syncthreads();
  const double P = M2[27 * 32 + 27];
  const double sdx = is * (sdz - mu * Sd);
  if (threadIdx.x == 0) {
    dbeta[co] = (float)Sd;
    dgamma[co] = (float)sdx;
    if (red_out != nullptr) { red_out[2 * co] = Sd; red_out[2 * co + 1] = sdx; }
  }
  if (threadIdx.x < 27) {
    const int t = threadIdx.x;
    double m2w = 0.0;
    for (int u = 0; u < 27; ++u) m2w += M2[t * 32 + u] * wv[u];
    const double mvt = M2[27 * 32 + t];
    const double sxp = is * (m2w + (b - mu) * mvt);
    dW[co * 27 + t] = (float)((double)scale[co] * (G[t] - (Sd / P) * mvt - (sdx / P) * sxp));
  }
}

static int conv1_blocks(long long ntiles, long long* blocks, const char* who) {
  int dev = 0, ncu = 0;
  hipError_t he = hipGetDevice(&dev);
  if (he == hipSuccess) he = hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev);
  if (he != hipSuccess || ncu <= 0) return cy_set_error((int)he, "%s: cannot query the CU count: %s", who, hipGetErrorString(he));
  *blocks = (ntiles + 3) / 4;
  if (*blocks > ncu) *blocks = ncu;         // persistent: one wave per SIMD
  return 0;
}

}  // namespace

extern "C" long long cy_conv1_3x3_wgrad_ws_floats(int B, int H, int Wd, int Cout) {
  long long blocks = 0;
  if (B <= 0 || H <= 0 || Wd <= 0 || Wd % 32 || conv1_blocks((long long)B * H * (Wd / 32), &blocks, "cy_conv1_3x3_wgrad_ws_floats")) return -1;
  return blocks * 4 * Cout * 32;
}

extern "C" int cy_conv1_3x3_wgrad(const float* X, const float* dZ, float* dW, float* ws, int B, int H, int Wd, int Cout,
                                  void* stream) {
  CY_REQUIRE(X && dZ && dW && ws && B > 0 && H > 0 && Wd > 0, "cy_conv1_3x3_wgrad: bad arguments");
  CY_REQUIRE(Wd % 32 == 0, "cy_conv1_3x3_wgrad: W=%d must be a multiple of 32", Wd);
  CY_REQUIRE(Cout == 32 || Cout == 64 || Cout == 128, "cy_conv1_3x3_wgrad: Cout=%d must be 32, 64 or 128", Cout);
  CY_REQUIRE((long long)3 * H * Wd < (1ll << 30), "cy_conv1_3x3_wgrad: image too large for 32-bit offsets");
  Conv1Args a;
  a.X = X; a.W = nullptr; a.bias = nullptr; a.Y = const_cast<float*>(dZ); a.stats = nullptr;
  a.scale = a.shift = nullptr; a.slope = 1.f;
  a.B = B; a.H = H; a.Wd = Wd; a.Cout = Cout;
  a.ntiles = (long long)B * H * (Wd / 32);
  long long blocks = 0;
  int rc = conv1_blocks(a.ntiles, &blocks, "cy_conv1_3x3_wgrad");
  if (rc) return rc;
  hipStream_t s = (hipStream_t)stream;
  if (Cout == 128) conv1_wgrad_kernel<4><<<(unsigned)blocks, 256, 0, s>>>(a, ws);
  else if (Cout == 64) conv1_wgrad_kernel<2><<<(unsigned)blocks, 256, 0, s>>>(a, ws);
  else conv1_wgrad_kernel<1><<<(unsigned)blocks, 256, 0, s>>>(a, ws);
  CY_LAUNCH_CHECK("cy_conv1_3x3_wgrad");
  conv1_wgrad_finish_kernel<<<Cout, 256, 0, s>>>(ws, dW, (int)(blocks * 4), Cout);
  CY_LAUNCH_CHECK("cy_conv1_3x3_wgrad (finish)");
  return 0;
}

static int conv1_bn_check(const char* who, const float* X, const float* W, const float* dA, const float* scale,
                          const float* shift, const float* mean, const float* invstd, float slope, int B, int H, int Wd,
                          int Cout) {
  CY_REQUIRE(X && W && dA && scale && shift && mean && invstd && B > 0 && H > 0 && Wd > 0, "%s: bad arguments", who);
  CY_REQUIRE(Wd % 32 == 0, "%s: W=%d must be a multiple of 32", who, Wd);
  CY_REQUIRE(Cout == 32 || Cout == 64 || Cout == 128, "%s: Cout=%d must be 32, 64 or 128", who, Cout);
  CY_REQUIRE((long long)3 * H * Wd < (1ll << 30), "%s: image too large for 32-bit offsets", who);
  CY_REQUIRE(slope >= 0.f && slope <= 1.f, "%s: slope must be in [0, 1]", who);
  return 0;
}

static int conv1_bn_bwd_reduce_impl(const char* who, const float* X, const float* W, const float* bias, const void* dA,
                                    bool da_bf16, const float* scale, const float* shift, const float* mean, const float* invstd,
                                    float slope, double* red, int B, int H, int Wd, int Cout, void* stream) {
  int rc = conv1_bn_check(who, X, W, (const float*)dA, scale, shift, mean, invstd, slope, B, H, Wd, Cout);
  if (rc) return rc;
  CY_REQUIRE(red != nullptr, "%s: red is NULL", who);
  Conv1BnArgs a;
  a.X = X; a.W = W; a.bias = bias; a.dA = (const float*)dA; a.scale = scale; a.shift = shift; a.mean = mean; a.invstd = invstd;
  a.slope = slope; a.red_out = red; a.red_in = nullptr; a.inv_count = 0.0; a.slabs = nullptr;
  a.B = B; a.H = H; a.Wd = Wd; a.Cout = Cout; a.ntiles = (long long)B * H * (Wd / 32);
  long long blocks = 0;
  rc = conv1_blocks(a.ntiles, &blocks, who);
  if (rc) return rc;
  hipStream_t s = (hipStream_t)stream;
  if (da_bf16) {
#ifdef CY_C1_F32MM
    if (Cout == 128) conv1_bn_bwd_kernel<4, 0, true><<<(unsigned)blocks, 256, 0, s>>>(a);
    else if (Cout == 64) conv1_bn_bwd_kernel<2, 0, true><<<(unsigned)blocks, 256, 0, s>>>(a);
    else conv1_bn_bwd_kernel<1, 0, true><<<(unsigned)blocks, 256, 0, s>>>(a);
#else
    if (Cout == 128) conv1_bn_bwd_bf16mm_kernel<4, 0><<<(unsigned)blocks, 256, 0, s>>>(a);
    else if (Cout == 64) conv1_bn_bwd_bf16mm_kernel<2, 0><<<(unsigned)blocks, 256, 0, s>>>(a);
    else conv1_bn_bwd_bf16mm_kernel<1, 0><<<(unsigned)blocks, 256, 0, s>>>(a);
#endif
  } else {
    if (Cout == 128) conv1_bn_bwd_kernel<4, 0><<<(unsigned)blocks, 256, 0, s>>>(a);
    else if (Cout == 64) conv1_bn_bwd_kernel<2, 0><<<(unsigned)blocks, 256, 0, s>>>(a);
    else conv1_bn_bwd_kernel<1, 0><<<(unsigned)blocks, 256, 0, s>>>(a);
  }
  CY_LAUNCH_CHECK(who);
  return 0;
}
extern "C" int cy_conv1_bn_bwd_reduce(const float* X, const float* W, const float* bias, const float* dA, const float* scale,
                                      const float* shift, const float* mean, const float* invstd, float slope, double* red,
                                      int B, int H, int Wd, int Cout, void* stream) {
  return conv1_bn_bwd_reduce_impl("cy_conv1_bn_bwd_reduce", X, W, bias, dA, false, scale, shift, mean, invstd, slope, red, B, H, Wd,
                                  Cout, stream);
}
extern "C" int cy_conv1_bn_bwd_reduce_bf16(const float* X, const float* W, const float* bias, const void* dA, const float* scale,
                                           const float* shift, const float* mean, const float* invstd, float slope, double* red,
                                           int B, int H, int Wd, int Cout, void* stream) {
  return conv1_bn_bwd_reduce_impl("cy_conv1_bn_bwd_reduce_bf16", X, W, bias, dA, true, scale, shift, mean, invstd, slope, red, B, H,
                                  Wd, Cout, stream);
}

// One pass over dA instead of two (cy_conv1_bn_bwd_reduce + cy_conv1_bn_bwd_wgrad): see conv1_bn_bwd_finish_kernel
static int conv1_bn_bwd_onepass_impl(const char* who, const float* X, const float* W, const float* bias, const void* dA, bool da_bf16,
                                     const float* scale, const float* shift, const float* mean, const float* invstd, float slope,
                                     const double* M2, double* redc, float* dW, float* dgamma, float* dbeta, double* red_out,
                                     float* ws, int B, int H, int Wd, int Cout, void* stream) {
  int rc = conv1_bn_check(who, X, W, (const float*)dA, scale, shift, mean, invstd, slope, B, H, Wd, Cout);
  if (rc) return rc;
  CY_REQUIRE(M2 && redc && dW && dgamma && dbeta && ws, "%s: NULL argument", who);
  Conv1BnArgs a;
  a.X = X; a.W = W; a.bias = bias; a.dA = (const float*)dA; a.scale = scale; a.shift = shift; a.mean = mean; a.invstd = invstd;
  a.slope = slope; a.red_out = redc; a.red_in = nullptr; a.inv_count = 0.0; a.slabs = ws;
  a.B = B; a.H = H; a.Wd = Wd; a.Cout = Cout; a.ntiles = (long long)B * H * (Wd / 32);
  long long blocks = 0;
  rc = conv1_blocks(a.ntiles, &blocks, who);
  if (rc) return rc;
  hipStream_t s = (hipStream_t)stream;
  if (da_bf16) {
#ifdef CY_C1_F32MM
    if (Cout == 128) conv1_bn_bwd_kernel<4, 2, true><<<(unsigned)blocks, 256, 0, s>>>(a);
    else if (Cout == 64) conv1_bn_bwd_kernel<2, 2, true><<<(unsigned)blocks, 256, 0, s>>>(a);
    else conv1_bn_bwd_kernel<1, 2, true><<<(unsigned)blocks, 256, 0, s>>>(a);
#else
    if (Cout == 128) conv1_bn_bwd_bf16mm_kernel<4, 2><<<(unsigned)blocks, 256, 0, s>>>(a);
    else if (Cout == 64) conv1_bn_bwd_bf16mm_kernel<2, 2><<<(unsigned)blocks, 256, 0, s>>>(a);
    else conv1_bn_bwd_bf16mm_kernel<1, 2><<<(unsigned)blocks, 256, 0, s>>>(a);
#endif
  } else {
    if (Cout == 128) conv1_bn_bwd_kernel<4, 2><<<(unsigned)blocks, 256, 0, s>>>(a);
    else if (Cout == 64) conv1_bn_bwd_kernel<2, 2><<<(unsigned)blocks, 256, 0, s>>>(a);
    else conv1_bn_bwd_kernel<1, 2><<<(unsigned)blocks, 256, 0, s>>>(a);
  }
  CY_LAUNCH_CHECK(who);
  conv1_bn_bwd_finish_kernel<<<Cout, 256, 0, s>>>(ws, (int)(blocks * 4), redc, M2, W, bias, scale, mean, invstd, dW, dgamma, dbeta,
                                                  red_out, Cout);
  CY_LAUNCH_CHECK(who);
  return 0;
}
extern "C" int cy_conv1_bn_bwd_onepass(const float* X, const float* W, const float* bias, const float* dA, const float* scale,
                                       const float* shift, const float* mean, const float* invstd, float slope, const double* M2,
                                       double* redc, float* dW, float* dgamma, float* dbeta, double* red_out, float* ws, int B,
                                       int H, int Wd, int Cout, void* stream) {
  return conv1_bn_bwd_onepass_impl("cy_conv1_bn_bwd_onepass", X, W, bias, dA, false, scale, shift, mean, invstd, slope, M2, redc, dW,
                                   dgamma, dbeta, red_out, ws, B, H, Wd, Cout, stream);
}
extern "C" int cy_conv1_bn_bwd_onepass_bf16(const float* X, const float* W, const float* bias, const void* dA, const float* scale,
                                            const float* shift, const float* mean, const float* invstd, float slope,
                                            const double* M2, double* redc, float* dW, float* dgamma, float* dbeta, double* red_out,
                                            float* ws, int B, int H, int Wd, int Cout, void* stream) {
  return conv1_bn_bwd_onepass_impl("cy_conv1_bn_bwd_onepass_bf16", X, W, bias, dA, true, scale, shift, mean, invstd, slope, M2, redc,
                                   dW, dgamma, dbeta, red_out, ws, B, H, Wd, Cout, stream);
}

extern "C" long long cy_conv1_bn_bwd_wgrad_ws_floats(int B, int H, int Wd, int Cout) {
  return cy_conv1_3x3_wgrad_ws_floats(B, H, Wd, Cout);
}

static int conv1_bn_bwd_wgrad_impl(const char* who, bool da_bf16, const float* X, const float* W, const float* bias, const void* dA, const float* scale,
                                     const float* shift, const float* mean, const float* invstd, float slope,
                                     const double* red, long long count, float* dW, float* ws, int B, int H, int Wd,
                                     int Cout, void* stream) {
  int rc = conv1_bn_check(who, X, W, (const float*)dA, scale, shift, mean, invstd, slope, B, H, Wd, Cout);
  if (rc) return rc;
  CY_REQUIRE(red && dW && ws && count > 0, "cy_conv1_bn_bwd_wgrad: bad arguments");
  Conv1BnArgs a;
  a.X = X; a.W = W; a.bias = bias; a.dA = (const float*)dA; a.scale = scale; a.shift = shift; a.mean = mean; a.invstd = invstd;
  a.slope = slope; a.red_out = nullptr; a.red_in = red; a.inv_count = 1.0 / (double)count; a.slabs = ws;
  a.B = B; a.H = H; a.Wd = Wd; a.Cout = Cout; a.ntiles = (long long)B * H * (Wd / 32);
  long long blocks = 0;
  rc = conv1_blocks(a.ntiles, &blocks, who);
  if (rc) return rc;
  hipStream_t s = (hipStream_t)stream;
  if (da_bf16) {
#ifdef CY_C1_F32MM
    if (Cout == 128) conv1_bn_bwd_kernel<4, 1, true><<<(unsigned)blocks, 256, 0, s>>>(a);
    else if (Cout == 64) conv1_bn_bwd_kernel<2, 1, true><<<(unsigned)blocks, 256, 0, s>>>(a);
    else conv1_bn_bwd_kernel<1, 1, true><<<(unsigned)blocks, 256, 0, s>>>(a);
#else
    if (Cout == 128) conv1_bn_bwd_bf16mm_kernel<4, 1><<<(unsigned)blocks, 256, 0, s>>>(a);
    else if (Cout == 64) conv1_bn_bwd_bf16mm_kernel<2, 1><<<(unsigned)blocks, 256, 0, s>>>(a);
    else conv1_bn_bwd_bf16mm_kernel<1, 1><<<(unsigned)blocks, 256, 0, s>>>(a);
#endif
  } else {
    if (Cout == 128) conv1_bn_bwd_kernel<4, 1><<<(unsigned)blocks, 256, 0, s>>>(a);
    else if (Cout == 64) conv1_bn_bwd_kernel<2, 1><<<(unsigned)blocks, 256, 0, s>>>(a);
    else conv1_bn_bwd_kernel<1, 1><<<(unsigned)blocks, 256, 0, s>>>(a);
  }
  CY_LAUNCH_CHECK(who);
  conv1_wgrad_finish_kernel<<<Cout, 256, 0, s>>>(ws, dW, (int)(blocks * 4), Cout);
  CY_LAUNCH_CHECK(who);
  return 0;
}

extern "C" int cy_conv1_bn_bwd_wgrad(const float* X, const float* W, const float* bias, const float* dA, const float* scale,
                                     const float* shift, const float* mean, const float* invstd, float slope,
                                     const double* red, long long count, float* dW, float* ws, int B, int H, int Wd,
                                     int Cout, void* stream) {
  return conv1_bn_bwd_wgrad_impl("cy_conv1_bn_bwd_wgrad", false, X, W, bias, dA, scale, shift, mean, invstd, slope, red, count, dW, ws, B,
                                 H, Wd, Cout, stream);
}
extern "C" int cy_conv1_bn_bwd_wgrad_bf16(const float* X, const float* W, const float* bias, const void* dA, const float* scale,
                                          const float* shift, const float* mean, const float* invstd, float slope,
                                          const double* red, long long count, float* dW, float* ws, int B, int H, int Wd,
                                          int Cout, void* stream) {
  return conv1_bn_bwd_wgrad_impl("cy_conv1_bn_bwd_wgrad_bf16", true, X, W, bias, dA, scale, shift, mean, invstd, slope, red, count, dW,
                                 ws, B, H, Wd, Cout, stream);
}

extern "C" int cy_conv1_3x3_fwd(const float* X, const float* W, const float* bias, float* Y, double* stats,
                                const float* scale, const float* shift, float slope, int B, int H, int Wd, int Cout,
                                void* stream) {
  CY_REQUIRE(X && W && (Y || (stats && !scale)) && B > 0 && H > 0 && Wd > 0, "cy_conv1_3x3_fwd: bad arguments");
  CY_REQUIRE((scale == nullptr) == (shift == nullptr), "cy_conv1_3x3_fwd: scale and shift go together");
  CY_REQUIRE(scale == nullptr || (stats == nullptr && slope >= 0.f && slope <= 1.f),
             "cy_conv1_3x3_fwd: the affine pass takes no statistics and a slope in [0, 1]");
  CY_REQUIRE(Wd % 32 == 0, "cy_conv1_3x3_fwd: W=%d must be a multiple of 32", Wd);
  CY_REQUIRE(Cout == 32 || Cout == 64 || Cout == 128, "cy_conv1_3x3_fwd: Cout=%d must be 32, 64 or 128", Cout);
  CY_REQUIRE((long long)3 * H * Wd < (1ll << 30), "cy_conv1_3x3_fwd: image too large for 32-bit offsets");
  Conv1Args a;
  a.X = X; a.W = W; a.bias = bias; a.Y = Y; a.stats = stats;
  a.scale = scale; a.shift = shift; a.slope = slope;
  a.B = B; a.H = H; a.Wd = Wd; a.Cout = Cout;
  a.ntiles = (long long)B * H * (Wd / 32);
  long long blocks = 0;
  int rc = conv1_blocks(a.ntiles, &blocks, "cy_conv1_3x3_fwd");
  if (rc) return rc;
  hipStream_t s = (hipStream_t)stream;
#define CY_CONV1_LAUNCH(NT_)                                                                  \
  do {                                                                                         \
    if (scale) conv1_fwd_kernel<NT_, true, true><<<(unsigned)blocks, 256, 0, s>>>(a);          \
    else if (Y) conv1_fwd_kernel<NT_, false, true><<<(unsigned)blocks, 256, 0, s>>>(a);        \
    else conv1_fwd_kernel<NT_, false, false><<<(unsigned)blocks, 256, 0, s>>>(a);              \
  } while (0)
  if (Cout == 128) CY_CONV1_LAUNCH(4);
  else if (Cout == 64) CY_CONV1_LAUNCH(2);
  else CY_CONV1_LAUNCH(1);
#undef CY_CONV1_LAUNCH
  CY_LAUNCH_CHECK("cy_conv1_3x3_fwd");
  return 0;
}

// the activation pass of the first block with a bf16 output (Y: bf16 [B][H][W][Cout]); scale / shift are mandatory
extern "C" int cy_conv1_3x3_fwd_act_bf16(const float* X, const float* W, const float* bias, void* Y, const float* scale,
                                         const float* shift, float slope, int B, int H, int Wd, int Cout, void* stream) {
  CY_REQUIRE(X && W && Y && scale && shift && B > 0 && H > 0 && Wd > 0, "cy_conv1_3x3_fwd_act_bf16: bad arguments");
  CY_REQUIRE(slope >= 0.f && slope <= 1.f, "cy_conv1_3x3_fwd_act_bf16: slope must be in [0, 1]");
  CY_REQUIRE(Wd % 32 == 0, "cy_conv1_3x3_fwd_act_bf16: W=%d must be a multiple of 32", Wd);
  CY_REQUIRE(Cout == 32 || Cout == 64 || Cout == 128, "cy_conv1_3x3_fwd_act_bf16: Cout=%d must be 32, 64 or 128", Cout);
  CY_REQUIRE((long long)3 * H * Wd < (1ll << 30), "cy_conv1_3x3_fwd_act_bf16: image too large for 32-bit offsets");
  Conv1Args a;
  a.X = X; a.W = W; a.bias = bias; a.Y = (float*)Y; a.stats = nullptr;
  a.scale = scale; a.shift = shift; a.slope = slope;
  a.B = B; a.H = H; a.Wd = Wd; a.Cout = Cout;
  a.ntiles = (long long)B * H * (Wd / 32);
  long long blocks = 0;
  int rc = conv1_blocks(a.ntiles, &blocks, "cy_conv1_3x3_fwd_act_bf16");
  if (rc) return rc;
  hipStream_t s = (hipStream_t)stream;
  // (-DCY_C1_F32MM: the fp32 matrix cores with a bf16 store, the round-2 form -- a developer switch for A/B timing)
#ifdef CY_C1_F32MM
  if (Cout == 128) conv1_fwd_kernel<4, true, true, true><<<(unsigned)blocks, 256, 0, s>>>(a);
  else if (Cout == 64) conv1_fwd_kernel<2, true, true, true><<<(unsigned)blocks, 256, 0, s>>>(a);
  else conv1_fwd_kernel<1, true, true, true><<<(unsigned)blocks, 256, 0, s>>>(a);
#else
  if (Cout == 128) conv1_fwd_bf16_kernel<4><<<(unsigned)blocks, 256, 0, s>>>(a);
  else if (Cout == 64) conv1_fwd_bf16_kernel<2><<<(unsigned)blocks, 256, 0, s>>>(a);
  else conv1_fwd_bf16_kernel<1><<<(unsigned)blocks, 256, 0, s>>>(a);
#endif
  CY_LAUNCH_CHECK("cy_conv1_3x3_fwd_act_bf16");
  return 0;
}
