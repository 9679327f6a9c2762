// BatchNorm statistics of the first layer WITHOUT computing the layer (gfx950).
//
// conv_1 is z[p][co] = w_co . patch(p) + b_co with patch(p) in R^27 (3 channels x 3 x 3 neighbourhood, zeros outside the
// image), so the per-channel sums BatchNorm needs are functions of the patch moments alone:
//     sum_p z      = w_co . m1 + P b_co
//     sum_p z^2    = w_co^T M2 w_co + 2 b_co (w_co . m1) + P b_co^2,      m1 = sum_p patch(p),  M2 = sum_p patch(p) patch(p)^T
// M2 (with a constant-one 28th patch entry that carries m1 and P along) is ONE 28 x 28 Gram matrix over all pixels: a GEMM
// with the pixels as the reduction dimension whose two operands are the same matrix -- v_mfma_f32_32x32x2f32 with
// A = B = (tap l % 32 of pixel 2 i + l / 32): 16 MFMAs per 32 pixels instead of the 14 x Cout / 32 (= 56 for 128 channels)
// of the statistics pass that recomputes z (csrc/conv1.hip), whatever Cout is.  Replaces that pass of
// models.py:347-349 / 132-136 (Conv2d -> BatchNorm2d in training mode) for the first layer.
#include "common.h"

namespace {

constexpr int MOM = 32 * 32;                // floats of one partial Gram matrix (28 x 28 used)

// one block = 4 waves, one per SIMD; a wave walks 32-pixel row segments s = gw, gw + nw, ...  MFMA lane l: tap m = l & 31 of
// pixel 2 i + (l >> 5).  Gathering those operands straight from the image costs a 128-byte line per 6 lanes (the CU's 64 B/cycle
// L1 then sets the pace: 150 cycles per MFMA); instead the segment's 9 (channel, dy) rows x 34 columns are loaded coalesced,
// parked in the wave's own LDS as [column][9 rows] -- the 27 taps of a pixel are then 27 consecutive words, conflict-free --
// and every MFMA operand is ONE ds_read_b32.  LDS operations of a wave complete in order: no barrier anywhere in the loop.
constexpr int SEGW = 32, SEGE = 9 * (SEGW + 2), SEGQ = (SEGE + 63) / 64, SEGB = 320;   // elements, loads per lane, floats per buffer

// Exactness: a data-set image is k / 128 with |k| <= 128 (utils.py:122-123), so every product of two taps is a multiple of 2^-14 of
// magnitude <= 1 and an fp32 accumulator holds the sum of up to 1024 of them EXACTLY (< 2^10 at a granularity of 2^-14: 24 bits).
// The MFMA accumulators are therefore flushed into per-lane doubles every FLUSH_SEGS segments (16 pixels per accumulator and
// segment): M2 comes out exact for such images whatever their mean or spatial correlation (bright, smooth images with
// zero-sum filters make w^T M2 w - (w . m1)^2 / P cancel by three to four digits: an fp32-accumulated M2 would leave 1e-3 of the
// variance there), and for general float images the fp32 chains are at most 960 terms long.
constexpr int FLUSH_SEGS = 60;

__global__ __launch_bounds__(256, 1) void conv1_moments_kernel(const float* __restrict__ X, double* __restrict__ part, int B, int H,
                                                              int W) {
  __shared__ float seg[4][2][SEGB];
  __shared__ double red[4][MOM];
  const int t = threadIdx.x, lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int m = lane & 31, k = lane >> 5;
  const bool tap = m < 27;
  const float fill = m == 27 ? 1.f : 0.f;   // the constant-one entry (carries m1 and the pixel count); entries 28..31 stay 0
  const float tapf = tap ? 1.f : 0.f;       // lanes without a tap read word 0 of the image (finite) and multiply it away
  const int rd0 = tap ? (k + m % 3) * 9 + m / 3 : 0;          // word of (pixel k, tap m) in the segment image; + 18 i for pixel pair i
  const int spr = W / SEGW;                                   // segments per row
  const long long nseg = (long long)B * H * spr;
  const int gw = blockIdx.x * 4 + wave, nw = gridDim.x * 4;
  // staging: element e = lane + 64 q of the segment = (row g = e / 34: channel g / 3, dy = g % 3 - 1; column j = e % 34 <-> x0 - 1 + j):
  // a lane's offset from the segment origin (image b, channel 0, row y, column x0) never changes
  int sdy[SEGQ], sj[SEGQ], slds[SEGQ];
  long long soff[SEGQ];
#pragma unroll
  for (int q = 0; q < SEGQ; ++q) {
    const int e = lane + 64 * q;
    const int g = e < SEGE ? e / (SEGW + 2) : 0;
    sj[q] = e < SEGE ? e % (SEGW + 2) : 1;           // (spare lanes of the last round: an element that is always inside)
    sdy[q] = e < SEGE ? g % 3 - 1 : 0;
    slds[q] = sj[q] * 9 + g;
    soff[q] = ((long long)(g / 3) * H + sdy[q]) * W + (sj[q] - 1);
  }
  // segment cursor (image, row, segment of the row), stepped by nw segments without divisions
  const int d_row = nw / spr, d_xs = nw % spr, d_b = d_row / H, d_y = d_row % H;
  int cb = (int)((gw / spr) / H), cy = (int)((gw / spr) % H), cxs = gw % spr;
  auto advance = [&]() {
    cxs += d_xs; cy += d_y; cb += d_b;
    if (cxs >= spr) { cxs -= spr; ++cy; }
    if (cy >= H) { cy -= H; ++cb; }
    if (cy >= H) { cy -= H; ++cb; }                  // (d_y + carry can reach 2 H - 1)
  };
  auto request = [&](float (&stg)[SEGQ]) {   // global -> registers (zeros outside the image) for the cursor's segment, then advance
    // ONE path, always SEGQ loads (the waits in front of park() can then be counted: behind a branch the compiler waited for
    // everything in flight, i.e. for the request issued a moment ago): elements outside the image load a clamped address and are
    // multiplied away; only segments at the image border pay for the clamping
    const float* org = X + ((long long)cb * 3 * H + cy) * W + cxs * SEGW;
    const bool inner = cy > 0 && cy + 1 < H && cxs > 0 && cxs + 1 < spr;   // uniform
#pragma unroll
    for (int q = 0; q < SEGQ; ++q) {
      long long off = soff[q];
      float okf = 1.f;
      if (!inner) {
        const int iy = cy + sdy[q], ix = cxs * SEGW - 1 + sj[q];
        const bool ok = (unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W;
        off = ok ? off : 0ll;                 // (the segment's origin itself is always inside)
        okf = ok ? 1.f : 0.f;
      }
      stg[q] = org[off] * okf;
    }
    advance();
  };
  auto park = [&](const float (&stg)[SEGQ], int buf) {   // registers -> the wave's LDS image [column][9 rows]
#pragma unroll
    for (int q = 0; q < SEGQ; ++q)
      if (lane + 64 * q < SEGE) seg[wave][buf][slds[q]] = stg[q];
  };
  f32x16 acc0, acc1;
  double dacc[16];
#pragma unroll
  for (int r = 0; r < 16; ++r) { acc0[r] = 0.f; acc1[r] = 0.f; dacc[r] = 0.0; }
  auto flush = [&]() {
#pragma unroll
    for (int r = 0; r < 16; ++r) { dacc[r] += (double)acc0[r] + (double)acc1[r]; acc0[r] = 0.f; acc1[r] = 0.f; }
  };
  // Segment n is consumed from LDS buffer n % 2; its successor is parked right behind its MFMAs, from a register set that was
  // requested THREE segments (~3 k cycles) earlier: every segment touches lines nobody on this CU has read yet.
  const long long nmine = gw < nseg ? (nseg - gw + nw - 1) / nw : 0;
  float st0[SEGQ], st1[SEGQ], st2[SEGQ];
  auto consume = [&](int buf) {
    const float* im = &seg[wave][buf][rd0];
    float v[SEGW / 2];
#pragma unroll
    for (int i = 0; i < SEGW / 2; ++i) v[i] = __builtin_fmaf(im[18 * i], tapf, fill);   // (a select here became a branch per read)
#pragma unroll
    for (int i = 0; i < SEGW / 2; i += 2) {
      acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(v[i], v[i], acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(v[i + 1], v[i + 1], acc1, 0, 0, 0);
    }
  };
  if (nmine > 0) { request(st0); park(st0, 0); }
  if (nmine > 1) request(st1);
  if (nmine > 2) request(st2);
  long long n = 0;
  int since_flush = 0;
  // steady state, straight-line (6 = 2 LDS buffers x 3 register sets): every request, consume and park of these steps exists
  for (; n + 9 <= nmine; n += 6) {
#define CY_MOM_STEP(K, REQ, PARK) request(REQ); consume((K) & 1); park(PARK, ((K) + 1) & 1);
    CY_MOM_STEP(0, st0, st1) CY_MOM_STEP(1, st1, st2) CY_MOM_STEP(2, st2, st0)
    CY_MOM_STEP(3, st0, st1) CY_MOM_STEP(4, st1, st2) CY_MOM_STEP(5, st2, st0)
#undef CY_MOM_STEP
    since_flush += 6;
    if (since_flush >= FLUSH_SEGS - 14) { flush(); since_flush = 0; }     // (the guarded tail below adds up to 14 more segments)
  }
  for (; n < nmine; n += 6) {                 // the last steps, guarded
#define CY_MOM_STEP(K, REQ, PARK)                                              \
    if (n + (K) < nmine) {                                                     \
      if (n + (K) + 3 < nmine) request(REQ);                                   \
      consume((K) & 1);                                                        \
      if (n + (K) + 1 < nmine) park(PARK, ((K) + 1) & 1);                      \
    }
    CY_MOM_STEP(0, st0, st1) CY_MOM_STEP(1, st1, st2) CY_MOM_STEP(2, st2, st0)
    CY_MOM_STEP(3, st0, st1) CY_MOM_STEP(4, st1, st2) CY_MOM_STEP(5, st2, st0)
#undef CY_MOM_STEP
  }
  // D[row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5)][col = lane & 31]
  flush();
#pragma unroll
  for (int r = 0; r < 16; ++r) red[wave][((r & 3) + 8 * (r >> 2) + 4 * k) * 32 + m] = dacc[r];
  __syncthreads();
  for (int e = t; e < MOM; e += 256) part[(long long)blockIdx.x * MOM + e] = (red[0][e] + red[1][e]) + (red[2][e] + red[3][e]);
}

// M2[e] = sum over the blocks' partial matrices, in double: block j -> entries 16 j .. 16 j + 15, thread = (entry, partial group)
__global__ __launch_bounds__(256) void conv1_moments_sum_kernel(const double* __restrict__ part, int nblocks, double* __restrict__ M2) {
  __shared__ double red[16][17];
  const int e = blockIdx.x * 16 + (threadIdx.x & 15), gsub = threadIdx.x >> 4;
  double s = 0.0;
  for (int p = gsub; p < nblocks; p += 16) s += part[(long long)p * MOM + e];
  red[threadIdx.x & 15][gsub] = s;
  __syncthreads();
  if (threadIdx.x < 16) {
    double a = 0.0;
#pragma unroll
    for (int g = 0; g < 16; ++g) a += red[threadIdx.x][g];
    M2[blockIdx.x * 16 + threadIdx.x] = a;
  }
}

// stats[co] += (sum z, sum z^2) from M2, the weights W[co][27] and the bias
__global__ __launch_bounds__(256) void conv1_moments_stats_kernel(const double* __restrict__ M2, const float* __restrict__ W,
                                                                 const float* __restrict__ bias, double* __restrict__ stats, int Cout) {
  __shared__ double m[28][28];
  for (int e = threadIdx.x; e < 28 * 28; e += 256) m[e / 28][e % 28] = M2[(e / 28) * 32 + e % 28];
  __syncthreads();
  const int co = blockIdx.x * 64 + threadIdx.x;
  if (threadIdx.x >= 64 || co >= Cout) return;
  double w[27];
#pragma unroll
  for (int i = 0; i < 27; ++i) w[i] = (double)W[co * 27 + i];
  const double b = bias != nullptr ? (double)bias[co] : 0.0, P = m[27][27];
  double wm1 = 0.0, q = 0.0;
#pragma unroll
  for (int i = 0; i < 27; ++i) {            // (fully unrolled: w[] stays in registers)
    wm1 += w[i] * m[27][i];
    double rowq = 0.0;
#pragma unroll
    for (int j = 0; j < 27; ++j) rowq += m[i][j] * w[j];
    q += w[i] * rowq;
  }
  atomicAdd(stats + 2 * co, wm1 + P * b);
  atomicAdd(stats + 2 * co + 1, q + 2.0 * b * wm1 + P * b * b);
}

int moments_blocks(long long rows, long long* blocks, const char* who) {
  int dev = 0, ncu = 0;
  hipError_t he = hipGetDevice(&dev);
  if (he == hipSuccess) he = hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev);
  if (he != hipSuccess || ncu <= 0) return cy_set_error((int)he, "%s: cannot query the CU count: %s", who, hipGetErrorString(he));
  *blocks = (rows + 3) / 4;
  if (*blocks > ncu) *blocks = ncu;
  return 0;
}

}  // namespace

extern "C" long long cy_conv1_3x3_stats_ws_floats(int B, int H) {
  long long blocks = 0;
  if (B <= 0 || H <= 0 || moments_blocks((long long)B * H, &blocks, "cy_conv1_3x3_stats_ws_floats")) return -1;
  return 2 * blocks * MOM + 2 * MOM;          // the blocks' partial matrices and M2, all in double
}

// float offset of the double M2[32][32] (row 27 = sum patch, [27][27] = pixel count) inside ws after cy_conv1_3x3_stats: the one-pass
// backward of the block (cy_conv1_bn_bwd_onepass) reads it
extern "C" long long cy_conv1_3x3_stats_m2_offset(int B, int H) {
  long long blocks = 0;
  if (B <= 0 || H <= 0 || moments_blocks((long long)B * H, &blocks, "cy_conv1_3x3_stats_m2_offset")) return -1;
  return 2 * blocks * MOM;
}

extern "C" int cy_conv1_3x3_stats(const float* X, const float* W, const float* bias, double* stats, float* ws, int B, int H, int Wd,
                                  int Cout, void* stream) {
  CY_REQUIRE(X && W && stats && ws && B > 0 && H > 0 && Wd > 0 && Cout > 0, "cy_conv1_3x3_stats: bad arguments");
  CY_REQUIRE(Wd % 32 == 0, "cy_conv1_3x3_stats: W=%d must be a multiple of 32", Wd);
  CY_REQUIRE((long long)3 * H * Wd < (1ll << 30), "cy_conv1_3x3_stats: image too large for 32-bit offsets");
  CY_REQUIRE(((uintptr_t)ws & 7) == 0, "cy_conv1_3x3_stats: ws must be 8-byte aligned");
  long long blocks = 0;
  int rc = moments_blocks((long long)B * H, &blocks, "cy_conv1_3x3_stats");
  if (rc) return rc;
  hipStream_t s = (hipStream_t)stream;
  double* M2 = (double*)(ws + 2 * blocks * MOM);
  conv1_moments_kernel<<<(unsigned)blocks, 256, 0, s>>>(X, (double*)ws, B, H, Wd);
  CY_LAUNCH_CHECK("cy_conv1_3x3_stats(moments)");
  conv1_moments_sum_kernel<<<MOM / 16, 256, 0, s>>>((const double*)ws, (int)blocks, M2);
  CY_LAUNCH_CHECK("cy_conv1_3x3_stats(sum)");
  conv1_moments_stats_kernel<<<(Cout + 63) / 64, 256, 0, s>>>(M2, W, bias, stats, Cout);
  CY_LAUNCH_CHECK("cy_conv1_3x3_stats(stats)");
  return 0;
}
