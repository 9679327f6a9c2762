// Small data-movement / elementwise kernels around the hot path (gfx950): layout permutes,
// MaxPool2d(2), nearest Upsample, the YOLO head activation, tanh, capsule gather.
// Replaces models.py:8-19 (Flatten/UnFlatten views), 80-82 (primary-capsule view+cat), 99-105
// (nn.Upsample), 122 (torch.gather of the true capsule), 135-195 (nn.MaxPool2d), 226-236 (head).
#include "common.h"

namespace {

inline unsigned grid_for(long long n) {
  long long b = cy_ceil_div(n, 256);
  if (b > 8192) b = 8192;
  if (b < 1) b = 1;
  return (unsigned)b;
}
#define CY_GRID_STRIDE(i, n) \
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < (n); i += (long long)gridDim.x * blockDim.x)

// uint8 NHWC image batch -> centred fp32 (x - 128) / 128 (utils.py:122-123), NCHW (what model.forward takes,
// main.py:57-59) or NHWC.  One thread converts 4 consecutive bytes of one pixel row segment.
__global__ void center_u8_kernel(const unsigned char* __restrict__ src, float* __restrict__ dst, long long npix, int C,
                                 long long HW, int to_nchw) {
  CY_GRID_STRIDE(i, npix * C) {
    const float v = ((float)src[i] - 128.0f) * 0.0078125f;
    if (!to_nchw) { dst[i] = v; continue; }
    const long long p = i / C; const int c = (int)(i - p * C);
    const long long b = p / HW, r = p - b * HW;
    dst[(b * C + c) * HW + r] = v;
  }
}

// y_to_boxes_vec (utils.py:288-334) on the device: boxes whose confidence exceeds the threshold, in np.argwhere order
// (image, row, col, box), de-normalised to pixels (utils.py:233-252) and converted to corners (utils.py:254-269),
// class = first arg-max of the cell's class scores.  One block: the candidates (B g g nb, 10 816 at B=32 / g=13) are
// swept in chunks of 1024 with an order-preserving block scan.  Arithmetic in double, like the reference's numpy.
__global__ __launch_bounds__(1024) void yolo_decode_kernel(const float* __restrict__ y, const long long* __restrict__ image_hw,
                                                           double img_h, double img_w, int B, int g, int nb, int C, float conf_th,
                                                           int* count, int* image_idx, double* xy, int* cls, int max_boxes) {
  __shared__ int wave_cnt[16];
  __shared__ int base_s;
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int D = 5 * nb + C;
  const long long total = (long long)B * g * g * nb;
  if (t == 0) base_s = 0;
  __syncthreads();
  for (long long c0 = 0; c0 < total; c0 += 1024) {
    const long long i = c0 + t;
    bool hit = false;
    int bi = 0, row = 0, col = 0, k = 0;
    if (i < total) {
      long long r = i;
      k = (int)(r % nb); r /= nb;
      col = (int)(r % g); r /= g;
      row = (int)(r % g); bi = (int)(r / g);
      hit = y[(((long long)bi * g + row) * g + col) * D + 5 * k] > conf_th;
    }
    const unsigned long long m = __ballot(hit);
    if (lane == 0) wave_cnt[wave] = __popcll(m);
    __syncthreads();
    int before = 0, chunk_total = 0;
    for (int w = 0; w < 16; ++w) { const int cnt = wave_cnt[w]; if (w < wave) before += cnt; chunk_total += cnt; }
    const int base = base_s;
    if (hit) {
      const int o = base + before + __popcll(m & ((1ull << lane) - 1ull));
      if (o < max_boxes) {
        const float* cell = y + (((long long)bi * g + row) * g + col) * D;
        const double ih = image_hw ? (double)image_hw[2 * bi] : img_h, iw = image_hw ? (double)image_hw[2 * bi + 1] : img_w;
        const double gw = 1.0 * iw / g, gh = 1.0 * ih / g;
        double xc = (double)cell[5 * k + 1] * gw, yc = (double)cell[5 * k + 2] * gh;
        const double w_ = (double)cell[5 * k + 3] * iw, h_ = (double)cell[5 * k + 4] * ih;
        xc += col * gw; yc += row * gh;
        image_idx[o] = bi;
        xy[4 * o + 0] = xc - w_ / 2; xy[4 * o + 1] = yc - h_ / 2;
        xy[4 * o + 2] = xc + w_ / 2; xy[4 * o + 3] = yc + h_ / 2;
        if (C > 0) {
          int best = 0; float bv = cell[5 * nb];
          for (int c = 1; c < C; ++c) { const float v = cell[5 * nb + c]; if (v > bv) { bv = v; best = c; } }
          cls[o] = best;
        }
      }
    }
    __syncthreads();
    if (t == 0) base_s = base + chunk_total;
    __syncthreads();
  }
  if (t == 0) *count = base_s;
}

// metrics.single_img_confusion (metrics.py:136-147 with calc_iou_individual 99-133) for every image of a batch: one block
// per image gathers its ground-truth and predicted boxes (decoded by yolo_decode_kernel, sorted by image) into LDS,
// tests all pairs (IoU in double, like the reference) and counts the boxes that found a partner.
// out[0..2] += TP, FP, FN; out[3] += malformed boxes (x1 > x2 or y1 > y2: the reference raises on those).
__global__ __launch_bounds__(256) void detect_confusion_kernel(const int* __restrict__ gt_idx, const double* __restrict__ gt_xy, int n_gt,
                                                               const int* __restrict__ pr_idx, const double* __restrict__ pr_xy, int n_pr,
                                                               double iou_th, int max_per_image, int* out) {
  extern __shared__ __attribute__((aligned(16))) double sm[];
  double* gxy = sm;                                   // [max_per_image][4]
  double* pxy = sm + (size_t)max_per_image * 4;       // [max_per_image][4]
  int* flags = (int*)(pxy + (size_t)max_per_image * 4);   // [2][max_per_image] hit flags
  __shared__ int cnt[4];
  const int img = blockIdx.x, t = threadIdx.x;
  if (t < 4) cnt[t] = 0;
  __syncthreads();
  // the boxes of one image are contiguous (np.argwhere order); every thread finds the range by a scan (n is a few
  // thousand at most)
  int g0 = n_gt, g1 = 0, p0 = n_pr, p1 = 0;
  for (int i = 0; i < n_gt; ++i) if (gt_idx[i] == img) { if (i < g0) g0 = i; g1 = i + 1; }
  for (int i = 0; i < n_pr; ++i) if (pr_idx[i] == img) { if (i < p0) p0 = i; p1 = i + 1; }
  const int n1 = g1 > g0 ? g1 - g0 : 0, n2 = p1 > p0 ? p1 - p0 : 0;
  if (n1 > max_per_image || n2 > max_per_image) { if (t == 0) atomicAdd(out + 3, 1 << 20); return; }
  for (int i = t; i < n1 * 4; i += 256) gxy[i] = gt_xy[(size_t)g0 * 4 + i];
  for (int i = t; i < n2 * 4; i += 256) pxy[i] = pr_xy[(size_t)p0 * 4 + i];
  for (int i = t; i < 2 * max_per_image; i += 256) flags[i] = 0;
  __syncthreads();
  int bad = 0;
  for (int i = t; i < n1; i += 256) bad += (gxy[4 * i] > gxy[4 * i + 2]) || (gxy[4 * i + 1] > gxy[4 * i + 3]);
  for (int j = t; j < n2; j += 256) bad += (pxy[4 * j] > pxy[4 * j + 2]) || (pxy[4 * j + 1] > pxy[4 * j + 3]);
  if (bad) atomicAdd(&cnt[3], bad);
  for (int pair = t; pair < n1 * n2; pair += 256) {
    const int i = pair / n2, j = pair - i * n2;
    const double x1t = gxy[4 * i], y1t = gxy[4 * i + 1], x2t = gxy[4 * i + 2], y2t = gxy[4 * i + 3];
    const double x1p = pxy[4 * j], y1p = pxy[4 * j + 1], x2p = pxy[4 * j + 2], y2p = pxy[4 * j + 3];
    if (x2t < x1p || x2p < x1t || y2t < y1p || y2p < y1t) continue;
    const double inter = (fmin(x2t, x2p) - fmax(x1t, x1p)) * (fmin(y2t, y2p) - fmax(y1t, y1p));
    const double iou = inter / ((x2t - x1t) * (y2t - y1t) + (x2p - x1p) * (y2p - y1p) - inter);
    if (iou > iou_th) { flags[i] = 1; flags[max_per_image + j] = 1; }
  }
  __syncthreads();
  int gh = 0, ph = 0;
  for (int i = t; i < n1; i += 256) gh += flags[i];
  for (int j = t; j < n2; j += 256) ph += flags[max_per_image + j];
  if (gh) atomicAdd(&cnt[0], gh);
  if (ph) atomicAdd(&cnt[1], ph);
  __syncthreads();
  if (t == 0) {
    atomicAdd(out + 0, cnt[0]);                 // TP = ground-truth boxes hit
    atomicAdd(out + 1, n2 - cnt[1]);            // FP = predictions that hit nothing
    atomicAdd(out + 2, n1 - cnt[0]);            // FN
    if (cnt[3]) atomicAdd(out + 3, cnt[3]);
  }
}

// generic 4-D permute: out[b][i1][i2][i3] (contiguous) = in[b*sb + i1*s1 + i2*s2 + i3*s3]
__global__ void permute4_kernel(const float* __restrict__ in, float* __restrict__ out, long long n, int d1, int d2, int d3,
                                long long sb, long long s1, long long s2, long long s3, int scatter) {
  CY_GRID_STRIDE(i, n) {
    long long r = i;
    const int i3 = (int)(r % d3); r /= d3;
    const int i2 = (int)(r % d2); r /= d2;
    const int i1 = (int)(r % d1); r /= d1;
    const long long j = r * sb + i1 * s1 + i2 * s2 + i3 * s3;
    if (scatter) out[j] = in[i]; else out[i] = in[j];
  }
}

// NHWC 2x2 max pooling; idx keeps the winning position (0..3) for the backward
__global__ void maxpool2_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, unsigned char* __restrict__ idx,
                                    long long n, int Ho, int Wo, int C) {
  CY_GRID_STRIDE(i, n) {
    long long r = i;
    const int c = (int)(r % C); r /= C;
    const int ox = (int)(r % Wo); r /= Wo;
    const int oy = (int)(r % Ho); r /= Ho;
    const long long base = ((r * (2 * Ho) + 2 * oy) * (2 * Wo) + 2 * ox) * C + c;
    const long long rs = (long long)2 * Wo * C;
    float best = x[base]; int bi = 0;
    float v = x[base + C]; if (v > best) { best = v; bi = 1; }
    v = x[base + rs]; if (v > best) { best = v; bi = 2; }
    v = x[base + rs + C]; if (v > best) { best = v; bi = 3; }
    y[i] = best; idx[i] = (unsigned char)bi;
  }
}
__global__ void maxpool2_bwd_kernel(const float* __restrict__ dy, const unsigned char* __restrict__ idx,
                                    float* __restrict__ dx, long long n, int Ho, int Wo, int C) {
  CY_GRID_STRIDE(i, n) {
    long long r = i;
    const int c = (int)(r % C); r /= C;
    const int ox = (int)(r % Wo); r /= Wo;
    const int oy = (int)(r % Ho); r /= Ho;
    const long long base = ((r * (2 * Ho) + 2 * oy) * (2 * Wo) + 2 * ox) * C + c;
    const long long rs = (long long)2 * Wo * C;
    const int bi = idx[i];
    const float g = dy[i];
    dx[base] = bi == 0 ? g : 0.f;
    dx[base + C] = bi == 1 ? g : 0.f;
    dx[base + rs] = bi == 2 ? g : 0.f;
    dx[base + rs + C] = bi == 3 ? g : 0.f;
  }
}

// the same with 4 channels per thread (C % 4 == 0, 16-byte aligned tensors): DarkNet's first block is pooled at 416 x 416 x 32 -- the
// one-element-per-thread kernels above ran its 88 M elements at 97 / 86 us
__global__ void maxpool2_fwd4_kernel(const float* __restrict__ x, float* __restrict__ y, unsigned char* __restrict__ idx,
                                     long long n4, int Ho, int Wo, int C) {
  const int c4n = C >> 2;
  CY_GRID_STRIDE(i, n4) {
    long long r = i;
    const int c = (int)(r % c4n) * 4; r /= c4n;
    const int ox = (int)(r % Wo); r /= Wo;
    const int oy = (int)(r % Ho); r /= Ho;
    const long long base = ((r * (2 * Ho) + 2 * oy) * (2 * Wo) + 2 * ox) * C + c;
    const long long rs = (long long)2 * Wo * C;
    const float4 a = *(const float4*)(x + base), b = *(const float4*)(x + base + C), c_ = *(const float4*)(x + base + rs),
                 d = *(const float4*)(x + base + rs + C);
    float4 best = a;
    uchar4 bi = {0, 0, 0, 0};
#define CY_MP(f) { if (b.f > best.f) { best.f = b.f; bi.f = 1; } if (c_.f > best.f) { best.f = c_.f; bi.f = 2; } if (d.f > best.f) { best.f = d.f; bi.f = 3; } }
    CY_MP(x) CY_MP(y) CY_MP(z) CY_MP(w)
#undef CY_MP
    *(float4*)(y + i * 4) = best;
    *(uchar4*)(idx + i * 4) = bi;
  }
}
__global__ void maxpool2_bwd4_kernel(const float* __restrict__ dy, const unsigned char* __restrict__ idx, float* __restrict__ dx,
                                     long long n4, int Ho, int Wo, int C) {
  const int c4n = C >> 2;
  CY_GRID_STRIDE(i, n4) {
    long long r = i;
    const int c = (int)(r % c4n) * 4; r /= c4n;
    const int ox = (int)(r % Wo); r /= Wo;
    const int oy = (int)(r % Ho); r /= Ho;
    const long long base = ((r * (2 * Ho) + 2 * oy) * (2 * Wo) + 2 * ox) * C + c;
    const long long rs = (long long)2 * Wo * C;
    const float4 g = *(const float4*)(dy + i * 4);
    const uchar4 bi = *(const uchar4*)(idx + i * 4);
    float4 d0, d1, d2, d3;
#define CY_MB(f) { d0.f = bi.f == 0 ? g.f : 0.f; d1.f = bi.f == 1 ? g.f : 0.f; d2.f = bi.f == 2 ? g.f : 0.f; d3.f = bi.f == 3 ? g.f : 0.f; }
    CY_MB(x) CY_MB(y) CY_MB(z) CY_MB(w)
#undef CY_MB
    *(float4*)(dx + base) = d0; *(float4*)(dx + base + C) = d1; *(float4*)(dx + base + rs) = d2; *(float4*)(dx + base + rs + C) = d3;
  }
}

// nearest-neighbour upsample by an integer factor f, NHWC
__global__ void upsample_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, long long n, int Hi, int Wi, int C,
                                    int f) {
  CY_GRID_STRIDE(i, n) {
    long long r = i;
    const int c = (int)(r % C); r /= C;
    const int ox = (int)(r % (Wi * f)); r /= (Wi * f);
    const int oy = (int)(r % (Hi * f)); r /= (Hi * f);
    y[i] = x[((r * Hi + oy / f) * Wi + ox / f) * C + c];
  }
}
__global__ void upsample_bwd_kernel(const float* __restrict__ dy, float* __restrict__ dx, long long n, int Hi, int Wi, int C,
                                    int f) {
  CY_GRID_STRIDE(i, n) {            // i over the INPUT elements
    long long r = i;
    const int c = (int)(r % C); r /= C;
    const int ix = (int)(r % Wi); r /= Wi;
    const int iy = (int)(r % Hi); r /= Hi;
    float s = 0.f;
    for (int a = 0; a < f; ++a)
      for (int b = 0; b < f; ++b) s += dy[((r * Hi * f + iy * f + a) * (long long)(Wi * f) + ix * f + b) * C + c];
    dx[i] = s;
  }
}

__global__ void tanh_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, long long n) {
  CY_GRID_STRIDE(i, n) y[i] = tanhf(x[i]);
}
__global__ void tanh_bwd_kernel(const float* __restrict__ y, const float* __restrict__ dy, float* __restrict__ dx, long long n) {
  CY_GRID_STRIDE(i, n) dx[i] = dy[i] * (1.f - y[i] * y[i]);
}

// YOLO head (models.py:226-236): sigmoid on the first `split` channels of each cell, softmax on the rest
__global__ void yolo_head_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, long long cells, int split, int C) {
  CY_GRID_STRIDE(cell, cells) {
    const float* xi = x + cell * (split + C);
    float* yi = y + cell * (split + C);
    for (int k = 0; k < split; ++k) yi[k] = 1.f / (1.f + expf(-xi[k]));
    if (C > 0) {
      float m = -INFINITY, s = 0.f;
      for (int k = 0; k < C; ++k) m = fmaxf(m, xi[split + k]);
      for (int k = 0; k < C; ++k) s += expf(xi[split + k] - m);
      for (int k = 0; k < C; ++k) yi[split + k] = expf(xi[split + k] - m) / s;
    }
  }
}
__global__ void yolo_head_bwd_kernel(const float* __restrict__ y, const float* __restrict__ dy, float* __restrict__ dx,
                                     long long cells, int split, int C) {
  CY_GRID_STRIDE(cell, cells) {
    const float* yi = y + cell * (split + C);
    const float* gi = dy + cell * (split + C);
    float* di = dx + cell * (split + C);
    for (int k = 0; k < split; ++k) di[k] = gi[k] * yi[k] * (1.f - yi[k]);
    if (C > 0) {
      float dot = 0.f;
      for (int k = 0; k < C; ++k) dot += gi[split + k] * yi[split + k];
      for (int k = 0; k < C; ++k) di[split + k] = yi[split + k] * (gi[split + k] - dot);
    }
  }
}

// out[b][:] = caps[b][y[b]][:]  /  scatter of its gradient (zero elsewhere)
__global__ void pick_capsule_kernel(const float* __restrict__ caps, const long long* __restrict__ y, float* __restrict__ out,
                                    int B, int C, int D, int backward) {
  const int n = backward ? B * C * D : B * D;
  CY_GRID_STRIDE(i, n) {
    if (!backward) {
      const int b = (int)(i / D), d = (int)(i % D);
      out[i] = caps[((long long)b * C + y[b]) * D + d];
    } else {       // caps = d(out) [B][D], out = d(caps) [B][C][D]
      const int d = (int)(i % D), c = (int)((i / D) % C), b = (int)(i / ((long long)C * D));
      out[i] = (y[b] == c) ? caps[(long long)b * D + d] : 0.f;
    }
  }
}

}  // namespace

#define CY_S ((hipStream_t)stream)

extern "C" int cy_center_u8(const unsigned char* src, float* dst, int B, int H, int W, int C, int to_nchw, void* stream) {
  CY_REQUIRE(src && dst && B > 0 && H > 0 && W > 0 && C > 0, "cy_center_u8: bad arguments");
  const long long npix = (long long)B * H * W;
  center_u8_kernel<<<grid_for(npix * C), 256, 0, CY_S>>>(src, dst, npix, C, (long long)H * W, to_nchw);
  CY_LAUNCH_CHECK("cy_center_u8");
  return 0;
}
extern "C" int cy_yolo_decode_boxes(const float* y, const long long* image_hw, double img_h, double img_w, int B, int g, int nb,
                                    int C, float conf_th, int* count, int* image_idx, double* xy, int* cls, int max_boxes,
                                    void* stream) {
  CY_REQUIRE(y && count && image_idx && xy && B > 0 && g > 0 && nb > 0 && C >= 0 && max_boxes > 0, "cy_yolo_decode_boxes: bad arguments");
  CY_REQUIRE(C == 0 || cls, "cy_yolo_decode_boxes: cls must be given when C > 0");
  yolo_decode_kernel<<<1, 1024, 0, CY_S>>>(y, image_hw, img_h, img_w, B, g, nb, C, conf_th, count, image_idx, xy, cls, max_boxes);
  CY_LAUNCH_CHECK("cy_yolo_decode_boxes");
  return 0;
}
extern "C" int cy_detect_confusion(const int* gt_idx, const double* gt_xy, int n_gt, const int* pr_idx, const double* pr_xy, int n_pr,
                                   int n_images, double iou_th, int max_per_image, int* out4, void* stream) {
  CY_REQUIRE(out4 && n_images > 0 && n_gt >= 0 && n_pr >= 0 && max_per_image > 0, "cy_detect_confusion: bad arguments");
  CY_REQUIRE((n_gt == 0 || (gt_idx && gt_xy)) && (n_pr == 0 || (pr_idx && pr_xy)), "cy_detect_confusion: null box arrays");
  const size_t lds = (size_t)max_per_image * (8 * 8 + 2 * 4);
  CY_REQUIRE(lds <= 60 * 1024, "cy_detect_confusion: max_per_image=%d too large", max_per_image);
  detect_confusion_kernel<<<n_images, 256, lds, CY_S>>>(gt_idx, gt_xy, n_gt, pr_idx, pr_xy, n_pr, iou_th, max_per_image, out4);
  CY_LAUNCH_CHECK("cy_detect_confusion");
  return 0;
}
extern "C" int cy_permute4(const float* in, float* out, long long nb, int d1, int d2, int d3, long long sb, long long s1,
                           long long s2, long long s3, int scatter, void* stream) {
  CY_REQUIRE(in && out && nb > 0 && d1 > 0 && d2 > 0 && d3 > 0, "cy_permute4: bad arguments");
  const long long n = nb * d1 * d2 * d3;
  permute4_kernel<<<grid_for(n), 256, 0, CY_S>>>(in, out, n, d1, d2, d3, sb, s1, s2, s3, scatter);
  CY_LAUNCH_CHECK("cy_permute4");
  return 0;
}
extern "C" int cy_maxpool2_fwd(const float* x, float* y, unsigned char* idx, int B, int Ho, int Wo, int C, void* stream) {
  CY_REQUIRE(x && y && idx && B > 0 && Ho > 0 && Wo > 0 && C > 0, "cy_maxpool2_fwd: bad arguments");
  const long long n = (long long)B * Ho * Wo * C;
  if (C % 4 == 0 && (((uintptr_t)x | (uintptr_t)y | (uintptr_t)idx) & 15) == 0)
    maxpool2_fwd4_kernel<<<grid_for(n / 4), 256, 0, CY_S>>>(x, y, idx, n / 4, Ho, Wo, C);
  else
    maxpool2_fwd_kernel<<<grid_for(n), 256, 0, CY_S>>>(x, y, idx, n, Ho, Wo, C);
  CY_LAUNCH_CHECK("cy_maxpool2_fwd");
  return 0;
}
extern "C" int cy_maxpool2_bwd(const float* dy, const unsigned char* idx, float* dx, int B, int Ho, int Wo, int C, void* stream) {
  CY_REQUIRE(dy && dx && idx && B > 0 && Ho > 0 && Wo > 0 && C > 0, "cy_maxpool2_bwd: bad arguments");
  const long long n = (long long)B * Ho * Wo * C;
  if (C % 4 == 0 && (((uintptr_t)dy | (uintptr_t)dx | (uintptr_t)idx) & 15) == 0)
    maxpool2_bwd4_kernel<<<grid_for(n / 4), 256, 0, CY_S>>>(dy, idx, dx, n / 4, Ho, Wo, C);
  else
    maxpool2_bwd_kernel<<<grid_for(n), 256, 0, CY_S>>>(dy, idx, dx, n, Ho, Wo, C);
  CY_LAUNCH_CHECK("cy_maxpool2_bwd");
  return 0;
}
extern "C" int cy_upsample_fwd(const float* x, float* y, int B, int Hi, int Wi, int C, int f, void* stream) {
  CY_REQUIRE(x && y && B > 0 && Hi > 0 && Wi > 0 && C > 0 && f >= 1, "cy_upsample_fwd: bad arguments");
  const long long n = (long long)B * Hi * f * Wi * f * C;
  upsample_fwd_kernel<<<grid_for(n), 256, 0, CY_S>>>(x, y, n, Hi, Wi, C, f);
  CY_LAUNCH_CHECK("cy_upsample_fwd");
  return 0;
}
extern "C" int cy_upsample_bwd(const float* dy, float* dx, int B, int Hi, int Wi, int C, int f, void* stream) {
  CY_REQUIRE(dy && dx && B > 0 && Hi > 0 && Wi > 0 && C > 0 && f >= 1, "cy_upsample_bwd: bad arguments");
  const long long n = (long long)B * Hi * Wi * C;
  upsample_bwd_kernel<<<grid_for(n), 256, 0, CY_S>>>(dy, dx, n, Hi, Wi, C, f);
  CY_LAUNCH_CHECK("cy_upsample_bwd");
  return 0;
}
extern "C" int cy_tanh_fwd(const float* x, float* y, long long n, void* stream) {
  CY_REQUIRE(x && y && n > 0, "cy_tanh_fwd: bad arguments");
  tanh_fwd_kernel<<<grid_for(n), 256, 0, CY_S>>>(x, y, n);
  CY_LAUNCH_CHECK("cy_tanh_fwd");
  return 0;
}
extern "C" int cy_tanh_bwd(const float* y, const float* dy, float* dx, long long n, void* stream) {
  CY_REQUIRE(y && dy && dx && n > 0, "cy_tanh_bwd: bad arguments");
  tanh_bwd_kernel<<<grid_for(n), 256, 0, CY_S>>>(y, dy, dx, n);
  CY_LAUNCH_CHECK("cy_tanh_bwd");
  return 0;
}
extern "C" int cy_yolo_head_fwd(const float* x, float* y, long long cells, int split, int C, void* stream) {
  CY_REQUIRE(x && y && cells > 0 && split >= 0 && C >= 0 && split + C > 0, "cy_yolo_head_fwd: bad arguments");
  yolo_head_fwd_kernel<<<grid_for(cells), 256, 0, CY_S>>>(x, y, cells, split, C);
  CY_LAUNCH_CHECK("cy_yolo_head_fwd");
  return 0;
}
extern "C" int cy_yolo_head_bwd(const float* y, const float* dy, float* dx, long long cells, int split, int C, void* stream) {
  CY_REQUIRE(y && dy && dx && cells > 0 && split >= 0 && C >= 0 && split + C > 0, "cy_yolo_head_bwd: bad arguments");
  yolo_head_bwd_kernel<<<grid_for(cells), 256, 0, CY_S>>>(y, dy, dx, cells, split, C);
  CY_LAUNCH_CHECK("cy_yolo_head_bwd");
  return 0;
}
extern "C" int cy_pick_capsule(const float* caps, const long long* y, float* out, int B, int C, int D, int backward, void* stream) {
  CY_REQUIRE(caps && y && out && B > 0 && C > 0 && D > 0, "cy_pick_capsule: bad arguments");
  const long long n = backward ? (long long)B * C * D : (long long)B * D;
  pick_capsule_kernel<<<grid_for(n), 256, 0, CY_S>>>(caps, y, out, B, C, D, backward);
  CY_LAUNCH_CHECK("cy_pick_capsule");
  return 0;
}

/* zero-fill of the per-step scratch arena (BatchNorm statistics / backward sums are accumulated with atomics into
 * memory that must start at zero): ONE launch per training step instead of one fill per buffer */
extern "C" int cy_zero_bytes(void* p, long long nbytes, void* stream) {
  CY_REQUIRE(p && nbytes >= 0, "cy_zero_bytes: bad arguments");
  if (nbytes == 0) return 0;
  hipError_t e = hipMemsetAsync(p, 0, (size_t)nbytes, (hipStream_t)stream);
  if (e != hipSuccess) return cy_set_error((int)e, "cy_zero_bytes: %s", hipGetErrorString(e));
  return 0;
}
