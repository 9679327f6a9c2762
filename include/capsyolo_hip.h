/* capsyolo_hip.h -- C-ABI of libcapsyolo_hip.so (hand-written gfx950 kernels).
 *
 * Drop-in boundary for the training hot path of
 * Cranial-XIX/cs231-capsule-yolo-traffic-sign-detection.  The reference has no FFI
 * (all arithmetic is torch ATen calls); each entry point below replaces the ATen
 * work issued by the cited reference lines.  Conventions for EVERY function:
 *   - plain C types only; every pointer is a DEVICE pointer owned by the caller
 *     (the PyTorch caching allocator), valid on `stream`; nothing is allocated,
 *     freed or retained by the library;
 *   - `stream` is a hipStream_t passed as void*; launches are asynchronous;
 *   - re-entrant, no mutable global state (forward is called from the Python
 *     main thread, backward from the autograd engine thread);
 *   - returns 0 on success, otherwise a non-zero code (hipError_t for runtime
 *     errors, CY_EINVAL for rejected arguments); capsyolo_last_error() gives
 *     the text for the calling thread.
 * Activations are NHWC fp32 unless a stride set says otherwise.
 */
#ifndef CAPSYOLO_HIP_H
#define CAPSYOLO_HIP_H
#ifdef __cplusplus
extern "C" {
#endif

#define CY_EINVAL 10001

const char* capsyolo_last_error(void);
int capsyolo_abi_version(void);

/* ------------------------------------------------------------------ convolution as implicit GEMM
 * Replaces nn.Conv2d forward / input-gradient / weight-gradient:
 * models.py:90 (conv1), 60-62 (primary capsule convs), 98-110 (decoder convs),
 * 132-223 (DarkNet), 347-363 (DarkCapsuleNet backbone).
 *
 * One description drives forward and input-gradient: a "virtual output grid"
 * [B,Ho,Wo] of pixels, each the sum over TH x TW taps of a Cin-vector of X times
 * a [Cin x N] weight slice:
 *   iy = oy*in_stride + dy0 + a*dstep,  ix = ox*in_stride + dx0 + b*dstep   (zero outside X)
 *   Y[b, oy*out_stride+out_oy, ox*out_stride+out_ox, n] = act(sum + bias[n])
 * X is addressed through element strides (xs_*), so NCHW inputs work (Cin%32 != 0 path).
 */
typedef struct {
  const float* X; const float* Wp; float* Y; const float* bias; double* stats;
  long long xs_b, xs_y, xs_x, xs_c;
  int B, Hi, Wi, Cin;
  int Ho, Wo, N;
  int TH, TW, in_stride, dy0, dx0, dstep;
  int Hy, Wy, out_stride, out_oy, out_ox;
  int act;            /* 0 none, 1 ReLU, 2 LeakyReLU with slope act_slope (last field) */
  /* Optional (input-gradient launches): Y is the gradient with respect to lrelu(bn_z * bn_scale + bn_shift), the
   * activated output of the PRODUCER block whose raw convolution output bn_z has Y's layout.  The epilogue then also
   * accumulates that BatchNorm's backward sums over the pixels it stores -- per channel n:
   *   d = Y * (bn_z*scale+shift > 0 ? 1 : bn_slope),  bn_red[copy][n][0] += d,  bn_red[copy][n][1] += d * (bn_z - mean) * invstd
   * (bn_red[CY_STATS_COPIES][N][2] doubles, zeroed by the caller) -- which replaces the cy_bn_bwd_reduce pass
   * (loss_fns / autograd of models.py:349-351).  All NULL / 0 when unused.
   * cy_conv_gemm_bf16 (bf16 output only): bn_z points to bf16 values, and the kernel STORES d (the premasked gradient, rounded to
   * bf16) and sums the stored values: the producer's cy_bn_bwd_apply_bf16 then runs with slope 1. */
  const float* bn_z; const float* bn_scale; const float* bn_shift; const float* bn_mean; const float* bn_invstd;
  double* bn_red;
  float bn_slope;
  float act_slope;    /* act == 2: Y = max(v, act_slope * v), 0 <= act_slope <= 1 -- the eval-mode forward of a conv -> BatchNorm ->
                       * LeakyReLU block whose BatchNorm was folded into weights and bias (cy_bn_fold_eval) */
  /* Optional workspace of cy_conv_gemm (fp32): ws_floats >= cy_conv_gemm_ws_floats(a) floats, 16-byte aligned.  A layer whose
   * output grid is far below the chip (CapsuleNet's primary-capsule convolution, models.py:60-62: 42 tiles, K = 20736) then runs
   * several blocks per tile on shares of the reduction; the shares' partial sums are added in a fixed order (deterministic).
   * NULL / 0: never split. */
  float* ws; long long ws_floats;
} cy_conv_gemm_t;

/* First layer of the backbones (models.py:347-349 DarkCapsuleNet conv_1 3 -> 128, models.py:132-136 DarkNet conv_1
 * 3 -> 32): 3x3 / stride 1 / pad 1 on the NCHW image X[B][3][H][W] (W % 32 == 0) with PyTorch-layout weights
 * W[Cout][3][3][3], Cout in {32, 64, 128}; Y[B][H][W][Cout] NHWC.  bias / stats as in cy_conv_gemm (either may be NULL).
 * Y may be NULL when stats is given (statistics only: the block's backward recomputes z, cy_conv1_bn_bwd_*).
 * Optional scale / shift [Cout] and slope (then stats must be NULL): Y = lrelu((conv + bias) * scale + shift), the second
 * pass of a conv -> BatchNorm -> LeakyReLU block -- recomputing this layer is cheaper than reading its output back.
 * A store-bound layer: persistent waves, operands from registers / L2, no LDS. */
int cy_conv1_3x3_fwd(const float* X, const float* W, const float* bias, float* Y, double* stats, const float* scale,
                     const float* shift, float slope, int B, int H, int Wd, int Cout, void* stream);
/* The statistics of that layer's output WITHOUT computing it: stats[co] += (sum z, sum z^2) over all pixels from the 28 x 28
 * moment matrix of the 27-element input patches (one Gram-matrix pass over the image, MFMA with both operands the same
 * register), the weights and the bias -- replaces the statistics-only call of cy_conv1_3x3_fwd for any Cout.  W % 32 == 0;
 * ws: cy_conv1_3x3_stats_ws_floats(B, H) floats (partial matrices; 8-byte aligned). */
long long cy_conv1_3x3_stats_ws_floats(int B, int H);
long long cy_conv1_3x3_stats_m2_offset(int B, int H);   /* float offset of the double M2[32][32] inside ws after the call */
int cy_conv1_3x3_stats(const float* X, const float* W, const float* bias, double* stats, float* ws, int B, int H, int Wd,
                       int Cout, void* stream);
/* ... and its weight gradient dW[Cout][3][3][3] from X and dZ[B][H][W][Cout] (the weight-gradient half of nn.Conv2d
 * backward for that layer); ws: cy_conv1_3x3_wgrad_ws_floats floats (one partial slab per persistent wave, added up in a
 * fixed order: deterministic). */
long long cy_conv1_3x3_wgrad_ws_floats(int B, int H, int Wd, int Cout);
int cy_conv1_3x3_wgrad(const float* X, const float* dZ, float* dW, float* ws, int B, int H, int Wd, int Cout, void* stream);
/* Backward of that layer's whole conv -> BatchNorm -> LeakyReLU block from dA, the gradient with respect to the activation
 * (nn.Conv2d / nn.BatchNorm2d / nn.LeakyReLU backward, models.py:347-348), without z and dz in memory: both passes
 * recompute z = conv(X) + bias tile by tile.  Pass 1: red[CY_STATS_COPIES][Cout][2] (zeroed by the caller) += (sum d,
 * sum d xhat) like cy_bn_bwd_reduce.  Pass 2: with red[Cout][2] summed over the copies (and ranks) and count = pixels per
 * channel of the (global) batch, dW[Cout][3][3][3] = weight gradient of dz = scale (d - red0/count - xhat red1/count);
 * dgamma = red1, dbeta = red0 as in cy_bn_bwd_apply.  ws: cy_conv1_bn_bwd_wgrad_ws_floats floats. */
int cy_conv1_bn_bwd_reduce(const float* X, const float* W, const float* bias, const float* dA, const float* scale,
                           const float* shift, const float* mean, const float* invstd, float slope, double* red, int B, int H,
                           int Wd, int Cout, void* stream);
long long cy_conv1_bn_bwd_wgrad_ws_floats(int B, int H, int Wd, int Cout);
int cy_conv1_bn_bwd_wgrad(const float* X, const float* W, const float* bias, const float* dA, const float* scale,
                          const float* shift, const float* mean, const float* invstd, float slope, const double* red,
                          long long count, float* dW, float* ws, int B, int H, int Wd, int Cout, void* stream);
/* The same three passes at the boundary to the bf16 path ("precision": "bf16", BASELINE configs[4]; models.py:347-351): the
 * first block's activation leaves as bf16 NHWC (Y) and its backward takes the second block's bf16 input gradient (dA) --
 * no fp32 copy of the activation or of its gradient exists.  Everything else as in the fp32 entry points above. */
int cy_conv1_3x3_fwd_act_bf16(const float* X, const float* W, const float* bias, void* Y, const float* scale,
                              const float* shift, float slope, int B, int H, int Wd, int Cout, void* stream);
int cy_conv1_bn_bwd_reduce_bf16(const float* X, const float* W, const float* bias, const void* dA, const float* scale,
                                const float* shift, const float* mean, const float* invstd, float slope, double* red, int B,
                                int H, int Wd, int Cout, void* stream);
int cy_conv1_bn_bwd_wgrad_bf16(const float* X, const float* W, const float* bias, const void* dA, const float* scale,
                               const float* shift, const float* mean, const float* invstd, float slope, const double* red,
                               long long count, float* dW, float* ws, int B, int H, int Wd, int Cout, void* stream);
/* The same backward in ONE pass over dA (instead of cy_conv1_bn_bwd_reduce + cy_conv1_bn_bwd_wgrad) for a block whose forward
 * statistics came from cy_conv1_3x3_stats: the pass sums d = dA * lrelu'(y) and the weight gradient OF d; since the layer is
 * linear in its input patches, sum d*xhat and the weight gradient of dz follow from those and the forward's patch moments M2
 * (ws + cy_conv1_3x3_stats_m2_offset of that call) in closed form.  redc: [CY_STATS_COPIES][Cout][2] doubles, zeroed by the
 * caller; dW[Cout][27], dgamma, dbeta [Cout] written; red_out (optional) [Cout][2] = (sum d, sum d*xhat);
 * ws: cy_conv1_bn_bwd_wgrad_ws_floats floats. */
int cy_conv1_bn_bwd_onepass(const float* X, const float* W, const float* bias, const float* dA, const float* scale,
                            const float* shift, const float* mean, const float* invstd, float slope, const double* M2,
                            double* redc, float* dW, float* dgamma, float* dbeta, double* red_out, float* ws, int B, int H,
                            int Wd, int Cout, void* stream);
int cy_conv1_bn_bwd_onepass_bf16(const float* X, const float* W, const float* bias, const void* dA, const float* scale,
                                 const float* shift, const float* mean, const float* invstd, float slope, const double* M2,
                                 double* redc, float* dW, float* dgamma, float* dbeta, double* red_out, float* ws, int B, int H,
                                 int Wd, int Cout, void* stream);

/* number of floats of a packed-weight buffer for (K = TH*TW*Cin, N) */
long long cy_conv_packed_floats(int K, int N);
/* pack PyTorch-layout weights W[Cout][Cin][KH][KW] for the forward GEMM of a (sub)set of taps:
 * tap (a,b) uses kernel element (kh0 + a*kstep, kw0 + b*kstep);  transpose=1 packs the
 * input-gradient operand (rows = (tap, cout), columns = cin). */
int cy_conv_pack_weights(const float* W, float* Wp, int Cout, int Cin, int KH, int KW,
                         int TH, int TW, int kh0, int kw0, int kstep, int transpose, void* stream);
/* Y = conv(X) (+bias, +act); if stats != NULL also accumulates per-channel sum / sum-of-squares of the
 * pre-activation output into stats[CY_STATS_COPIES][N][2] (double), which the caller zeroed: every block adds into
 * copy (block id mod CY_STATS_COPIES), so that tens of thousands of blocks do not queue on 2 N addresses (conv_1's
 * launch was bound by exactly that); cy_bn_finalize adds the copies up. */
#define CY_STATS_COPIES 16
int cy_conv_gemm(const cy_conv_gemm_t* a, void* stream);
long long cy_conv_gemm_ws_floats(const cy_conv_gemm_t* a);   /* 0: this launch does not split its reduction */

/* weight gradient: dW[Cout][Cin][KH][KW] = sum over pixels of X-patch (x) dZ.
 * slabs: workspace of cy_conv_wgrad_ws_floats() floats.  dZ is NHWC [B,Ho,Wo,N] contiguous. */
typedef struct {
  const float* X; const float* dZ; float* dW; float* slabs;
  long long xs_b, xs_y, xs_x, xs_c;
  int B, Hi, Wi, Cin;
  int Ho, Wo, N;
  int KH, KW, stride, pad;
} cy_conv_wgrad_t;
long long cy_conv_wgrad_ws_floats(const cy_conv_wgrad_t* a);
int cy_conv_wgrad(const cy_conv_wgrad_t* a, void* stream);

/* Fused Winograd F(2x2,3x3) for 3x3 / stride 1 / pad 1 convolutions on NHWC (2.25x fewer multiplies than the
 * implicit GEMM; fp32 error ~1e-6 relative).  U = cy_wino_pack_weights(W[Cout][Cin][3][3]); transpose=1 packs
 * the input-gradient operand (then call with X = dZ, Cin = layer Cout, Cout = layer Cin).  Cin % 8 == 0.
 * bias / stats as in cy_conv_gemm (either may be NULL). */
long long cy_wino_packed_floats(int Cin, int N);
int cy_wino_pack_weights(const float* W, float* U, int Cout, int Cin, int transpose, void* stream);
int cy_conv3x3_winograd(const float* X, const float* U, float* Y, const float* bias, double* stats, float out_slope,
                        int B, int H, int W, int Cin, int Cout, void* stream);
/* The same with an optional workspace: a launch WITHOUT an epilogue (plain = no bias, no statistics, no activation: the input
 * gradients) whose tiles fill at most half of the CUs (DarkNet's 13 x 13 layers, models.py:196-223: 128 tiles) splits its
 * reduction channels over up to 4 blocks per tile; the shares' slabs are added in a fixed order.  ws: cy_wino_split_ws_floats
 * floats (0 = this shape does not split), 16-byte aligned. */
long long cy_wino_split_ws_floats(int B, int H, int W, int Cin, int Cout, int plain);
int cy_conv3x3_winograd_ws(const float* X, const float* U, float* Y, const float* bias, double* stats, float out_slope,
                           int B, int H, int W, int Cin, int Cout, float* ws, long long ws_floats, void* stream);
/* The same convolution through Winograd F(4x4,3x3) (36 multiplies per 4x4 outputs: 1.78x fewer MFMAs than F(2x2,3x3);
 * fp32 error ~2e-6 relative; replaces the same nn.Conv2d forward / input gradient, models.py:349-351).  Same arguments
 * and meaning as the three functions above; U has its own layout (cy_wino4_packed_floats / cy_wino4_pack_weights). */
long long cy_wino4_packed_floats(int Cin, int N);
int cy_wino4_pack_weights(const float* W, float* U, int Cout, int Cin, int transpose, void* stream);
int cy_conv3x3_winograd4(const float* X, const float* U, float* Y, const float* bias, double* stats, float out_slope,
                         int B, int H, int W, int Cin, int Cout, void* stream);
/* Weight gradient of the same layers through Winograd F(3x3,4x4) (36 multiplies per 4x4 tile of dZ and 3x3 taps: 1.78x fewer MFMAs
 * than F(3x3,2x2) below; fp32 error ~6e-6 relative; replaces the weight-gradient half of nn.Conv2d backward, models.py:349-351).
 * Shapes: cy_wino4_wgrad_ok (H % 4 == 0, W % 16 == 0, Cin % 32 == 0, Cout % 64 == 0, each image below 256 MB); ws:
 * cy_wino4_wgrad_ws_floats floats.  _bn: as cy_conv3x3_winograd_wgrad_bn with a PREMASKED gradient (dA is d = dA lrelu'(y)):
 * dz = d scale + (Z - mean) kb + kc is formed on the way in (kb, kc from red / count) and written to dZ for the input-gradient
 * kernel. */
int cy_wino4_wgrad_ok(int B, int H, int W, int Cin, int Cout);
long long cy_wino4_wgrad_ws_floats(int B, int H, int W, int Cin, int Cout);
int cy_conv3x3_winograd4_wgrad(const float* X, const float* dZ, float* dW, float* ws,
                               int B, int H, int W, int Cin, int Cout, void* stream);
int cy_conv3x3_winograd4_wgrad_bn(const float* X, const float* Z, const float* dA, float* dZ, const float* scale,
                                  const float* mean, const float* invstd, const double* red, long long count,
                                  float* dW, float* ws, int B, int H, int W, int Cin, int Cout, void* stream);
/* Weight gradient of the same layers through Winograd F(3x3,2x2): dW[Cout][Cin][3][3] from X[B][H][W][Cin] and
 * dZ[B][H][W][Cout] (replaces the weight-gradient half of nn.Conv2d backward, models.py:132-223 conv_2 class of
 * layers).  Cin % 64 == 0 and Cout % 64 == 0.  ws: cy_wino_wgrad_ws_floats(B, Cin, Cout) floats (Winograd-domain
 * partial sums, one slab per contiguous range of tile groups -- the batch's B * ceil(H/4) * ceil(W/8) groups are dealt to
 * about one block per CU whatever the batch and map size -- reduced in a fixed order: the result is deterministic). */
long long cy_wino_wgrad_ws_floats(int B, int Cin, int Cout);
int cy_conv3x3_winograd_wgrad(const float* X, const float* dZ, float* dW, float* ws,
                              int B, int H, int W, int Cin, int Cout, void* stream);
/* The same weight gradient with the block's BatchNorm + LeakyReLU backward applied on the way in (replaces pass 2 of
 * nn.BatchNorm2d / nn.LeakyReLU backward, models.py:347-351, for conv_2-class layers: 17 GB of elementwise traffic at the
 * headline shape): the kernel reads the raw convolution output Z and dA, the gradient with respect to the activated
 * output, forms dz = scale * (d - mean(d) - xhat * mean(d * xhat)), d = dA * lrelu'(Z * scale + shift), in registers
 * (red = the sums of cy_bn_bwd_reduce over `count` pixels per channel), feeds it to the MFMAs and also writes it to
 * dZ[B][H][W][Cout] for the input-gradient kernel.  dZ must not alias Z or dA.  dgamma / dbeta: cy_bn_param_grad.
 * premasked != 0: dA is already d (cy_conv4x4s2_winograd_dgrad with bn_red set stores the masked gradient). */
int cy_conv3x3_winograd_wgrad_bn(const float* X, const float* Z, const float* dA, float* dZ, const float* scale,
                                 const float* shift, const float* mean, const float* invstd, float slope, int premasked,
                                 const double* red, long long count, float* dW, float* ws,
                                 int B, int H, int W, int Cin, int Cout, void* stream);

/* Fused Winograd F(2x2,2x2) forward for 4x4 / stride 2 / pad 1 convolutions on NHWC (DarkCapsuleNet conv_3..5,
 * models.py:352-363): the layer is a 2x2 stride-1 convolution of the shifted space-to-depth view of its input, which
 * needs 9 instead of 16 multiplies per 2x2 outputs.  X [B][H][W][Cin] (H, W even, Cin % 8 == 0) -> Y [B][H/2][W/2][Cout];
 * U = cy_wino2_pack_weights(W[Cout][Cin][4][4]); bias / stats as in cy_conv_gemm. */
long long cy_wino2_packed_floats(int Cin, int N);
int cy_wino2_pack_weights(const float* W, float* U, int Cout, int Cin, void* stream);
/* in_scale / in_shift [Cin] (both or neither) + in_slope in (0, 1]: the layer's input is lrelu(X * in_scale[c] + in_shift[c])
 * -- the producer's BatchNorm + LeakyReLU (models.py:349-351) applied on the way into LDS, so that the activation tensor
 * of the previous layer is never written to HBM (X is then the previous layer's raw convolution output). */
/* out_slope in [0, 1] (1 = none): Y = lrelu(conv + bias) -- the eval-mode forward with the block's BatchNorm folded into U and
 * bias (cy_bn_fold_eval); then stats and in_scale must be NULL.  cy_conv3x3_winograd takes the same argument. */
int cy_conv4x4s2_winograd(const float* X, const float* U, float* Y, const float* bias, double* stats,
                          const float* in_scale, const float* in_shift, float in_slope, float out_slope,
                          int B, int H, int W, int Cin, int Cout, void* stream);
/* The same forward through Winograd F(4x4,2x2) (25 multiplies per 4x4 outputs: 1.44x fewer MFMAs than F(2x2,2x2); fp32 error ~3e-6
 * relative; same arguments and meaning as cy_conv4x4s2_winograd, own U layout).  Shapes: cy_wino4s2_ok (H, W even, Cin % 8 == 0, each
 * image below 256 MB). */
int cy_wino4s2_ok(int B, int H, int W, int Cin, int Cout);
long long cy_wino4s2_packed_floats(int Cin, int N);
int cy_wino4s2_pack_weights(const float* W, float* U, int Cout, int Cin, void* stream);
int cy_conv4x4s2_winograd4(const float* X, const float* U, float* Y, const float* bias, double* stats,
                           const float* in_scale, const float* in_shift, float in_slope, float out_slope,
                           int B, int H, int W, int Cin, int Cout, void* stream);
/* Input gradient of the same layers through F(4x4,2x2): the arguments of cy_conv4x4s2_winograd_dgrad below (U =
 * cy_wino4s2_pack_dgrad_weights(W[Cout][Cin][4][4]), cy_wino4s2_dgrad_packed_floats(Cin, Cout) floats; bn_* as there: with bn_red
 * the kernel stores dX * lrelu'(z * scale + shift) and adds the producer BatchNorm's backward sums).  H, W, Cin, Cout are the
 * LAYER's (dX is [B][H][W][Cin]); shapes: cy_wino4s2_dgrad_ok (H, W even, Cin % 64 == 0, Cout % 8 == 0). */
int cy_wino4s2_dgrad_ok(int B, int H, int W, int Cin, int Cout);
long long cy_wino4s2_dgrad_packed_floats(int Cin, int Cout);
int cy_wino4s2_pack_dgrad_weights(const float* W, float* U, int Cout, int Cin, void* stream);
int cy_conv4x4s2_winograd4_dgrad(const float* dZ, const float* U, float* dX, const float* bn_z, const float* bn_scale,
                                 const float* bn_shift, const float* bn_mean, const float* bn_invstd, float bn_slope,
                                 double* bn_red, int B, int H, int W, int Cin, int Cout, void* stream);
/* Weight gradient of the same layers through F(2x2,2x2): dW[Cout][Cin][4][4] from X[B][H][W][Cin] and
 * dZ[B][H/2][W/2][Cout].  Cin % 32 == 0, Cout % 64 == 0, H and W even.  ws: cy_wino2_wgrad_ws_floats(B, Cin, Cout) floats
 * (per-image partial sums in the Winograd domain, reduced in a fixed order: deterministic). */
long long cy_wino2_wgrad_ws_floats(int B, int Cin, int Cout);
int cy_conv4x4s2_winograd_wgrad(const float* X, const float* dZ, float* dW, float* ws,
                                const float* in_scale, const float* in_shift, float in_slope,
                                int B, int H, int W, int Cin, int Cout, void* stream);

/* Input gradient of the same layers through F(2x2,2x2): dX[B][H][W][Cin] from dZ[B][H/2][W/2][Cout]
 * (U = cy_wino2_pack_dgrad_weights(W[Cout][Cin][4][4])).  Cin % 64 == 0, Cout % 8 == 0, H and W even.  bn_* as in
 * cy_conv_gemm_t (optional BatchNorm-backward sums of the producer block; bn_red[CY_STATS_COPIES][Cin][2]).  With bn_red set
 * the kernel stores d = dX * lrelu'(bn_z * bn_scale + bn_shift) -- the gradient at the producer's BatchNorm output, which it
 * has in registers for the sums -- instead of dX: the producer's backward then applies no activation mask (slope 1). */
long long cy_wino2_dgrad_packed_floats(int Cin, int Cout);
int cy_wino2_pack_dgrad_weights(const float* W, float* U, int Cout, int Cin, void* stream);
int cy_conv4x4s2_winograd_dgrad(const float* dZ, const float* U, float* dX, const float* bn_z, const float* bn_scale,
                                const float* bn_shift, const float* bn_mean, const float* bn_invstd, float bn_slope,
                                double* bn_red, int B, int H, int W, int Cin, int Cout, void* stream);

/* per-channel sum over pixels: out[N] = sum_p dZ[p][n]  (bias gradient of convs without BatchNorm) */
int cy_channel_sum(const float* dZ, float* out, long long P, int N, void* stream);

/* ------------------------------------------------------------------ bf16 convolution path (params.json "precision": "bf16")
 * The backbone's 3x3/s1 and 4x4/s2 layers (models.py:350-363) on v_mfma_f32_32x32x16_bf16 with fp32 accumulation:
 * activations and raw conv outputs are bf16 NHWC tensors (passed as void* / through the float* fields of cy_conv_gemm_t),
 * weights stay fp32 masters and are packed to bf16 per call, BatchNorm statistics stay double.  Cin and N multiples of 64. */
long long cy_conv_bf16_packed_elems(int K, int N);
/* bf16 image [Np][K] of the GEMM B operand (same tap / transpose semantics as cy_conv_pack_weights) */
int cy_conv_bf16_pack_weights(const float* W, void* Wp, int Cout, int Cin, int KH, int KW, int TH, int TW, int kh0, int kw0,
                              int kstep, int transpose, void* stream);
/* forward / per-parity-class input gradient; X, Wp bf16; Y bf16, or fp32 when out_f32 (the consumer is an fp32 kernel) */
int cy_conv_gemm_bf16(const cy_conv_gemm_t* a, int out_f32, void* stream);
/* a[0 .. ncls-1] (ncls <= 4): descriptors that differ only in Wp, dy0, dx0, out_oy, out_ox -- the output-parity classes of a strided
 * input gradient (models.py:352-363 backward) -- in ONE launch: the classes of a pixel tile run on neighbouring blocks at the same
 * time, so X is fetched from HBM once instead of once per class. */
int cy_conv_gemm_bf16_classes(const cy_conv_gemm_t* a, int ncls, int out_f32, void* stream);
/* weight gradient dW[Cout][Cin][KH][KW] (fp32) of a pad-1 3x3/stride-1 or 4x4/stride-2 layer from bf16 X and dZ;
 * ws: cy_conv_wgrad_bf16_ws_floats() floats of per-split partial sums, added in a fixed order (-1: unsupported shape) */
long long cy_conv_wgrad_bf16_ws_floats(int B, int Ho, int Wo, int Cin, int Cout, int KH, int stride);
int cy_conv_wgrad_bf16(const void* X, const void* dZ, float* dW, float* ws, int B, int Hi, int Wi, int Cin, int Ho, int Wo,
                       int Cout, int KH, int stride, void* stream);
/* ... with the block's BatchNorm backward (pass 2) on the way in (models.py:350-365 backward, "precision": "bf16"): D = the PREMASKED
 * gradient dA * lrelu'(y) (bf16: what the consumer's fused input-gradient epilogue stored), Z = the block's raw convolution output;
 * dz = scale (d - m1 - xhat m2) is written to dZ (bf16, for the input-gradient kernel; bit-identical to cy_bn_bwd_apply_bf16 with
 * slope 1; must not alias D or Z), dW = its weight gradient, dgamma / dbeta (optional) from the sums red[N][2].
 * ws: cy_conv_wgrad_bf16_bn_ws_floats() floats. */
long long cy_conv_wgrad_bf16_bn_ws_floats(int B, int Ho, int Wo, int Cin, int Cout, int KH, int stride);
int cy_conv_wgrad_bf16_bn(const void* X, const void* D, const void* Z, void* dZ, float* dW, float* ws, const float* scale,
                          const float* mean, const float* invstd, const double* red, float* dgamma, float* dbeta, int B, int Hi,
                          int Wi, int Cin, int Ho, int Wo, int Cout, int KH, int stride, void* stream);
/* BatchNorm apply + LeakyReLU and the two backward passes on bf16 tensors (N divides 2048); dA may be fp32 (da_f32) */
int cy_affine_act_bf16(const void* Z, void* A, const float* scale, const float* shift, float slope, long long P, int N,
                       int out_f32, void* stream);
int cy_bn_bwd_reduce_bf16(const void* Z, const void* dA, int da_f32, const float* scale, const float* shift, const float* mean,
                          const float* invstd, float slope, double* red, long long P, int N, void* stream);
int cy_bn_bwd_apply_bf16(const void* Z, const void* dA, int da_f32, void* dZ, const float* scale, const float* shift,
                         const float* mean, const float* invstd, float slope, const double* red, float* dgamma, float* dbeta,
                         long long P, int N, void* stream);
int cy_cast_f32_bf16(const float* in, void* out, long long n, void* stream);
int cy_cast_bf16_f32(const void* in, float* out, long long n, void* stream);

/* ------------------------------------------------------------------ BatchNorm2d (+LeakyReLU), NHWC
 * Replaces nn.BatchNorm2d + nn.LeakyReLU in training and eval mode (models.py:132-223, 347-365). */
/* stats[CY_STATS_COPIES][N][2] (sum, sumsq from cy_conv_gemm / cy_conv3x3_winograd) -> scale/shift (gamma*invstd, beta-mean*scale),
 * mean, invstd; updates running_mean/var with `momentum` (unbiased variance), as torch does. */
int cy_bn_finalize(const double* stats, long long count, const float* gamma, const float* beta,
                   float* running_mean, float* running_var, float momentum, float eps,
                   float* scale, float* shift, float* mean, float* invstd, int N, long long* num_batches_tracked, void* stream);
/* Folds striped double sums: red_out[c][k] = scale * sum_copies red_copies[copy][c][k] (k = 0, 1); dbeta / dgamma
 * (optional) receive the two columns as floats.  Replaces the host-side copy reduction of the BatchNorm backward. */
int cy_bn_red_fold(const double* red_copies, int copies, double scale, double* red_out, float* dgamma, float* dbeta,
                   int N, void* stream);
/* eval mode: scale/shift from the running statistics */
int cy_bn_eval_scale_shift(const float* gamma, const float* beta, const float* running_mean,
                           const float* running_var, float eps, float* scale, float* shift, int N, void* stream);
/* eval mode: the BatchNorm FOLDED into the convolution in front of it (the eval forward of predict_fns.py:38-43, 65-69 over
 * models.py:132-223, 347-365): Wf[co][...] = W[co][...] * s[co], bf[co] = (bias[co] - running_mean[co]) * s[co] + beta[co] with
 * s = gamma / sqrt(running_var + eps); W is any [Cout][per_out] weight tensor (PyTorch layout), bias may be NULL.  The block's
 * eval forward is then ONE launch: the conv kernel on (Wf, bf) with its LeakyReLU epilogue (out_slope / act = 2). */
int cy_bn_fold_eval(const float* W, const float* bias, const float* gamma, const float* beta, const float* running_mean,
                    const float* running_var, float eps, float* Wf, float* bf, int Cout, int per_out, void* stream);
/* A = lrelu(Z*scale + shift), negative slope `slope` (1.0 = identity, 0.0 = ReLU); scale/shift may be NULL */
int cy_affine_act(const float* Z, float* A, const float* scale, const float* shift, float slope,
                  long long P, int N, void* stream);
/* backward, pass 1: red[N][2] += (sum dYhat, sum dYhat*xhat), dYhat = dA * lrelu'(Z*scale+shift) */
int cy_bn_bwd_reduce(const float* Z, const float* dA, const float* scale, const float* shift,
                     const float* mean, const float* invstd, float slope, double* red,
                     long long P, int N, void* stream);
/* backward, pass 2: dZ = gamma*invstd*(dYhat - mean(dYhat) - xhat*mean(dYhat*xhat)); dgamma, dbeta written */
int cy_bn_bwd_apply(const float* Z, const float* dA, float* dZ, const float* scale, const float* shift,
                    const float* mean, const float* invstd, const float* gamma, float slope,
                    const double* red, float* dgamma, float* dbeta, long long P, int N, void* stream);
/* dgamma = red[n][1], dbeta = red[n][0] alone (for callers that fuse pass 2 into another kernel) */
int cy_bn_param_grad(const double* red, float* dgamma, float* dbeta, int N, void* stream);
/* activation-only backward: dZ = dA * act'(Z)  (ReLU convs without BN) */
int cy_act_bwd(const float* Z, const float* dA, float* dZ, float slope, long long n, void* stream);

/* ------------------------------------------------------------------ capsule routing
 * Replaces CapsuleLayer.forward, routing branch (models.py:70-79), all n_iter iterations in one
 * launch, and its autograd backward.  u rows are either contiguous [R][N*Din] (gather_g == 0) or
 * read straight from the NHWC feature map [B][4g][4g][256] through the DarkCapsuleNet "cell
 * gather" (models.py:393-398; row = k*B + b).  W is route_weights [N][C][Din][Dout].
 * v_out [R][C][Dout] in row order, or, with the cell gather, [B][g*g][C][Dout] (the layout that
 * view(g,g,B,.).permute(2,0,1,3), models.py:399, presents); dv of the backward uses the same
 * layout.  s_hist [n_iter][R][C][Dout] keeps every iteration's pre-squash sums for the backward. */
typedef struct {
  const float* u; const float* W; float* v_out; float* s_hist;
  int R, N, C, Din, Dout, n_iter;
  int gather_g, gather_B;
  float* ws;            /* workspace of cy_routing_fwd_ws_floats() floats (0 for most shapes: may then be NULL) */
} cy_routing_fwd_t;
/* Few rows (R < ~1000) cannot fill the chip row-wise: the input capsules are then split over blocks too and each
 * routing iteration becomes one phase launch + one finish launch (needs the workspace). */
long long cy_routing_fwd_ws_floats(const cy_routing_fwd_t* a);
int cy_routing_fwd(const cy_routing_fwd_t* a, void* stream);

typedef struct {
  const float* u; const float* W; const float* s_hist; const float* dv;   /* dv [R][C][Dout] */
  float* du;            /* same addressing as u (contiguous rows or NHWC feature-map gradient) */
  float* dW;            /* [N][C][Din][Dout], overwritten */
  float* ws;            /* workspace, cy_routing_bwd_ws_floats() floats */
  int R, N, C, Din, Dout, n_iter;
  int gather_g, gather_B;
} cy_routing_bwd_t;
long long cy_routing_bwd_ws_floats(const cy_routing_bwd_t* a);
int cy_routing_bwd(const cy_routing_bwd_t* a, void* stream);

/* squash over the last dim (models.py:64-67), rows of D floats; and its backward */
int cy_squash_fwd(const float* s, float* v, long long rows, int D, void* stream);
int cy_squash_bwd(const float* s, const float* dv, float* ds, long long rows, int D, void* stream);
/* capsule length (x**2).sum(-1)**0.5 (models.py:117) and backward */
int cy_length_fwd(const float* v, float* len, long long rows, int D, void* stream);
int cy_length_bwd(const float* v, const float* len, const float* dlen, float* dv, long long rows, int D, void* stream);

/* ------------------------------------------------------------------ losses (forward value + input gradient in one launch)
 * loss_out: one float, overwritten.  Gradients are d(loss)/d(input) for an upstream gradient of 1. */
/* loss_fns.py:187-204 + utils.py:69-85; caps [B,g,g,5] fp32, y [B,g,g,5+C] fp64 */
int cy_darkcapsule_loss(const float* caps, const double* y, int ystride, float* loss_out, float* dcaps,
                        int B, int cells, void* stream);
/* loss_fns.py:145-160 (darkcapsule2_loss): caps [B,g,g,5+C] fp32, y [B,g,g,5+C] fp64 */
int cy_darkcapsule2_loss(const float* caps, const double* y, float* loss_out, float* dcaps, int B, int cells, int C,
                         void* stream);
/* loss_fns.py:163-184 (darkcapsule3_loss, recon off): caps [B,g,g,C,D] fp32 with D = 5 + 16, y [B,g,g,5+C] fp64 */
int cy_darkcapsule3_loss(const float* caps, const double* y, float* loss_out, float* dcaps, int B, int cells, int C, int D,
                         void* stream);
/* loss_fns.py:11-23 margin part; scores [B,C], labels int64 */
int cy_margin_loss(const float* scores, const long long* y, float* loss_out, float* dscores,
                   int B, int C, void* stream);
/* coef * sum((x-recon)^2) / B added into loss_out (loss_fns.py:19-21); drecon written */
int cy_recon_loss_add(const float* x, const float* recon, float coef_over_B, float* loss_out, float* drecon,
                      long long n, void* stream);
/* loss_fns.py:60-142 (dense form); y_pred [B,g,g,5nb+C] fp32, y_true [B,g,g,5+C] fp64; avg_iou_out one float */
int cy_dark_loss(const float* y_pred, const double* y_true, float* loss_out, float* avg_iou_out, float* dpred,
                 int B, int g, int nb, int C, float l_coord, float l_noobj, float img_size, void* stream);
/* out = in * (*scalar)  (backward of a scalar loss with upstream gradient on the device) */
int cy_scale_by_device_scalar(const float* in, const float* scalar, float* out, long long n, void* stream);

/* ------------------------------------------------------------------ data movement / elementwise around the path
 * Layout permutes standing in for the reference's view/permute/cat glue (models.py:8-19, 80-82):
 * out[b][i1][i2][i3] (contiguous, dims nb x d1 x d2 x d3) = in[b*sb + i1*s1 + i2*s2 + i3*s3];
 * scatter=1 runs the inverse (in is contiguous, out is strided): the backward of the gather. */
/* Device side of the input pipeline: uint8 NHWC [B,H,W,C] -> fp32 (x - 128) / 128, the centring of
 * utils.py:122-123, written NCHW (to_nchw = 1: the layout main.py:57-59 hands to model.forward) or NHWC.  Exact:
 * every value is k / 128.  Lets the host ship 1 byte per sample instead of 4 (GTSRB) or 8 (GTSDB float64). */
int cy_center_u8(const unsigned char* src, float* dst, int B, int H, int W, int C, int to_nchw, void* stream);
int cy_permute4(const float* in, float* out, long long nb, int d1, int d2, int d3, long long sb, long long s1,
                long long s2, long long s3, int scatter, void* stream);
/* nn.MaxPool2d(2) on NHWC (models.py:135-195): x [B,2Ho,2Wo,C] -> y [B,Ho,Wo,C]; idx keeps the argmax (0..3) */
/* conv -> BatchNorm -> LeakyReLU -> MaxPool2d(2) blocks (models.py:135 ... 195) without the full-resolution activation:
 * Y[B][Ho][Wo][C] = max over the 2 x 2 window of lrelu(Z * scale + shift) (Z [B][2 Ho][2 Wo][C], the block's raw convolution output),
 * idx = the winning position 0 .. 3 (first of equals in row-major order, as nn.MaxPool2d); C % 4 == 0.
 * Backward: D[B][2 Ho][2 Wo][C] = [position == idx] dP lrelu'(Z * scale + shift) -- the premasked gradient the block's backward takes
 * (cy_conv3x3_winograd_wgrad_bn premasked = 1 / cy_bn_bwd_apply with slope 1) -- and red[C][2] (doubles, zeroed by the caller) +=
 * (sum D, sum D xhat): the cy_bn_bwd_reduce pass of that block.  C a multiple of 64 (or 4 .. 32). */
int cy_affine_act_maxpool2(const float* Z, const float* scale, const float* shift, float slope, float* Y, unsigned char* idx,
                           int B, int Ho, int Wo, int C, void* stream);
int cy_maxpool2_bwd_bn(const float* dP, const unsigned char* idx, const float* Z, const float* scale, const float* shift,
                       const float* mean, const float* invstd, float slope, float* D, double* red, int B, int Ho, int Wo, int C,
                       void* stream);
int cy_maxpool2_fwd(const float* x, float* y, unsigned char* idx, int B, int Ho, int Wo, int C, void* stream);
int cy_maxpool2_bwd(const float* dy, const unsigned char* idx, float* dx, int B, int Ho, int Wo, int C, void* stream);
/* nn.Upsample (nearest, integer factor f; models.py:99-105) on NHWC and its backward */
int cy_upsample_fwd(const float* x, float* y, int B, int Hi, int Wi, int C, int f, void* stream);
int cy_upsample_bwd(const float* dy, float* dx, int B, int Hi, int Wi, int C, int f, void* stream);
int cy_tanh_fwd(const float* x, float* y, long long n, void* stream);
int cy_tanh_bwd(const float* y, const float* dy, float* dx, long long n, void* stream);
/* DarkNet head (models.py:226-236): sigmoid on the first `split` values of each cell, softmax on the other C */
int cy_yolo_head_fwd(const float* x, float* y, long long cells, int split, int C, void* stream);
int cy_yolo_head_bwd(const float* y, const float* dy, float* dx, long long cells, int split, int C, void* stream);
/* Eval / predict side: utils.y_to_boxes_vec (utils.py:288-334 with 233-269) on the device.
 * y [B][g][g][5 nb + C] network output (or ground truth, nb = 1); boxes with confidence > conf_th come out in
 * np.argwhere order: image_idx[n], xy[n][4] = (x1, y1, x2, y2) in pixels (double, the reference's numpy precision),
 * cls[n] (only if C > 0).  image_hw: [B][2] int64 (height, width) per image, or NULL for img_h x img_w everywhere.
 * *count = number of boxes found (may exceed max_boxes: only the first max_boxes are written). */
int cy_yolo_decode_boxes(const float* y, const long long* image_hw, double img_h, double img_w, int B, int g, int nb, int C,
                         float conf_th, int* count, int* image_idx, double* xy, int* cls, int max_boxes, void* stream);
/* Detection metric on the device: metrics.single_img_confusion / calc_iou_individual (metrics.py:99-147) over a
 * batch.  gt / pr: boxes as cy_yolo_decode_boxes returns them (image index ascending, xy[n][4] double).  Adds to
 * out4[0..2] TP (ground-truth boxes overlapped by some prediction with IoU > iou_th), FP (predictions that overlap no
 * ground truth), FN; out4[3] counts malformed boxes (x1 > x2 or y1 > y2; the reference raises AssertionError) and
 * gets +2^20 per image with more than max_per_image boxes.  The caller zeroes out4. */
int cy_detect_confusion(const int* gt_idx, const double* gt_xy, int n_gt, const int* pr_idx, const double* pr_xy, int n_pr,
                        int n_images, double iou_th, int max_per_image, int* out4, void* stream);
/* torch.gather of the labelled capsule (models.py:122): backward=0: out[B][D] = caps[b][y[b]][:];
 * backward=1: caps is d(out) [B][D], out = d(caps) [B][C][D] (zero off the labelled capsule) */
int cy_pick_capsule(const float* caps, const long long* y, float* out, int B, int C, int D, int backward, void* stream);

/* ------------------------------------------------------------------ optimizer
 * torch.optim.Adam step (main.py:72,280) for a list of tensors in ONE launch.
 * table: device array of n_tensors records {param, grad, exp_avg, exp_avg_sq, numel} (5 x 8 bytes);
 * blockmap: device array of n_blocks int2 {tensor index, chunk index}. */
/* zero-fill (asynchronous on `stream`) of the per-step scratch arena that the statistics kernels accumulate into */
int cy_zero_bytes(void* p, long long nbytes, void* stream);
/* Gradient bucket of the data-parallel step (between loss.backward() and optimizer.step(), main.py:71-72): every
 * tensor of a list copied into / out of one flat buffer with a scale, ONE launch.  table[k] = {float* tensor,
 * int64 offset into flat, int64 numel}; blockmap[b] = {tensor index, chunk index}. */
int cy_multi_copy(const void* table, const void* blockmap, int n_blocks, int chunk, float* flat, int unpack, float scale,
                  void* stream);
int cy_adam_multi(const void* table, const void* blockmap, int n_blocks, int chunk,
                  float lr, float beta1, float beta2, float eps, float bias_corr1, float bias_corr2,
                  void* stream);
/* The same step with its six scalars {lr, beta1, beta2, eps, bias_corr1, bias_corr2} read from device memory (hyper): for a
 * training step captured in a HIP graph, whose kernel arguments are frozen (the host refreshes hyper before every replay). */
int cy_adam_multi_dev(const void* table, const void* blockmap, int n_blocks, int chunk, const float* hyper, void* stream);

#ifdef __cplusplus
}
#endif
#endif
