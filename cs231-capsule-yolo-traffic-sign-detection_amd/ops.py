"""torch.autograd wrappers over the C-ABI kernels (include/capsyolo_hip.h).

PyTorch supplies device memory, the current HIP stream and the autograd graph; every
arithmetic step below is a call into libcapsyolo_hip.so.  All activations are NHWC fp32
tensors ([B,H,W,C], plain contiguous).  No function here has a CPU or ATen fallback.
"""
import ctypes as C

import torch

from . import _lib
from ._lib import ConvGemm, ConvWgrad, RoutingBwd, RoutingFwd, call, query


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


class KernelTimer(object):
    """HIP-event brackets around selected launches on the stream they are launched on (bench.py's live
    per-kernel durations).  Disabled by default: then ``range`` costs one attribute test."""

    def __init__(self):
        self.enabled = False
        self.events = {}

    class _Range(object):
        def __init__(self, timer, name):
            self.timer, self.name = timer, name

        def __enter__(self):
            if self.timer.enabled:
                self.start = torch.cuda.Event(enable_timing=True)
                self.start.record(torch.cuda.current_stream())

        def __exit__(self, *exc):
            if self.timer.enabled:
                end = torch.cuda.Event(enable_timing=True)
                end.record(torch.cuda.current_stream())
                self.timer.events.setdefault(self.name, []).append((self.start, end))
            return False

    def range(self, name):
        return KernelTimer._Range(self, name)

    def reset(self):
        self.events = {}

    def summary(self):
        """name -> (launches, mean milliseconds); call after a device synchronize."""
        return dict((k, (len(v), sum(a.elapsed_time(b) for a, b in v) / len(v))) for k, v in self.events.items())


timer = KernelTimer()


def _ptr(t):
    return None if t is None else C.c_void_p(t.data_ptr())


def _f32(t, name='tensor'):
    if not (t.is_cuda and t.dtype == torch.float32):
        raise _lib.HipExtensionError('%s must be a float32 tensor on the GPU (got %s on %s); there is no CPU fallback'
                                     % (name, t.dtype, t.device))
    return t if t.is_contiguous() else t.contiguous()


def _empty(shape, like, dtype=torch.float32):
    return torch.empty(shape, dtype=dtype, device=like.device)


class _ZeroPool(object):
    """Scratch that must start at zero (BatchNorm statistics and backward sums are accumulated with atomics): slices of
    one arena per device that is zeroed by ONE cy_zero_bytes launch per training step (FusedBackbone.forward calls
    reset()), instead of one fill kernel per buffer (~23 per step at the headline model)."""

    def __init__(self):
        self.pools = {}

    def reset(self, device):
        st = self.pools.get(device)
        if st is None:
            return
        if st['need'] > st['buf'].numel():                       # grow: the new arena is born zeroed
            st['buf'] = torch.zeros(2 * st['need'], dtype=torch.uint8, device=device)
        elif st['off']:
            call('cy_zero_bytes', C.c_void_p(st['buf'].data_ptr()), st['off'], _stream())
        st['off'], st['need'] = 0, 0

    def take(self, shape, dtype, device):
        n = 1
        for d in shape:
            n *= int(d)
        nbytes = (n * torch.empty((), dtype=dtype).element_size() + 255) & ~255
        st = self.pools.get(device)
        if st is None:
            st = self.pools[device] = {'buf': torch.zeros(1 << 20, dtype=torch.uint8, device=device), 'off': 0, 'need': 0}
        st['need'] += nbytes
        if st['off'] + nbytes > st['buf'].numel():               # arena exhausted: a fresh fill now, a larger arena next step
            return torch.zeros(shape, dtype=dtype, device=device)
        out = st['buf'][st['off']:st['off'] + nbytes].view(dtype)[:n].view(shape)
        st['off'] += nbytes
        return out


zero_pool = _ZeroPool()
_zero_consts = {}


def _const_zeros(n, like):
    """A shared, never-written vector of zeros: the gradient of a conv bias in front of BatchNorm (analytically 0)."""
    key = (int(n), like.device)
    t = _zero_consts.get(key)
    if t is None:
        t = _zero_consts[key] = torch.zeros(n, dtype=torch.float32, device=like.device)
    return t


# ------------------------------------------------------------------------------------------------ convolution
def _x_geometry(x, nchw):
    if nchw:
        B, Cin, Hi, Wi = x.shape
        return B, Hi, Wi, Cin, (Cin * Hi * Wi, Wi, 1, Hi * Wi)
    B, Hi, Wi, Cin = x.shape
    return B, Hi, Wi, Cin, (Hi * Wi * Cin, Wi * Cin, Cin, 1)


STATS_COPIES = 16       # CY_STATS_COPIES of include/capsyolo_hip.h: BatchNorm statistics are accumulated in 16 striped copies
USE_CONV1_MOMENTS = True  # ... whose BatchNorm statistics come from the 28 x 28 moment matrix of the input patches (csrc/conv1_moments.hip)
FUSE_BN_BWD_APPLY_BF16 = True   # bf16 path: BatchNorm-backward pass 2 inside the weight gradient (cy_conv_wgrad_bf16_bn) where the gradient arrives premasked
USE_CONV1_ONEPASS = True  # ... and whose backward then needs ONE pass over the gradient (cy_conv1_bn_bwd_onepass)
CONV1_MOMENTS_MIN_PIXELS = 1 << 18   # ... from this many pixels on: below, its three launches cost more than the one recompute pass saves
USE_CONV1_BWD = True     # ... and the backward of its whole conv -> BatchNorm -> LeakyReLU block without z / dz in memory
USE_CONV1 = True         # 3 -> {32, 64, 128} channels, 3x3, NCHW image (the backbones' first layer): dedicated store-bound kernels
USE_WINOGRAD_S2_DGRAD = True   # ... and their input gradient (K = Cout: short reductions; kept switchable)
USE_WINOGRAD_S2 = True   # 4x4 / stride 2 / pad 1 layers: fused Winograd F(2x2,2x2) forward on the space-to-depth view
USE_WINOGRAD = True      # 3x3 / stride 1 / pad 1 layers with Cin % 8 == 0 take the fused Winograd F(2x2,3x3) kernel
USE_WINOGRAD4_S2 = True  # 4x4 / stride-2 layers: forward on F(4x4,2x2) (winograd4_s2.hip: 1.44x fewer MFMAs than F(2x2,2x2)) from
USE_WINOGRAD4_S2_DGRAD = True      # ... and their input gradient (the same kernel over dY shifted by one pixel, scattered to the four classes)
WINOGRAD4_S2_MIN_PIXELS = 1 << 16   # this many output pixels on (its 16 x 32-pixel blocks are the F(2x2,2x2) kernel's)
USE_WINOGRAD4 = True     # ... forward and input gradient on F(4x4,3x3) (winograd4.hip: 1.78x fewer MFMAs) from WINOGRAD4_MIN_PIXELS output pixels on
WINOGRAD4_MIN_PIXELS = 1 << 17   # 512 output pixels x 64 channels per block: far below, its tile grid leaves most of the 256 CUs idle.  The value is a
                                 # trade of speed against LeakyReLU kink flips, measured in round 4 (tools/bench_darknet.py 16 <log2 pixels>): darknet_d
                                 # 8.83 ms per step at 2^18 (round 3's value), 8.69 at 2^17 / 2^16, 8.47 at 2^15, 8.44 at 2^13 -- but F(4x4,3x3) is 2.3e-6
                                 # from fp64 against 3.6e-7 for F(2x2,3x3), and on the 19-block net every further layer on it flips more pre-activations
                                 # inside the rounding noise: gradients of the lower blocks against an fp64 run of the same net (128 x 128, three seeds) 5e-3 / 4e-4 /
                                 # 5e-3 at 2^18 and 2e-2 / 9e-3 / 1.7e-2 at 2^15 (the reference's own fp32 path: 5e-6 / 3e-3 / 8e-3).  2^17 takes the
                                 # layers where the kernel buys most (>= 131072 output pixels) and leaves the deep small maps on F(2x2,3x3).


FUSE_BN_BWD_REDUCE = True  # ... and that block's input-gradient epilogue sums the producer's BatchNorm backward
FUSE_BN_BWD_APPLY = True   # 3x3 blocks whose weight gradient is the Winograd kernel: BatchNorm backward pass 2 inside that kernel
FUSE_INPUT_AFFINE = True  # a BN+LeakyReLU block in front of a 4x4/stride-2 block hands over its raw output + scale/shift


def s2_fusable(k, stride, pad, cin, cout, hi, wi, nchw=False):
    """The 4x4 / stride-2 Winograd forward AND weight-gradient kernels both take this layer (then they can apply the
    producer's BatchNorm + LeakyReLU on their input loads)."""
    return (USE_WINOGRAD and USE_WINOGRAD_S2 and k == 4 and stride == 2 and pad == 1 and not nchw and cin % 32 == 0
            and cout % 64 == 0 and hi % 2 == 0 and wi % 2 == 0)


def _winograd_ok(k, stride, pad, cin, nchw):
    return USE_WINOGRAD and k == 3 and stride == 1 and pad == 1 and not nchw and cin % 8 == 0 and cin >= 8


def _winograd(x, weight, bias, stats, transpose, tag, out_slope=1.0):
    """x [B,H,W,Cg] NHWC; weight in PyTorch layout; transpose=True computes the input gradient of the layer.
    out_slope != 1: LeakyReLU epilogue (eval forward with the BatchNorm folded into weight / bias)."""
    B, H, W_, Cg = x.shape
    Cout_l, Cin_l = weight.shape[0], weight.shape[1]
    n = Cin_l if transpose else Cout_l
    st = _stream()
    f4 = USE_WINOGRAD4 and B * H * W_ >= WINOGRAD4_MIN_PIXELS
    u = _empty((query('cy_wino4_packed_floats' if f4 else 'cy_wino_packed_floats', Cg, n),), x)
    call('cy_wino4_pack_weights' if f4 else 'cy_wino_pack_weights', _ptr(weight), _ptr(u), Cout_l, Cin_l, 1 if transpose else 0, st)
    y = _empty((B, H, W_, n), x)
    # F(2x2,3x3) launches without an epilogue that fill at most half the chip (DarkNet's 13 x 13 input gradients) split their reduction
    nws = 0 if f4 else query('cy_wino_split_ws_floats', B, H, W_, Cg, n, int(bias is None and stats is None and out_slope == 1.0))
    ws = _empty((nws,), x) if nws > 0 else None
    with timer.range(('conv_wino4_' if f4 else 'conv_wino_') + ('dgrad/' if transpose else 'fwd/') + tag):
        if f4:
            call('cy_conv3x3_winograd4', _ptr(x), _ptr(u), _ptr(y), _ptr(bias), _ptr(stats), float(out_slope), B, H, W_, Cg, n, st)
        else:
            call('cy_conv3x3_winograd_ws', _ptr(x), _ptr(u), _ptr(y), _ptr(bias), _ptr(stats), float(out_slope), B, H, W_, Cg, n,
                 _ptr(ws), nws, st)
    return y


def conv1_ok(x, weight, k, stride, pad, nchw):
    """The dedicated first-layer kernels (csrc/conv1.hip) apply: 3 -> {32, 64, 128} channels, 3x3 s1 p1, NCHW image."""
    return (USE_CONV1 and nchw and k == 3 and stride == 1 and pad == 1 and x.dim() == 4 and x.shape[1] == 3
            and weight.shape[0] in (32, 64, 128) and x.shape[3] % 32 == 0 and x.is_contiguous())


def conv1_affine_act(x, weight, bias, scale, shift, slope, tag='conv', out_bf16=False):
    """lrelu((conv(x) + bias) * scale + shift) in one pass of the first-layer kernel (the second pass of its conv ->
    BatchNorm -> LeakyReLU block: recomputing the layer costs less than reading its 2.8 GB output back).
    out_bf16: the activation is written as bf16 (the bf16 path's second block reads it as such)."""
    B, _, Hi, Wi = x.shape
    Cout = weight.shape[0]
    with timer.range('conv1_fwd_act/' + tag):
        if out_bf16:
            out = torch.empty((B, Hi, Wi, Cout), dtype=torch.bfloat16, device=x.device)
            call('cy_conv1_3x3_fwd_act_bf16', _ptr(x), _ptr(weight.contiguous()), _ptr(bias), _ptr(out), _ptr(scale), _ptr(shift),
                 float(slope), B, Hi, Wi, Cout, _stream())
        else:
            out = _empty((B, Hi, Wi, Cout), x)
            call('cy_conv1_3x3_fwd', _ptr(x), _ptr(weight.contiguous()), _ptr(bias), _ptr(out), None, _ptr(scale), _ptr(shift),
                 float(slope), B, Hi, Wi, Cout, _stream())
    return out


def conv_forward(x, weight, bias, k, stride, pad, nchw=False, stats=None, relu=False, tag='conv', in_affine=None, lrelu=None):
    """z[B,Ho,Wo,Cout] = conv2d(x) (+bias, optional fused ReLU); optional BN statistics side output.
    lrelu = slope in [0, 1]: LeakyReLU epilogue (the eval-mode forward of a block whose BatchNorm is folded into weight / bias)."""
    if lrelu is not None and (relu or stats is not None or in_affine is not None or not 0.0 <= lrelu <= 1.0):
        raise _lib.HipExtensionError('conv_forward: the LeakyReLU epilogue (slope %r) is the eval-mode forward: no statistics, no '
                                     'fused input affine, slope in [0, 1]' % (lrelu,))
    osl = 1.0 if lrelu is None else float(lrelu)
    x, weight = _f32(x, 'conv input'), _f32(weight, 'conv weight')
    B, Hi, Wi, Cin, xs = _x_geometry(x, nchw)
    Cout = weight.shape[0]
    Ho, Wo = (Hi + 2 * pad - k) // stride + 1, (Wi + 2 * pad - k) // stride + 1
    if in_affine is not None and not s2_fusable(k, stride, pad, Cin, Cout, Hi, Wi, nchw):
        raise _lib.HipExtensionError('a fused input affine needs the 4x4/stride-2 Winograd kernels (got k=%d s=%d Cin=%d Cout=%d)'
                                     % (k, stride, Cin, Cout))
    if not relu and _winograd_ok(k, stride, pad, Cin, nchw):
        return _winograd(x, weight, bias, stats, False, tag, osl)
    if (USE_WINOGRAD and USE_WINOGRAD_S2 and not relu and k == 4 and stride == 2 and pad == 1 and not nchw
            and Cin % 8 == 0 and Hi % 2 == 0 and Wi % 2 == 0 and x.is_contiguous()):
        st = _stream()
        y = _empty((B, Ho, Wo, Cout), x)
        isc, ish, isl = in_affine if in_affine is not None else (None, None, 1.0)
        if USE_WINOGRAD4_S2 and B * Ho * Wo >= WINOGRAD4_S2_MIN_PIXELS and query('cy_wino4s2_ok', B, Hi, Wi, Cin, Cout):
            u = _empty((query('cy_wino4s2_packed_floats', Cin, Cout),), x)
            call('cy_wino4s2_pack_weights', _ptr(weight), _ptr(u), Cout, Cin, st)
            with timer.range('conv_wino42_fwd/' + tag):
                call('cy_conv4x4s2_winograd4', _ptr(x), _ptr(u), _ptr(y), _ptr(bias), _ptr(stats), _ptr(isc), _ptr(ish),
                     float(isl), osl, B, Hi, Wi, Cin, Cout, st)
            return y
        u = _empty((query('cy_wino2_packed_floats', Cin, Cout),), x)
        call('cy_wino2_pack_weights', _ptr(weight), _ptr(u), Cout, Cin, st)
        with timer.range('conv_wino2_fwd/' + tag):
            call('cy_conv4x4s2_winograd', _ptr(x), _ptr(u), _ptr(y), _ptr(bias), _ptr(stats), _ptr(isc), _ptr(ish),
                 float(isl), osl, B, Hi, Wi, Cin, Cout, st)
        return y
    if not relu and lrelu is None and conv1_ok(x, weight, k, stride, pad, nchw):
        # the backbones' first layer (store-bound): persistent waves, operands from registers / L2
        st = _stream()
        z = _empty((B, Ho, Wo, Cout), x)
        with timer.range('conv1_fwd/' + tag):
            call('cy_conv1_3x3_fwd', _ptr(x), _ptr(weight.contiguous()), _ptr(bias), _ptr(z), _ptr(stats), None, None, 1.0,
                 B, Hi, Wi, Cout, st)
        return z
    st = _stream()
    wp = _empty((query('cy_conv_packed_floats', k * k * Cin, Cout),), x)
    call('cy_conv_pack_weights', _ptr(weight), _ptr(wp), Cout, Cin, k, k, k, k, 0, 0, 1, 0, st)
    z = _empty((B, Ho, Wo, Cout), x)
    a = ConvGemm(X=x.data_ptr(), Wp=wp.data_ptr(), Y=z.data_ptr(),
                 bias=bias.data_ptr() if bias is not None else None,
                 stats=stats.data_ptr() if stats is not None else None,
                 xs_b=xs[0], xs_y=xs[1], xs_x=xs[2], xs_c=xs[3], B=B, Hi=Hi, Wi=Wi, Cin=Cin, Ho=Ho, Wo=Wo, N=Cout,
                 TH=k, TW=k, in_stride=stride, dy0=-pad, dx0=-pad, dstep=1,
                 Hy=Ho, Wy=Wo, out_stride=1, out_oy=0, out_ox=0, act=1 if relu else 2 if lrelu is not None else 0, act_slope=osl)
    ws = _gemm_workspace(a, x)
    with timer.range('conv_gemm_fwd/' + tag):
        call('cy_conv_gemm', C.byref(a), st)
    del ws
    return z


def _gemm_workspace(a, like):
    """The optional workspace of one cy_conv_gemm launch (split reduction of under-filled grids, capsyolo_hip.h); the returned tensor
    must stay referenced until the call has been issued."""
    n = query('cy_conv_gemm_ws_floats', C.byref(a))
    if n <= 0:
        return None
    ws = _empty((n,), like)
    a.ws, a.ws_floats = ws.data_ptr(), n
    return ws


def dgrad_classes(Hi, Wi, k, stride, pad):
    """Input-gradient plan: one implicit GEMM per output-parity class (py, px) of the stride.

    Input pixel y = stride*oy' + py receives dz[oy' + dy0 - a] * W[kh0 + stride*a] for a in range(TH)
    (forward relation y = oy*stride - pad + kh).  Returns dicts with the cy_conv_gemm_t / cy_conv_pack_weights
    fields of each non-empty class; raises if some input pixel has no tap at all (kernel < stride)."""
    out = []
    for py in range(stride):
        kh0 = (py + pad) % stride
        TH = len(range(kh0, k, stride))
        for px in range(stride):
            kw0 = (px + pad) % stride
            TW = len(range(kw0, k, stride))
            Hv, Wv = (Hi - py + stride - 1) // stride, (Wi - px + stride - 1) // stride
            if Hv <= 0 or Wv <= 0:
                continue
            if TH == 0 or TW == 0:
                raise _lib.HipExtensionError('conv_dgrad: kernel %d / stride %d leaves input pixels without taps'
                                             % (k, stride))
            out.append(dict(TH=TH, TW=TW, kh0=kh0, kw0=kw0, kstep=stride, Ho=Hv, Wo=Wv,
                            dy0=(py + pad - kh0) // stride, dx0=(px + pad - kw0) // stride, dstep=-1,
                            out_stride=stride, out_oy=py, out_ox=px))
    return out


def conv_dgrad(dz, weight, in_shape, k, stride, pad, tag='conv', bn_fuse=None, info=None):
    """dx[B,Hi,Wi,Cin] (NHWC) from dz[B,Ho,Wo,Cout]: one GEMM per output-parity class of the stride.
    bn_fuse = (z, scale, shift, mean, invstd, slope, red): dx is the gradient with respect to lrelu(z*scale+shift) of the
    producer block; the epilogues also accumulate that BatchNorm's backward sums into red[STATS_COPIES][Cin][2]."""
    dz, weight = _f32(dz, 'grad'), _f32(weight, 'conv weight')
    B, Hi, Wi, Cin = in_shape
    _, Ho, Wo, Cout = dz.shape
    if _winograd_ok(k, stride, pad, Cout, False):
        if bn_fuse is not None:
            raise _lib.HipExtensionError('bn_fuse is implemented in the direct input-gradient kernel only')
        return _winograd(dz, weight, None, None, True, tag)
    st = _stream()
    dx = _empty((B, Hi, Wi, Cin), dz)
    if (USE_WINOGRAD and USE_WINOGRAD_S2 and USE_WINOGRAD_S2_DGRAD and k == 4 and stride == 2 and pad == 1 and Cin % 64 == 0
            and Cout % 8 == 0 and Hi % 2 == 0 and Wi % 2 == 0 and dz.is_contiguous()):
        bz = bsc = bsh = bmu = bis = bred = None
        bsl = 0.0
        if bn_fuse is not None:
            bz, bsc, bsh, bmu, bis, bsl, bred = bn_fuse
        if (USE_WINOGRAD4_S2 and USE_WINOGRAD4_S2_DGRAD and B * Ho * Wo >= WINOGRAD4_S2_MIN_PIXELS
                and query('cy_wino4s2_dgrad_ok', B, Hi, Wi, Cin, Cout)):
            u = _empty((query('cy_wino4s2_dgrad_packed_floats', Cin, Cout),), dz)
            call('cy_wino4s2_pack_dgrad_weights', _ptr(weight), _ptr(u), Cout, Cin, st)
            with timer.range('conv_wino42_dgrad/' + tag):
                call('cy_conv4x4s2_winograd4_dgrad', _ptr(dz), _ptr(u), _ptr(dx), _ptr(bz), _ptr(bsc), _ptr(bsh), _ptr(bmu),
                     _ptr(bis), float(bsl), _ptr(bred), B, Hi, Wi, Cin, Cout, st)
        else:
            u = _empty((query('cy_wino2_dgrad_packed_floats', Cin, Cout),), dz)
            call('cy_wino2_pack_dgrad_weights', _ptr(weight), _ptr(u), Cout, Cin, st)
            with timer.range('conv_wino2_dgrad/' + tag):
                call('cy_conv4x4s2_winograd_dgrad', _ptr(dz), _ptr(u), _ptr(dx), _ptr(bz), _ptr(bsc), _ptr(bsh), _ptr(bmu),
                     _ptr(bis), float(bsl), _ptr(bred), B, Hi, Wi, Cin, Cout, st)
        if info is not None and bn_fuse is not None:
            info['premasked'] = True      # with the fused sums this kernel stores dx * lrelu'(y), not dx (capsyolo_hip.h)
        return dx
    wp = _empty((query('cy_conv_packed_floats', ((k + stride - 1) // stride) ** 2 * Cout, Cin),), dz)
    for c in dgrad_classes(Hi, Wi, k, stride, pad):
        call('cy_conv_pack_weights', _ptr(weight), _ptr(wp), Cout, Cin, k, k, c['TH'], c['TW'], c['kh0'], c['kw0'],
             c['kstep'], 1, st)
        a = ConvGemm(X=dz.data_ptr(), Wp=wp.data_ptr(), Y=dx.data_ptr(), bias=None, stats=None,
                     xs_b=Ho * Wo * Cout, xs_y=Wo * Cout, xs_x=Cout, xs_c=1, B=B, Hi=Ho, Wi=Wo, Cin=Cout,
                     Ho=c['Ho'], Wo=c['Wo'], N=Cin, TH=c['TH'], TW=c['TW'], in_stride=1,
                     dy0=c['dy0'], dx0=c['dx0'], dstep=c['dstep'],
                     Hy=Hi, Wy=Wi, out_stride=c['out_stride'], out_oy=c['out_oy'], out_ox=c['out_ox'], act=0)
        if bn_fuse is not None:
            bz, bsc, bsh, bmu, bis, bsl, bred = bn_fuse
            a.bn_z, a.bn_scale, a.bn_shift = bz.data_ptr(), bsc.data_ptr(), bsh.data_ptr()
            a.bn_mean, a.bn_invstd, a.bn_red, a.bn_slope = bmu.data_ptr(), bis.data_ptr(), bred.data_ptr(), float(bsl)
        ws = _gemm_workspace(a, dz)
        with timer.range('conv_gemm_dgrad/' + tag):
            call('cy_conv_gemm', C.byref(a), st)
        del ws
    return dx


def _wino_wgrad_ok(k, stride, pad, cin, cout):
    return USE_WINOGRAD and k == 3 and stride == 1 and pad == 1 and cin % 64 == 0 and cout % 64 == 0


USE_WINOGRAD4_WGRAD = True   # the 3x3 layers' weight gradient on F(3x3,4x4) (winograd4_wgrad.hip) where its shape conditions hold and the
                             # map has WINOGRAD4_MIN_PIXELS output pixels (below, the F(3x3,2x2) kernel)


def _wino4_wgrad_ok(B, H, W, cin, cout):
    return (USE_WINOGRAD4_WGRAD and B * H * W >= WINOGRAD4_MIN_PIXELS and bool(query('cy_wino4_wgrad_ok', B, H, W, cin, cout)))


def conv_wgrad(x, dz, k, stride, pad, nchw=False, tag='conv', in_affine=None):
    x, dz = _f32(x, 'conv input'), _f32(dz, 'grad')
    B, Hi, Wi, Cin, xs = _x_geometry(x, nchw)
    _, Ho, Wo, Cout = dz.shape
    st = _stream()
    dW = _empty((Cout, Cin, k, k), dz)
    if not nchw and USE_WINOGRAD and k == 3 and stride == 1 and pad == 1 and _wino4_wgrad_ok(B, Hi, Wi, Cin, Cout):
        ws = _empty((query('cy_wino4_wgrad_ws_floats', B, Hi, Wi, Cin, Cout),), dz)
        with timer.range('conv_wino4_wgrad/' + tag):
            call('cy_conv3x3_winograd4_wgrad', _ptr(x), _ptr(dz), _ptr(dW), _ptr(ws), B, Hi, Wi, Cin, Cout, st)
        return dW
    if not nchw and _wino_wgrad_ok(k, stride, pad, Cin, Cout):
        ws = _empty((query('cy_wino_wgrad_ws_floats', B, Cin, Cout),), dz)
        with timer.range('conv_wino_wgrad/' + tag):
            call('cy_conv3x3_winograd_wgrad', _ptr(x), _ptr(dz), _ptr(dW), _ptr(ws), B, Hi, Wi, Cin, Cout, st)
        return dW
    if (USE_WINOGRAD and USE_WINOGRAD_S2 and k == 4 and stride == 2 and pad == 1 and not nchw and Cin % 32 == 0
            and Cout % 64 == 0 and Hi % 2 == 0 and Wi % 2 == 0 and x.is_contiguous() and dz.is_contiguous()):
        ws = _empty((query('cy_wino2_wgrad_ws_floats', B, Cin, Cout),), dz)
        isc, ish, isl = in_affine if in_affine is not None else (None, None, 1.0)
        with timer.range('conv_wino2_wgrad/' + tag):
            call('cy_conv4x4s2_winograd_wgrad', _ptr(x), _ptr(dz), _ptr(dW), _ptr(ws), _ptr(isc), _ptr(ish), float(isl),
                 B, Hi, Wi, Cin, Cout, st)
        return dW
    if (USE_CONV1 and nchw and k == 3 and stride == 1 and pad == 1 and Cin == 3 and Cout in (32, 64, 128) and Wi % 32 == 0
            and in_affine is None and x.is_contiguous() and dz.is_contiguous()):
        ws = _empty((query('cy_conv1_3x3_wgrad_ws_floats', B, Hi, Wi, Cout),), dz)
        with timer.range('conv1_wgrad/' + tag):
            call('cy_conv1_3x3_wgrad', _ptr(x), _ptr(dz), _ptr(dW), _ptr(ws), B, Hi, Wi, Cout, st)
        return dW
    if in_affine is not None:
        raise _lib.HipExtensionError('a fused input affine needs the 4x4/stride-2 Winograd weight-gradient kernel')
    a = ConvWgrad(X=x.data_ptr(), dZ=dz.data_ptr(), dW=dW.data_ptr(), slabs=None,
                  xs_b=xs[0], xs_y=xs[1], xs_x=xs[2], xs_c=xs[3], B=B, Hi=Hi, Wi=Wi, Cin=Cin, Ho=Ho, Wo=Wo, N=Cout,
                  KH=k, KW=k, stride=stride, pad=pad)
    ws = _empty((query('cy_conv_wgrad_ws_floats', C.byref(a)),), dz)
    a.slabs = ws.data_ptr()
    with timer.range('conv_wgrad/' + tag):
        call('cy_conv_wgrad', C.byref(a), st)
    return dW


FOLD_EVAL_BN = True   # eval mode: the BatchNorm is folded into weights / bias once and the block's forward is conv + LeakyReLU epilogue
PARAM_EPOCH = 0       # bumped by everything that writes parameters / running statistics through raw pointers (optim.Adam.step,
                      # cy_bn_finalize in a training forward): part of the fold cache's key next to the tensors' version counters


def _bump_param_epoch():
    global PARAM_EPOCH
    PARAM_EPOCH += 1


def fold_eval_bn(weight, bias, gamma, beta, bn):
    """(W', b') of the eval-mode block: W'[co] = W[co] * s[co], b' = (b - running_mean) * s + beta, s = gamma / sqrt(running_var
    + eps) (cy_bn_fold_eval), cached on the BatchNorm module until a parameter or running statistic changes."""
    ts = (weight, bias, gamma, beta, bn.running_mean, bn.running_var)
    key = (PARAM_EPOCH, float(bn.eps)) + tuple((t.data_ptr(), t._version) if t is not None else None for t in ts)
    hit = getattr(bn, '_cy_fold', None)
    if hit is not None and hit[0] == key:
        return hit[1], hit[2]
    weight = _f32(weight, 'conv weight')
    Wf, bf = torch.empty_like(weight), _empty((weight.shape[0],), weight)
    call('cy_bn_fold_eval', _ptr(weight), _ptr(bias), _ptr(gamma), _ptr(beta), _ptr(bn.running_mean), _ptr(bn.running_var),
         float(bn.eps), _ptr(Wf), _ptr(bf), weight.shape[0], weight.numel() // weight.shape[0], _stream())
    bn._cy_fold = (key, Wf, bf)
    return Wf, bf


class ConvBlockCfg(object):
    """Static description of one conv (+BatchNorm) (+activation) block."""

    def __init__(self, k, stride, pad, nchw_in=False, bn=None, slope=None, name='conv', defer_act=False, in_slope=None):
        self.k, self.stride, self.pad, self.nchw_in, self.name = k, stride, pad, nchw_in, name
        self.defer_act = defer_act   # return (z, scale, shift): the consumer applies BatchNorm + LeakyReLU on its loads
        self.in_slope = in_slope     # not None: x is the producer's raw output, in_scale / in_shift come with it
        self.in_holder = None        # the producer's hand-over dict (mean, invstd; this block's backward fills 'red')
        self.bn = bn            # module with running_mean / running_var / momentum / eps / training, or None
        self.slope = slope      # None: no activation; 0.0: ReLU; else LeakyReLU slope


SYNC_BN = False     # data parallel only: BatchNorm statistics (forward sums, backward sums) summed over all ranks, so
                    # that N ranks x batch B train exactly like one process on the global batch N*B


def _sync_world():
    import torch.distributed as dist
    if SYNC_BN and dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        return dist, dist.get_world_size()
    return None, 1


class _ConvBlock(torch.autograd.Function):
    """conv -> [BatchNorm (batch statistics from the conv epilogue)] -> [Leaky]ReLU, saving only x and z."""

    @staticmethod
    def forward(ctx, x, weight, bias, gamma, beta, cfg, in_scale=None, in_shift=None):
        x = _f32(x, 'conv input')
        weight = _f32(weight, 'conv weight')
        st = _stream()
        N = weight.shape[0]
        bn = cfg.bn
        ctx.cfg, ctx.has_bias, ctx.bn_train = cfg, bias is not None, False
        ctx.set_materialize_grads(False)      # the (scale, shift) side outputs never get a gradient: no zero tensors for them
        ctx.holder = None
        ina = (in_scale, in_shift, float(cfg.in_slope)) if cfg.in_slope is not None else None
        ctx.in_affine = ina
        if bn is None:
            if cfg.defer_act:
                raise _lib.HipExtensionError('defer_act needs a BatchNorm block')
            relu = cfg.slope is not None and cfg.slope == 0.0
            z = conv_forward(x, weight, bias, cfg.k, cfg.stride, cfg.pad, cfg.nchw_in, None, relu, cfg.name, ina)
            out = z
            if cfg.slope is not None and not relu:
                out = torch.empty_like(z)
                call('cy_affine_act', _ptr(z), _ptr(out), None, None, float(cfg.slope), z.numel() // N, N, st)
            ctx.save_for_backward(x, weight, z)
            return out
        scale, shift = _empty((N,), x), _empty((N,), x)
        mean, invstd = _empty((N,), x), _empty((N,), x)
        slope = 1.0 if cfg.slope is None else float(cfg.slope)
        # the first block (csrc/conv1.hip): z is never written -- this pass only takes the statistics, the activation
        # pass and both backward passes recompute the convolution
        ctx.conv1_m2 = None
        ctx.conv1_fused = bool(USE_CONV1_BWD and bn.training and not cfg.defer_act and not x.requires_grad
                               and 0.0 <= slope <= 1.0 and conv1_ok(x, weight, cfg.k, cfg.stride, cfg.pad, cfg.nchw_in))
        if bn.training:
            stats = zero_pool.take((STATS_COPIES, N, 2), torch.float64, x.device)
            if ctx.conv1_fused:
                Bx, _, Hx, Wx = x.shape
                if USE_CONV1_MOMENTS and Bx * Hx * Wx >= CONV1_MOMENTS_MIN_PIXELS:
                    # sum z and sum z^2 from the moment matrix of the input patches: the layer is not computed for them
                    wsm = _empty((query('cy_conv1_3x3_stats_ws_floats', Bx, Hx),), x)
                    with timer.range('conv1_fwd_stats/' + cfg.name):
                        call('cy_conv1_3x3_stats', _ptr(x), _ptr(weight.contiguous()), _ptr(bias), _ptr(stats), _ptr(wsm),
                             Bx, Hx, Wx, N, st)
                    off = query('cy_conv1_3x3_stats_m2_offset', Bx, Hx)
                    ctx.conv1_m2 = wsm[off:off + 2048]        # the double M2[32][32]: the one-pass backward reads it
                else:
                    with timer.range('conv1_fwd_stats/' + cfg.name):
                        call('cy_conv1_3x3_fwd', _ptr(x), _ptr(weight.contiguous()), _ptr(bias), None, _ptr(stats), None, None,
                             1.0, Bx, Hx, Wx, N, st)
                z = x.new_empty(0)
                P = Bx * Hx * Wx
            else:
                z = conv_forward(x, weight, bias, cfg.k, cfg.stride, cfg.pad, cfg.nchw_in, stats, False, cfg.name, ina)
                P = z.numel() // N
            dist, world = _sync_world()
            if dist is not None:              # sum z / sum z^2 over the global batch (equal shards: dp.shard_range)
                dist.all_reduce(stats)
            call('cy_bn_finalize', _ptr(stats), P * world, _ptr(gamma), _ptr(beta), _ptr(bn.running_mean),
                 _ptr(bn.running_var), float(bn.momentum), float(bn.eps), _ptr(scale), _ptr(shift), _ptr(mean),
                 _ptr(invstd), N, _ptr(bn.num_batches_tracked), st)
            ctx.bn_train = True
            _bump_param_epoch()               # running statistics were written through raw pointers
        else:
            ctx.folded = False
            if FOLD_EVAL_BN and not cfg.defer_act and ina is None and 0.0 <= slope <= 1.0:
                # eval mode (predict_fns.py:38-43, 65-69): ONE launch per block.  The first layer applies scale / shift in its
                # own epilogue (its kernel takes them); every other layer runs on weights and bias with the BatchNorm folded in
                ctx.folded = True
                ctx.save_for_backward()
                if conv1_ok(x, weight, cfg.k, cfg.stride, cfg.pad, cfg.nchw_in):
                    call('cy_bn_eval_scale_shift', _ptr(gamma), _ptr(beta), _ptr(bn.running_mean), _ptr(bn.running_var),
                         float(bn.eps), _ptr(scale), _ptr(shift), N, st)
                    return conv1_affine_act(x, weight, bias, scale, shift, slope, cfg.name, bool(getattr(cfg, 'out_bf16', False)))
                Wf, bf = fold_eval_bn(weight, bias, gamma, beta, bn)
                out = conv_forward(x, Wf, bf, cfg.k, cfg.stride, cfg.pad, cfg.nchw_in, None, False, cfg.name, None, lrelu=slope)
                if getattr(cfg, 'out_bf16', False):
                    ob = torch.empty(out.shape, dtype=torch.bfloat16, device=out.device)
                    call('cy_cast_f32_bf16', _ptr(out), _ptr(ob), out.numel(), st)
                    return ob
                return out
            z = conv_forward(x, weight, bias, cfg.k, cfg.stride, cfg.pad, cfg.nchw_in, None, False, cfg.name, ina)
            P = z.numel() // N
            call('cy_bn_eval_scale_shift', _ptr(gamma), _ptr(beta), _ptr(bn.running_mean), _ptr(bn.running_var),
                 float(bn.eps), _ptr(scale), _ptr(shift), N, st)
        ctx.P = P
        ctx.save_for_backward(x, weight, z, scale, shift, mean, invstd, gamma, *([bias] if bias is not None else []))
        if cfg.defer_act:                     # the consumer block applies lrelu(z * scale + shift) on its loads
            ctx.mark_non_differentiable(scale, shift)
            # hand-over for the backward: the consumer's input-gradient kernel can also produce this block's
            # BatchNorm-backward sums (it has z and the gradient in registers) and leaves them in holder['red']
            ctx.holder = cfg.out_holder = {'mean': mean, 'invstd': invstd, 'red': None} if ctx.bn_train else None
            return z, scale, shift
        ctx.holder = None
        out_bf16 = bool(getattr(cfg, 'out_bf16', False))     # the consumer is a bf16 block (FusedBackbone._forward_bf16)
        if conv1_ok(x, weight, cfg.k, cfg.stride, cfg.pad, cfg.nchw_in) and 0.0 <= slope <= 1.0:
            return conv1_affine_act(x, weight, bias, scale, shift, slope, cfg.name, out_bf16)
        out = torch.empty_like(z)
        call('cy_affine_act', _ptr(z), _ptr(out), _ptr(scale), _ptr(shift), slope, P, N, st)
        if out_bf16:
            ob = torch.empty(out.shape, dtype=torch.bfloat16, device=out.device)
            call('cy_cast_f32_bf16', _ptr(out), _ptr(ob), out.numel(), st)
            return ob
        return out

    @staticmethod
    def backward(ctx, da, *unused):
        cfg = ctx.cfg
        if da is None:
            raise _lib.HipExtensionError('conv block %s: no gradient reached its output' % cfg.name)
        if getattr(ctx, 'folded', False):
            raise _lib.HipExtensionError('backward through an eval-mode BatchNorm block is not implemented')
        da_bf16 = da.dtype == torch.bfloat16 and bool(getattr(cfg, 'out_bf16', False))
        if da_bf16 and not (cfg.bn is not None and ctx.bn_train and ctx.conv1_fused):
            daf = torch.empty(da.shape, dtype=torch.float32, device=da.device)   # only the first-layer kernels read a bf16 gradient
            call('cy_cast_bf16_f32', _ptr(da.contiguous()), _ptr(daf), da.numel(), _stream())
            da, da_bf16 = daf, False
        if not da_bf16:
            da = _f32(da, 'grad')
        st = _stream()
        saved = ctx.saved_tensors
        x, weight, z = saved[0], saved[1], saved[2]
        N = weight.shape[0]
        P = ctx.P if cfg.bn is not None else z.numel() // N
        dgamma = dbeta = dbias = None
        fused_wgrad = False
        if cfg.bn is None:
            if cfg.slope is not None:
                dz = torch.empty_like(z)
                call('cy_act_bwd', _ptr(z), _ptr(da), _ptr(dz), float(cfg.slope), z.numel(), st)
            else:
                dz = da
            if ctx.has_bias:
                dbias = _empty((N,), z)
                call('cy_channel_sum', _ptr(dz), _ptr(dbias), P, N, st)
        else:
            if not ctx.bn_train:
                raise _lib.HipExtensionError('backward through an eval-mode BatchNorm block is not implemented')
            scale, shift, mean, invstd, gamma = saved[3:8]
            slope = 1.0 if cfg.slope is None else float(cfg.slope)
            if ctx.conv1_fused:
                da = da.contiguous()
                # the first block: both BatchNorm-backward passes recompute z tile by tile and read only da; dz is
                # formed in registers in the layout the weight-gradient MFMA consumes (csrc/conv1.hip)
                bias_t = saved[8] if ctx.has_bias else None
                B, _, Hi, Wi = x.shape
                redc = zero_pool.take((STATS_COPIES, N, 2), torch.float64, z.device)
                if USE_CONV1_ONEPASS and getattr(ctx, 'conv1_m2', None) is not None and _sync_world()[0] is None:
                    # one pass over da: sum d and the weight gradient OF d; the rest follows from the forward's patch moments
                    dW = _empty(tuple(weight.shape), z)
                    dbeta, dgamma = _empty((N,), z), _empty((N,), z)
                    ws = _empty((query('cy_conv1_bn_bwd_wgrad_ws_floats', B, Hi, Wi, N),), z)
                    with timer.range('conv1_bn_bwd_onepass/' + cfg.name):
                        call('cy_conv1_bn_bwd_onepass_bf16' if da_bf16 else 'cy_conv1_bn_bwd_onepass', _ptr(x), _ptr(weight),
                             _ptr(bias_t), _ptr(da), _ptr(scale), _ptr(shift), _ptr(mean), _ptr(invstd), slope,
                             _ptr(ctx.conv1_m2), _ptr(redc), _ptr(dW), _ptr(dgamma), _ptr(dbeta), None, _ptr(ws), B, Hi, Wi, N, st)
                    dbias = _const_zeros(N, z) if ctx.has_bias else None
                    return None, dW, dbias, dgamma, dbeta, None, None, None
                with timer.range('conv1_bn_bwd_reduce/' + cfg.name):
                    call('cy_conv1_bn_bwd_reduce_bf16' if da_bf16 else 'cy_conv1_bn_bwd_reduce', _ptr(x), _ptr(weight), _ptr(bias_t), _ptr(da), _ptr(scale), _ptr(shift),
                         _ptr(mean), _ptr(invstd), slope, _ptr(redc), B, Hi, Wi, N, st)
                red = _empty((N, 2), z, torch.float64)
                dbeta, dgamma = _empty((N,), z), _empty((N,), z)
                dist, world = _sync_world()
                # copies folded (and, data parallel, pre-scaled by 1 / world so that the all-reduce SUM is the mean)
                call('cy_bn_red_fold', _ptr(redc), STATS_COPIES, 1.0 / world, _ptr(red), _ptr(dgamma), _ptr(dbeta), N, st)
                if dist is not None:
                    dist.all_reduce(red)
                    call('cy_bn_red_fold', _ptr(red), 1, 1.0, None, _ptr(dgamma), _ptr(dbeta), N, st)
                dW = _empty(tuple(weight.shape), z)
                ws = _empty((query('cy_conv1_bn_bwd_wgrad_ws_floats', B, Hi, Wi, N),), z)
                with timer.range('conv1_bn_bwd_wgrad/' + cfg.name):
                    call('cy_conv1_bn_bwd_wgrad_bf16' if da_bf16 else 'cy_conv1_bn_bwd_wgrad', _ptr(x), _ptr(weight), _ptr(bias_t), _ptr(da), _ptr(scale), _ptr(shift),
                         _ptr(mean), _ptr(invstd), slope, _ptr(red), P, _ptr(dW), _ptr(ws), B, Hi, Wi, N, st)
                dbias = _const_zeros(N, z) if ctx.has_bias else None
                return None, dW, dbias, dgamma, dbeta, None, None, None
            premasked = False
            if ctx.holder is not None and ctx.holder.get('red') is not None:
                red = ctx.holder['red']        # summed by the consumer block's input-gradient epilogues
                ctx.holder['red'] = None
                # ... whose stride-2 Winograd kernel has then also applied the activation's derivative to da already
                premasked = bool(ctx.holder.pop('premasked', False))
                if premasked:
                    slope = 1.0
            else:
                red = _empty((N, 2), z, torch.float64)
                call('cy_bn_bwd_reduce', _ptr(z), _ptr(da), _ptr(scale), _ptr(shift), _ptr(mean), _ptr(invstd), slope,
                     _ptr(red), P, N, st)
            dist, world = _sync_world()
            if dist is not None:
                # global sums / world: the apply kernel divides by the LOCAL pixel count, which then gives the global
                # means; dgamma / dbeta come out as (global sum) / world, what the gradient all-reduce MEAN expects
                red = red.contiguous()
                call('cy_bn_red_fold', _ptr(red), 1, 1.0 / world, _ptr(red), None, None, N, st)    # in place: sum of the pre-scaled = mean
                dist.all_reduce(red)
            dz = torch.empty_like(z)
            dgamma, dbeta = _empty((N,), z), _empty((N,), z)
            if ctx.has_bias:
                # a bias in front of BatchNorm has an analytically zero gradient: sum(dz) == 0
                dbias = _const_zeros(N, z)
            if (FUSE_BN_BWD_APPLY and ctx.in_affine is None and not cfg.nchw_in
                    and _wino_wgrad_ok(cfg.k, cfg.stride, cfg.pad, x.shape[3], N)):
                # pass 2 of the BatchNorm backward inside the Winograd weight-gradient kernel, which has every dz element in
                # registers on its way to LDS anyway and writes it out for the input-gradient kernel
                red = red.contiguous()
                call('cy_bn_param_grad', _ptr(red), _ptr(dgamma), _ptr(dbeta), N, st)
                B_, Hi, Wi, Cin = x.shape
                dW = _empty(tuple(weight.shape), z)
                if premasked and _wino4_wgrad_ok(B_, Hi, Wi, Cin, N):
                    ws = _empty((query('cy_wino4_wgrad_ws_floats', B_, Hi, Wi, Cin, N),), z)
                    with timer.range('conv_wino4_wgrad_bn/' + cfg.name):
                        call('cy_conv3x3_winograd4_wgrad_bn', _ptr(_f32(x, 'conv input')), _ptr(z), _ptr(da), _ptr(dz), _ptr(scale),
                             _ptr(mean), _ptr(invstd), _ptr(red), P, _ptr(dW), _ptr(ws), B_, Hi, Wi, Cin, N, st)
                else:
                    ws = _empty((query('cy_wino_wgrad_ws_floats', B_, Cin, N),), z)
                    with timer.range('conv_wino_wgrad_bn/' + cfg.name):
                        call('cy_conv3x3_winograd_wgrad_bn', _ptr(_f32(x, 'conv input')), _ptr(z), _ptr(da), _ptr(dz), _ptr(scale),
                             _ptr(shift), _ptr(mean), _ptr(invstd), slope, 1 if premasked else 0, _ptr(red), P, _ptr(dW), _ptr(ws),
                             B_, Hi, Wi, Cin, N, st)
                fused_wgrad = True
            else:
                call('cy_bn_bwd_apply', _ptr(z), _ptr(da), _ptr(dz), _ptr(scale), _ptr(shift), _ptr(mean), _ptr(invstd),
                     _ptr(gamma), slope, _ptr(red), _ptr(dgamma), _ptr(dbeta), P, N, st)
        if not fused_wgrad:
            dW = conv_wgrad(x, dz, cfg.k, cfg.stride, cfg.pad, cfg.nchw_in, cfg.name, ctx.in_affine)
        dx = None
        if ctx.needs_input_grad[0]:
            if cfg.nchw_in:
                raise _lib.HipExtensionError('input gradient of an NCHW-input convolution is not implemented')
            fuse, h = None, cfg.in_holder
            if (FUSE_BN_BWD_REDUCE and ctx.in_affine is not None and h is not None and x.shape[3] % 4 == 0
                    and not _winograd_ok(cfg.k, cfg.stride, cfg.pad, N, False)):
                bred = zero_pool.take((STATS_COPIES, x.shape[3], 2), torch.float64, x.device)
                fuse = (x, ctx.in_affine[0], ctx.in_affine[1], h['mean'], h['invstd'], ctx.in_affine[2], bred)
            info = {}
            dx = conv_dgrad(dz, weight, tuple(x.shape), cfg.k, cfg.stride, cfg.pad, cfg.name, fuse, info)
            if fuse is not None:
                h['premasked'] = bool(info.get('premasked', False))
                h['red'] = _empty((x.shape[3], 2), x, torch.float64)
                call('cy_bn_red_fold', _ptr(bred), STATS_COPIES, 1.0, _ptr(h['red']), None, None, x.shape[3], st)
        return dx, dW, dbias, dgamma, dbeta, None, None, None


# ------------------------------------------------------------------------------------------------ bf16 convolution path
def _bf(t, name='tensor'):
    if not (t.is_cuda and t.dtype == torch.bfloat16):
        raise _lib.HipExtensionError('%s must be a bfloat16 tensor on the GPU (got %s on %s)' % (name, t.dtype, t.device))
    return t if t.is_contiguous() else t.contiguous()


def bf16_layer_ok(k, stride, pad, cin, cout):
    """The layers the bf16 kernels are built for: the backbone's 3x3/s1/p1 and 4x4/s2/p1 convs (models.py:350-363)."""
    return pad == 1 and ((k == 3 and stride == 1 and cout % 128 == 0) or (k == 4 and stride == 2)) and cin % 64 == 0 and cout % 64 == 0


def cast_bf16(x):
    """fp32 -> bf16 (round to nearest even); the gradient passes through unchanged (it arrives in fp32)."""
    return _CastBF16.apply(x)


class _CastBF16(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        x = _f32(x)
        y = torch.empty(x.shape, dtype=torch.bfloat16, device=x.device)
        call('cy_cast_f32_bf16', _ptr(x), _ptr(y), x.numel(), _stream())
        return y

    @staticmethod
    def backward(ctx, dy):
        if dy.dtype == torch.bfloat16:
            out = torch.empty(dy.shape, dtype=torch.float32, device=dy.device)
            call('cy_cast_bf16_f32', _ptr(_bf(dy)), _ptr(out), dy.numel(), _stream())
            return out
        return dy


def conv_forward_bf16(x, weight, bias, k, stride, pad, stats=None, tag='conv', lrelu=None, out_f32=False):
    """z (bf16 NHWC) = conv2d(x bf16 NHWC) + bias with fp32 accumulation; optional BatchNorm statistics.
    lrelu = slope in [0, 1]: LeakyReLU epilogue (eval forward, BatchNorm folded into weight / bias); out_f32: fp32 output."""
    x, weight = _bf(x, 'conv input'), _f32(weight, 'conv weight')
    B, Hi, Wi, Cin = x.shape
    Cout = weight.shape[0]
    Ho, Wo = (Hi + 2 * pad - k) // stride + 1, (Wi + 2 * pad - k) // stride + 1
    st = _stream()
    wp = torch.empty((query('cy_conv_bf16_packed_elems', k * k * Cin, Cout),), dtype=torch.bfloat16, device=x.device)
    call('cy_conv_bf16_pack_weights', _ptr(weight), _ptr(wp), Cout, Cin, k, k, k, k, 0, 0, 1, 0, st)
    z = torch.empty((B, Ho, Wo, Cout), dtype=torch.float32 if out_f32 else torch.bfloat16, device=x.device)
    a = ConvGemm(X=x.data_ptr(), Wp=wp.data_ptr(), Y=z.data_ptr(), bias=bias.data_ptr() if bias is not None else None,
                 stats=stats.data_ptr() if stats is not None else None,
                 xs_b=Hi * Wi * Cin, xs_y=Wi * Cin, xs_x=Cin, xs_c=1, B=B, Hi=Hi, Wi=Wi, Cin=Cin, Ho=Ho, Wo=Wo, N=Cout,
                 TH=k, TW=k, in_stride=stride, dy0=-pad, dx0=-pad, dstep=1, Hy=Ho, Wy=Wo, out_stride=1, out_oy=0, out_ox=0,
                 act=0 if lrelu is None else 2, act_slope=1.0 if lrelu is None else float(lrelu))
    with timer.range('conv_bf16_fwd/' + tag):
        call('cy_conv_gemm_bf16', C.byref(a), 1 if out_f32 else 0, st)
    return z


BF16_DGRAD_ONE_LAUNCH = True       # bf16 path: the output-parity classes of a strided input gradient in one launch (cy_conv_gemm_bf16_classes)
FUSE_BN_BWD_REDUCE_BF16 = True     # bf16 path: the producer block's BatchNorm-backward sums out of the consumer's input-gradient epilogue


def conv_dgrad_bf16(dz, weight, in_shape, k, stride, pad, out_f32=False, tag='conv', bn_fuse=None):
    """dx [B,Hi,Wi,Cin] (bf16, or fp32 for an fp32 producer) from dz (bf16): one GEMM per output-parity class.
    bn_fuse = (z bf16, scale, shift, mean, invstd, slope, red[STATS_COPIES][Cin][2]) of the PRODUCER block (bf16 output only): the
    epilogues store d = dx * lrelu'(z * scale + shift) -- the premasked gradient -- and add sum d, sum d * xhat of the stored
    (bf16-rounded) values to red, all parity classes together (cy_conv_gemm_t.bn_*)."""
    dz, weight = _bf(dz, 'grad'), _f32(weight, 'conv weight')
    if bn_fuse is not None and out_f32:
        raise _lib.HipExtensionError('conv_dgrad_bf16: the fused BatchNorm-backward sums are built for the bf16 output')
    B, Hi, Wi, Cin = in_shape
    _, Ho, Wo, Cout = dz.shape
    st = _stream()
    dx = torch.empty((B, Hi, Wi, Cin), dtype=torch.float32 if out_f32 else torch.bfloat16, device=dz.device)
    classes = dgrad_classes(Hi, Wi, k, stride, pad)
    nel = query('cy_conv_bf16_packed_elems', ((k + stride - 1) // stride) ** 2 * Cout, Cin)
    wp = torch.empty((len(classes), nel), dtype=torch.bfloat16, device=dz.device)
    descs = (ConvGemm * len(classes))()
    for i, c in enumerate(classes):
        call('cy_conv_bf16_pack_weights', _ptr(weight), _ptr(wp[i]), Cout, Cin, k, k, c['TH'], c['TW'], c['kh0'], c['kw0'],
             c['kstep'], 1, st)
        a = ConvGemm(X=dz.data_ptr(), Wp=wp[i].data_ptr(), Y=dx.data_ptr(), bias=None, stats=None,
                     xs_b=Ho * Wo * Cout, xs_y=Wo * Cout, xs_x=Cout, xs_c=1, B=B, Hi=Ho, Wi=Wo, Cin=Cout,
                     Ho=c['Ho'], Wo=c['Wo'], N=Cin, TH=c['TH'], TW=c['TW'], in_stride=1, dy0=c['dy0'], dx0=c['dx0'],
                     dstep=c['dstep'], Hy=Hi, Wy=Wi, out_stride=c['out_stride'], out_oy=c['out_oy'], out_ox=c['out_ox'], act=0)
        if bn_fuse is not None:
            bz, bsc, bsh, bmu, bis, bsl, bred = bn_fuse
            a.bn_z, a.bn_scale, a.bn_shift = _bf(bz, 'producer z').data_ptr(), bsc.data_ptr(), bsh.data_ptr()
            a.bn_mean, a.bn_invstd, a.bn_red, a.bn_slope = bmu.data_ptr(), bis.data_ptr(), bred.data_ptr(), float(bsl)
        descs[i] = a
    same = all((c['TH'], c['TW'], c['Ho'], c['Wo'], c['dstep'], c['out_stride']) ==
               (classes[0]['TH'], classes[0]['TW'], classes[0]['Ho'], classes[0]['Wo'], classes[0]['dstep'], classes[0]['out_stride'])
               for c in classes)
    if BF16_DGRAD_ONE_LAUNCH and 1 < len(classes) <= 4 and same:
        # the parity classes of a strided layer in ONE launch (even sizes: all classes have the same grid and taps)
        with timer.range('conv_bf16_dgrad/' + tag):
            call('cy_conv_gemm_bf16_classes', descs, len(classes), 1 if out_f32 else 0, st)
    else:
        for i in range(len(classes)):
            with timer.range('conv_bf16_dgrad/' + tag):
                call('cy_conv_gemm_bf16', C.byref(descs[i]), 1 if out_f32 else 0, st)
    return dx


def conv_wgrad_bf16(x, dz, k, stride, pad, tag='conv'):
    x, dz = _bf(x, 'conv input'), _bf(dz, 'grad')
    B, Hi, Wi, Cin = x.shape
    _, Ho, Wo, Cout = dz.shape
    nws = query('cy_conv_wgrad_bf16_ws_floats', B, Ho, Wo, Cin, Cout, k, stride)
    if nws < 0 or pad != 1:
        raise _lib.HipExtensionError('bf16 weight gradient: unsupported layer k=%d s=%d p=%d Cin=%d Cout=%d' % (k, stride, pad, Cin, Cout))
    ws = _empty((nws,), dz)
    dW = _empty((Cout, Cin, k, k), dz)
    with timer.range('conv_bf16_wgrad/' + tag):
        call('cy_conv_wgrad_bf16', _ptr(x), _ptr(dz), _ptr(dW), _ptr(ws), B, Hi, Wi, Cin, Ho, Wo, Cout, k, stride, _stream())
    return dW


class _ConvBlockBF16(torch.autograd.Function):
    """conv -> BatchNorm (batch statistics from the conv epilogue) -> LeakyReLU on the bf16 kernels: x, the raw conv output
    z and the activation are bf16 NHWC; statistics, scale / shift and all parameter gradients are fp32 / double.
    cfg.out_f32: the activation leaves in fp32 (its consumer is an fp32 kernel: the routing head); cfg.in_f32: x was
    cast from an fp32 producer, whose backward wants its gradient in fp32."""

    @staticmethod
    def forward(ctx, x, weight, bias, gamma, beta, cfg):
        st = _stream()
        if getattr(cfg, 'in_f32', False):
            # the producer is an fp32 kernel (the first block): round its activation to bf16 HERE, so that autograd sees an
            # fp32 input and takes the fp32 gradient this block's input-gradient kernel writes (a separate cast Function made
            # the engine convert that gradient to bf16 and back: two 3 GB passes at 608 x 608)
            xf = _f32(x, 'conv input')
            x = torch.empty(xf.shape, dtype=torch.bfloat16, device=xf.device)
            call('cy_cast_f32_bf16', _ptr(xf), _ptr(x), xf.numel(), st)
        x, weight = _bf(x, 'conv input'), _f32(weight, 'conv weight')
        N = weight.shape[0]
        bn = cfg.bn
        if bn is None or cfg.slope is None:
            raise _lib.HipExtensionError('the bf16 path is built for conv -> BatchNorm -> LeakyReLU blocks')
        ctx.cfg, ctx.has_bias = cfg, bias is not None
        scale, shift, mean, invstd = (_empty((N,), weight) for _ in range(4))
        if bn.training:
            stats = zero_pool.take((STATS_COPIES, N, 2), torch.float64, x.device)
            z = conv_forward_bf16(x, weight, bias, cfg.k, cfg.stride, cfg.pad, stats, cfg.name)
            P = z.numel() // N
            dist, world = _sync_world()
            if dist is not None:
                dist.all_reduce(stats)
            call('cy_bn_finalize', _ptr(stats), P * world, _ptr(gamma), _ptr(beta), _ptr(bn.running_mean), _ptr(bn.running_var),
                 float(bn.momentum), float(bn.eps), _ptr(scale), _ptr(shift), _ptr(mean), _ptr(invstd), N,
                 _ptr(bn.num_batches_tracked), st)
            _bump_param_epoch()
        else:
            if FOLD_EVAL_BN and 0.0 <= float(cfg.slope) <= 1.0:     # eval mode: ONE launch, BatchNorm folded into weights / bias
                Wf, bf = fold_eval_bn(weight, bias, gamma, beta, bn)
                ctx.bn_train = False
                ctx.save_for_backward()
                return conv_forward_bf16(x, Wf, bf, cfg.k, cfg.stride, cfg.pad, None, cfg.name, lrelu=float(cfg.slope),
                                         out_f32=bool(getattr(cfg, 'out_f32', False)))
            z = conv_forward_bf16(x, weight, bias, cfg.k, cfg.stride, cfg.pad, None, cfg.name)
            P = z.numel() // N
            call('cy_bn_eval_scale_shift', _ptr(gamma), _ptr(beta), _ptr(bn.running_mean), _ptr(bn.running_var), float(bn.eps),
                 _ptr(scale), _ptr(shift), N, st)
        ctx.bn_train, ctx.P = bool(bn.training), P
        out_f32 = bool(getattr(cfg, 'out_f32', False))
        out = torch.empty(z.shape, dtype=torch.float32 if out_f32 else torch.bfloat16, device=z.device)
        call('cy_affine_act_bf16', _ptr(z), _ptr(out), _ptr(scale), _ptr(shift), float(cfg.slope), P, N, 1 if out_f32 else 0, st)
        ctx.save_for_backward(x, weight, z, scale, shift, mean, invstd)
        h = getattr(cfg, 'out_holder', None)
        if h is not None:
            # what the CONSUMER block's input-gradient epilogue needs for this block's BatchNorm-backward sums (bf16 output only)
            h.clear()
            if bn.training and not out_f32:
                h.update(z=z, scale=scale, shift=shift, mean=mean, invstd=invstd, slope=float(cfg.slope), red=None)
        return out

    @staticmethod
    def backward(ctx, da):
        cfg = ctx.cfg
        if not ctx.bn_train:
            raise _lib.HipExtensionError('backward through an eval-mode BatchNorm block is not implemented')
        x, weight, z, scale, shift, mean, invstd = ctx.saved_tensors
        st = _stream()
        N, P = weight.shape[0], ctx.P
        da_f32 = da.dtype == torch.float32
        da = _f32(da, 'grad') if da_f32 else _bf(da, 'grad')
        slope = float(cfg.slope)
        ho = getattr(cfg, 'out_holder', None)
        if ho is not None and ho.get('red') is not None and not da_f32:
            red = ho['red']                    # summed by the consumer block's input-gradient epilogues, which also stored da premasked
            ho['red'] = None
            slope = 1.0
        else:
            red = _empty((N, 2), weight, torch.float64)
            call('cy_bn_bwd_reduce_bf16', _ptr(z), _ptr(da), 1 if da_f32 else 0, _ptr(scale), _ptr(shift), _ptr(mean), _ptr(invstd),
                 slope, _ptr(red), P, N, st)
        dist, world = _sync_world()
        if dist is not None:
            call('cy_bn_red_fold', _ptr(red), 1, 1.0 / world, _ptr(red), None, None, N, st)
            dist.all_reduce(red)
        dz = torch.empty_like(z)
        dgamma, dbeta = _empty((N,), weight), _empty((N,), weight)
        Bx, Hx, Wx, Cx = x.shape
        nws = -1
        if FUSE_BN_BWD_APPLY_BF16 and slope == 1.0 and not da_f32 and cfg.pad == 1:
            # premasked gradient: BatchNorm-backward pass 2 inside the weight gradient's loader, dz written for the input gradient
            nws = query('cy_conv_wgrad_bf16_bn_ws_floats', Bx, z.shape[1], z.shape[2], Cx, N, cfg.k, cfg.stride)
        if nws >= 0:
            ws = _empty((nws,), weight)
            dW = _empty((N, Cx, cfg.k, cfg.k), weight)
            with timer.range('conv_bf16_wgrad_bn/' + cfg.name):
                call('cy_conv_wgrad_bf16_bn', _ptr(x), _ptr(da.contiguous()), _ptr(z), _ptr(dz), _ptr(dW), _ptr(ws), _ptr(scale), _ptr(mean),
                     _ptr(invstd), _ptr(red), _ptr(dgamma), _ptr(dbeta), Bx, Hx, Wx, Cx, z.shape[1], z.shape[2], N, cfg.k, cfg.stride, st)
        else:
            call('cy_bn_bwd_apply_bf16', _ptr(z), _ptr(da), 1 if da_f32 else 0, _ptr(dz), _ptr(scale), _ptr(shift), _ptr(mean),
                 _ptr(invstd), slope, _ptr(red), _ptr(dgamma), _ptr(dbeta), P, N, st)
            dW = conv_wgrad_bf16(x, dz, cfg.k, cfg.stride, cfg.pad, cfg.name)
        dx = None
        if ctx.needs_input_grad[0]:
            in_f32 = bool(getattr(cfg, 'in_f32', False))
            hi = getattr(cfg, 'in_holder', None)
            fuse = None
            if (FUSE_BN_BWD_REDUCE_BF16 and hi is not None and hi.get('z') is not None and not in_f32
                    and tuple(hi['z'].shape) == tuple(x.shape)):
                Cin = x.shape[3]
                bred = zero_pool.take((STATS_COPIES, Cin, 2), torch.float64, x.device)
                fuse = (hi['z'], hi['scale'], hi['shift'], hi['mean'], hi['invstd'], hi['slope'], bred)
            dx = conv_dgrad_bf16(dz, weight, tuple(x.shape), cfg.k, cfg.stride, cfg.pad, in_f32, cfg.name, fuse)
            if fuse is not None:
                hi['red'] = _empty((x.shape[3], 2), weight, torch.float64)
                call('cy_bn_red_fold', _ptr(fuse[6]), STATS_COPIES, 1.0, _ptr(hi['red']), None, None, x.shape[3], st)
        dbias = _const_zeros(N, weight) if ctx.has_bias else None       # in front of BatchNorm: analytically zero
        return dx, dW, dbias, dgamma, dbeta, None


def conv_block_bf16(x, weight, bias, gamma, beta, cfg):
    return _ConvBlockBF16.apply(x, weight, bias, gamma, beta, cfg)


def conv_block(x, weight, bias, gamma, beta, cfg, in_scale=None, in_shift=None):
    """One conv (+BN) (+activation) block.  cfg.defer_act: returns (z, scale, shift) instead of the activation; the next
    block then runs with cfg.in_slope set and these tensors as in_scale / in_shift (its input gradient is the gradient
    with respect to the ACTIVATED value, which is what this block's backward expects)."""
    return _ConvBlock.apply(x, weight, bias, gamma, beta, cfg, in_scale, in_shift)


# ------------------------------------------------------------------------------------------------ routing
class _Routing(torch.autograd.Function):
    @staticmethod
    def forward(ctx, u, W, n_iter, gather_g, gather_B):
        u, W = _f32(u, 'capsule input'), _f32(W, 'route_weights')
        _, N, Cc, Din, Dout = W.shape
        if gather_g:
            R = gather_g * gather_g * gather_B
            out_shape = (gather_B, gather_g, gather_g, Cc, Dout)
        else:
            R = u.shape[0]
            out_shape = (R, Cc, Dout)
        v = _empty(out_shape, u)
        s_hist = _empty((n_iter, R, Cc, Dout), u)
        a = RoutingFwd(u=u.data_ptr(), W=W.data_ptr(), v_out=v.data_ptr(), s_hist=s_hist.data_ptr(), R=R, N=N, C=Cc,
                       Din=Din, Dout=Dout, n_iter=n_iter, gather_g=gather_g, gather_B=gather_B, ws=None)
        nws = query('cy_routing_fwd_ws_floats', C.byref(a))
        if nws:
            ws = _empty((nws,), u)
            a.ws = ws.data_ptr()
        with timer.range('routing_fwd'):
            call('cy_routing_fwd', C.byref(a), _stream())
        ctx.save_for_backward(u, W, s_hist)
        ctx.dims = (R, N, Cc, Din, Dout, n_iter, gather_g, gather_B)
        return v

    @staticmethod
    def backward(ctx, dv):
        u, W, s_hist = ctx.saved_tensors
        R, N, Cc, Din, Dout, n_iter, g, gB = ctx.dims
        dv = _f32(dv, 'grad')
        du, dW = torch.empty_like(u), torch.empty_like(W)
        a = RoutingBwd(u=u.data_ptr(), W=W.data_ptr(), s_hist=s_hist.data_ptr(), dv=dv.data_ptr(), du=du.data_ptr(),
                       dW=dW.data_ptr(), ws=None, R=R, N=N, C=Cc, Din=Din, Dout=Dout, n_iter=n_iter, gather_g=g,
                       gather_B=gB)
        ws = _empty((query('cy_routing_bwd_ws_floats', C.byref(a)),), u)
        a.ws = ws.data_ptr()
        with timer.range('routing_bwd'):
            call('cy_routing_bwd', C.byref(a), _stream())
        return du, dW, None, None, None


def routing(u, W, n_iter=3, gather_g=0, gather_B=0):
    """u [R,N,Din] (or the NHWC feature map [B,4g,4g,256] with gather_g=g), W [1,N,C,Din,Dout] -> v."""
    return _Routing.apply(u, W, int(n_iter), int(gather_g), int(gather_B))


# ------------------------------------------------------------------------------------------------ small vector ops
class _Rows(torch.autograd.Function):
    """squash / length over the last dimension."""

    @staticmethod
    def forward(ctx, x, kind):
        x = _f32(x)
        D = x.shape[-1]
        rows = x.numel() // D
        st = _stream()
        ctx.kind = kind
        if kind == 'squash':
            y = torch.empty_like(x)
            call('cy_squash_fwd', _ptr(x), _ptr(y), rows, D, st)
            ctx.save_for_backward(x)
        else:
            y = _empty(x.shape[:-1], x)
            call('cy_length_fwd', _ptr(x), _ptr(y), rows, D, st)
            ctx.save_for_backward(x, y)
        return y

    @staticmethod
    def backward(ctx, dy):
        dy = _f32(dy)
        st = _stream()
        x = ctx.saved_tensors[0]
        D = x.shape[-1]
        rows = x.numel() // D
        dx = torch.empty_like(x)
        if ctx.kind == 'squash':
            call('cy_squash_bwd', _ptr(x), _ptr(dy), _ptr(dx), rows, D, st)
        else:
            call('cy_length_bwd', _ptr(x), _ptr(ctx.saved_tensors[1]), _ptr(dy), _ptr(dx), rows, D, st)
        return dx, None


def squash(x):
    return _Rows.apply(x, 'squash')


def length(x):
    return _Rows.apply(x, 'length')


class _Permute4(torch.autograd.Function):
    """out[b][i1][i2][i3] = in[b*sb + i1*s1 + i2*s2 + i3*s3]; backward is the inverse scatter."""

    @staticmethod
    def forward(ctx, x, nb, dims, strides):
        x = _f32(x)
        out = _empty((nb,) + tuple(dims), x)
        call('cy_permute4', _ptr(x), _ptr(out), nb, dims[0], dims[1], dims[2], strides[0], strides[1], strides[2],
             strides[3], 0, _stream())
        ctx.args = (nb, dims, strides, tuple(x.shape))
        return out

    @staticmethod
    def backward(ctx, dout):
        nb, dims, strides, xshape = ctx.args
        dout = _f32(dout)
        dx = _empty(xshape, dout)
        call('cy_permute4', _ptr(dout), _ptr(dx), nb, dims[0], dims[1], dims[2], strides[0], strides[1], strides[2],
             strides[3], 1, _stream())
        return dx, None, None, None


def nchw_to_nhwc(x):
    B, Cc, H, W = x.shape
    return _Permute4.apply(x, B, (H, W, Cc), (Cc * H * W, W, 1, H * W))


def nhwc_to_nchw(x):
    B, H, W, Cc = x.shape
    return _Permute4.apply(x, B, (Cc, H, W), (H * W * Cc, 1, W * Cc, Cc))


def primary_caps_rows(z, n_caps):
    """conv output [B,h,w,(o,cap)] -> capsule rows [B, o*h*w + y*w + x, cap] (models.py:81: view(B,-1,1) + cat)."""
    B, H, W, Cc = z.shape
    O = Cc // n_caps
    out = _Permute4.apply(z, B, (O, H * W, n_caps), (H * W * Cc, n_caps, Cc, 1))
    return out.view(B, O * H * W, n_caps)


class _MaxPool2(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        x = _f32(x)
        B, H, W, Cc = x.shape
        y = _empty((B, H // 2, W // 2, Cc), x)
        idx = _empty((B, H // 2, W // 2, Cc), x, torch.uint8)
        call('cy_maxpool2_fwd', _ptr(x), _ptr(y), _ptr(idx), B, H // 2, W // 2, Cc, _stream())
        ctx.save_for_backward(idx)
        ctx.shape = (B, H, W, Cc)
        return y

    @staticmethod
    def backward(ctx, dy):
        (idx,) = ctx.saved_tensors
        B, H, W, Cc = ctx.shape
        dy = _f32(dy)
        dx = _empty((B, H, W, Cc), dy) if (H % 2 == 0 and W % 2 == 0) else torch.zeros((B, H, W, Cc), device=dy.device)
        call('cy_maxpool2_bwd', _ptr(dy), _ptr(idx), _ptr(dx), B, H // 2, W // 2, Cc, _stream())
        return dx


def maxpool2(x):
    return _MaxPool2.apply(x)


FUSE_POOL = True    # conv -> BatchNorm -> LeakyReLU -> MaxPool blocks (DarkNet): activation + pooling in one pass over z, pooling backward +
                    # the block's BatchNorm-backward sums in one pass (the full-resolution activation and its gradient are never stored twice)


def pool_fusable(C, H, W, slope):
    """The fused BatchNorm-apply + LeakyReLU + 2x2 max-pool pass takes this block's output [B,H,W,C]."""
    return (FUSE_POOL and H % 2 == 0 and W % 2 == 0 and C % 4 == 0 and (C % 64 == 0 or C in (4, 8, 16, 32)) and slope is not None
            and 0.0 < slope <= 1.0)


class _AffineActMaxPool(torch.autograd.Function):
    """y = maxpool2(lrelu(z * scale + shift)) of a block that deferred its activation (ConvBlockCfg.defer_act); plays the CONSUMER of the
    producer's hand-over dict: its backward returns the premasked gradient d = [pos == argmax] dy lrelu'(.) as the gradient of z and leaves
    the producer's BatchNorm-backward sums in holder['red'] (models.py:135 ... 195: nn.LeakyReLU + nn.MaxPool2d of the DarkNet blocks)."""

    @staticmethod
    def forward(ctx, z, scale, shift, slope, holder):
        z = _f32(z, 'conv output')
        B, H, W, Cc = z.shape
        y = _empty((B, H // 2, W // 2, Cc), z)
        idx = torch.empty((B, H // 2, W // 2, Cc), dtype=torch.uint8, device=z.device)
        with timer.range('affine_act_maxpool'):
            call('cy_affine_act_maxpool2', _ptr(z), _ptr(scale), _ptr(shift), float(slope), _ptr(y), _ptr(idx), B, H // 2, W // 2, Cc, _stream())
        ctx.save_for_backward(z, scale, shift, idx)
        ctx.slope, ctx.holder = float(slope), holder
        return y

    @staticmethod
    def backward(ctx, dy):
        z, scale, shift, idx = ctx.saved_tensors
        B, H, W, Cc = z.shape
        dy = _f32(dy, 'pooled gradient')
        h = ctx.holder
        d = torch.empty_like(z)
        red = zero_pool.take((Cc, 2), torch.float64, z.device)
        with timer.range('maxpool_bwd_bn'):
            call('cy_maxpool2_bwd_bn', _ptr(dy), _ptr(idx), _ptr(z), _ptr(scale), _ptr(shift), _ptr(h['mean']), _ptr(h['invstd']), ctx.slope,
                 _ptr(d), _ptr(red), B, H // 2, W // 2, Cc, _stream())
        h['red'], h['premasked'] = red, True
        return d, None, None, None, None


def affine_act_maxpool(z, scale, shift, slope, holder):
    return _AffineActMaxPool.apply(z, scale, shift, slope, holder)


class _Upsample(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, f):
        x = _f32(x)
        B, H, W, Cc = x.shape
        y = _empty((B, H * f, W * f, Cc), x)
        call('cy_upsample_fwd', _ptr(x), _ptr(y), B, H, W, Cc, f, _stream())
        ctx.args = (B, H, W, Cc, f)
        return y

    @staticmethod
    def backward(ctx, dy):
        B, H, W, Cc, f = ctx.args
        dy = _f32(dy)
        dx = _empty((B, H, W, Cc), dy)
        call('cy_upsample_bwd', _ptr(dy), _ptr(dx), B, H, W, Cc, f, _stream())
        return dx, None


def upsample_nearest(x, f):
    return _Upsample.apply(x, int(f))


class _Tanh(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        x = _f32(x)
        y = torch.empty_like(x)
        call('cy_tanh_fwd', _ptr(x), _ptr(y), x.numel(), _stream())
        ctx.save_for_backward(y)
        return y

    @staticmethod
    def backward(ctx, dy):
        (y,) = ctx.saved_tensors
        dy = _f32(dy)
        dx = torch.empty_like(y)
        call('cy_tanh_bwd', _ptr(y), _ptr(dy), _ptr(dx), y.numel(), _stream())
        return dx


def tanh(x):
    return _Tanh.apply(x)


class _YoloHead(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, split, n_classes):
        x = _f32(x)
        y = torch.empty_like(x)
        cells = x.numel() // (split + n_classes)
        call('cy_yolo_head_fwd', _ptr(x), _ptr(y), cells, split, n_classes, _stream())
        ctx.save_for_backward(y)
        ctx.args = (cells, split, n_classes)
        return y

    @staticmethod
    def backward(ctx, dy):
        (y,) = ctx.saved_tensors
        cells, split, n_classes = ctx.args
        dy = _f32(dy)
        dx = torch.empty_like(y)
        call('cy_yolo_head_bwd', _ptr(y), _ptr(dy), _ptr(dx), cells, split, n_classes, _stream())
        return dx, None, None


def yolo_head(x, split, n_classes):
    return _YoloHead.apply(x, int(split), int(n_classes))


class _PickCapsule(torch.autograd.Function):
    @staticmethod
    def forward(ctx, caps, y):
        caps = _f32(caps)
        B, Cc, D = caps.shape
        y = y.to(torch.int64).contiguous()
        out = _empty((B, D), caps)
        call('cy_pick_capsule', _ptr(caps), _ptr(y), _ptr(out), B, Cc, D, 0, _stream())
        ctx.save_for_backward(y)
        ctx.shape = (B, Cc, D)
        return out

    @staticmethod
    def backward(ctx, dout):
        (y,) = ctx.saved_tensors
        B, Cc, D = ctx.shape
        dout = _f32(dout)
        dcaps = _empty((B, Cc, D), dout)
        call('cy_pick_capsule', _ptr(dout), _ptr(y), _ptr(dcaps), B, Cc, D, 1, _stream())
        return dcaps, None


def pick_capsule(caps, y):
    return _PickCapsule.apply(caps, y)


# ------------------------------------------------------------------------------------------------ losses
def _scaled(grad, gout):
    out = torch.empty_like(grad)
    gout = gout.to(torch.float32).contiguous()
    call('cy_scale_by_device_scalar', _ptr(grad), _ptr(gout), _ptr(out), grad.numel(), _stream())
    return out


class _DarkCapsuleLoss(torch.autograd.Function):
    @staticmethod
    def forward(ctx, caps, y):
        caps = _f32(caps, 'caps')
        if not (y.is_cuda and y.dtype == torch.float64):
            y = y.to(device=caps.device, dtype=torch.float64)
        y = y.contiguous()
        B = caps.shape[0]
        cells = caps.numel() // (5 * B)
        loss, dcaps = _empty((), caps), torch.empty_like(caps)
        call('cy_darkcapsule_loss', _ptr(caps), _ptr(y), y.shape[-1], _ptr(loss), _ptr(dcaps), B, cells, _stream())
        ctx.save_for_backward(dcaps)
        return loss

    @staticmethod
    def backward(ctx, gout):
        return _scaled(ctx.saved_tensors[0], gout), None


class _DarkCapsule23Loss(torch.autograd.Function):
    """darkcapsule2_loss (caps [B,g,g,5+C]) / darkcapsule3_loss (caps [B,g,g,C,D]); loss_fns.py:145-184."""

    @staticmethod
    def forward(ctx, caps, y, variant):
        caps = _f32(caps, 'caps')
        if not (y.is_cuda and y.dtype == torch.float64):
            y = y.to(device=caps.device, dtype=torch.float64)
        y = y.contiguous()
        B = caps.shape[0]
        Cc = y.shape[-1] - 5
        cells = y.numel() // (y.shape[-1] * B)
        loss, dcaps = _empty((), caps), torch.empty_like(caps)
        if variant == 2:
            if caps.shape[-1] != 5 + Cc:
                raise _lib.HipExtensionError('darkcapsule2_loss: caps last dim %d != 5 + n_classes %d' % (caps.shape[-1], Cc))
            call('cy_darkcapsule2_loss', _ptr(caps), _ptr(y), _ptr(loss), _ptr(dcaps), B, cells, Cc, _stream())
        else:
            if caps.dim() != 5 or caps.shape[3] != Cc:
                raise _lib.HipExtensionError('darkcapsule3_loss: caps must be [B,g,g,n_classes,D]; got %s' % (tuple(caps.shape),))
            call('cy_darkcapsule3_loss', _ptr(caps), _ptr(y), _ptr(loss), _ptr(dcaps), B, cells, Cc, caps.shape[4], _stream())
        ctx.save_for_backward(dcaps)
        return loss

    @staticmethod
    def backward(ctx, gout):
        return _scaled(ctx.saved_tensors[0], gout), None, None


def darkcapsule2_loss_fn(caps, y):
    return _DarkCapsule23Loss.apply(caps, y, 2)


def darkcapsule3_loss_fn(caps, y):
    return _DarkCapsule23Loss.apply(caps, y, 3)


class _CapsuleLoss(torch.autograd.Function):
    """margin loss (+ recon_coef * sum (x - recon)^2), all divided by B."""

    @staticmethod
    def forward(ctx, scores, y, x, recon, coef):
        scores = _f32(scores, 'scores')
        B, Cc = scores.shape
        y = y.to(torch.int64).contiguous()
        st = _stream()
        loss, dscores = _empty((), scores), torch.empty_like(scores)
        call('cy_margin_loss', _ptr(scores), _ptr(y), _ptr(loss), _ptr(dscores), B, Cc, st)
        if recon is not None:
            x, recon = _f32(x, 'x'), _f32(recon, 'recon')
            drecon = torch.empty_like(recon)
            call('cy_recon_loss_add', _ptr(x), _ptr(recon), float(coef) / B, _ptr(loss), _ptr(drecon), recon.numel(), st)
            ctx.save_for_backward(dscores, drecon)
        else:
            ctx.save_for_backward(dscores)
        return loss

    @staticmethod
    def backward(ctx, gout):
        saved = ctx.saved_tensors
        drecon = _scaled(saved[1], gout) if len(saved) > 1 else None
        return _scaled(saved[0], gout), None, None, drecon, None


class _DarkLoss(torch.autograd.Function):
    @staticmethod
    def forward(ctx, y_pred, y_true, nb, n_classes, l_coord, l_noobj, img):
        y_pred = _f32(y_pred, 'y_pred')
        if not (y_true.is_cuda and y_true.dtype == torch.float64):
            y_true = y_true.to(device=y_pred.device, dtype=torch.float64)
        y_true = y_true.contiguous()
        B, g = y_pred.shape[0], y_pred.shape[1]
        loss, avg_iou = _empty((), y_pred), _empty((), y_pred)
        dpred = torch.empty_like(y_pred)
        call('cy_dark_loss', _ptr(y_pred), _ptr(y_true), _ptr(loss), _ptr(avg_iou), _ptr(dpred), B, g, nb, n_classes,
             float(l_coord), float(l_noobj), float(img), _stream())
        ctx.save_for_backward(dpred)
        ctx.mark_non_differentiable(avg_iou)
        return loss, avg_iou

    @staticmethod
    def backward(ctx, gout, _g_iou):
        return _scaled(ctx.saved_tensors[0], gout), None, None, None, None, None, None


def darkcapsule_loss_fn(caps, y):
    return _DarkCapsuleLoss.apply(caps, y)


def capsule_loss_fn(scores, y, x=None, recon=None, coef=0.0):
    return _CapsuleLoss.apply(scores, y, x, recon, coef)


def dark_loss_fn(y_pred, y_true, nb, n_classes, l_coord, l_noobj, img):
    return _DarkLoss.apply(y_pred, y_true, nb, n_classes, l_coord, l_noobj, img)
