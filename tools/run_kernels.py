"""Run the hot kernels alone (for rocprofv3 --pmc passes and quick A/B timing).
usage: python3 tools/run_kernels.py [conv2|routing|all] [B] [reps]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import capsyolo_amd
from capsyolo_amd import ops

what = sys.argv[1] if len(sys.argv) > 1 else 'all'
B = int(sys.argv[2]) if len(sys.argv) > 2 else 32
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
dev = torch.device('cuda:0')
torch.manual_seed(0)


def timeit(name, fn, flops=None, nbytes=None):
    fn(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        fn()
    e.record(); torch.cuda.synchronize()
    ms = s.elapsed_time(e) / reps
    extra = ''
    if flops: extra += '  %.1f TFLOP/s' % (flops / ms / 1e9)
    if nbytes: extra += '  %.1f GB/s' % (nbytes / ms / 1e6)
    print('%-28s %9.3f ms%s' % (name, ms, extra), flush=True)


if what in ('conv2', 'all'):
    H = 416
    x = torch.randn(B, H, H, 128, device=dev)
    w = torch.randn(256, 128, 3, 3, device=dev) * 0.03
    b = torch.zeros(256, device=dev)
    dz = torch.randn(B, H, H, 256, device=dev)
    fl = 2.0 * B * H * H * 256 * 1152
    stats = torch.zeros(ops.STATS_COPIES, 256, 2, dtype=torch.float64, device=dev)
    timeit('conv2 fwd (+stats)', lambda: ops.conv_forward(x, w, b, 3, 1, 1, False, stats, False), fl)
    timeit('conv2 dgrad', lambda: ops.conv_dgrad(dz, w, (B, H, H, 128), 3, 1, 1), fl)
    timeit('conv2 wgrad', lambda: ops.conv_wgrad(x, dz, 3, 1, 1), fl)
    # the same with the BatchNorm backward pass 2 inside (dz here plays dA; z2 = the raw convolution output)
    from capsyolo_amd._lib import call, query
    z2 = torch.randn(B, H, H, 256, device=dev)
    dzo = torch.empty_like(z2)
    sc, sh = torch.rand(256, device=dev) + 0.5, torch.randn(256, device=dev) * 0.1
    mu, isd = torch.randn(256, device=dev) * 0.1, torch.rand(256, device=dev) + 0.5
    red = torch.randn(256, 2, device=dev).double()
    dW = torch.empty(256, 128, 3, 3, device=dev)
    ws = torch.empty(query('cy_wino_wgrad_ws_floats', B, 128, 256), device=dev)
    st = torch.cuda.current_stream().cuda_stream
    timeit('conv2 wgrad + bn pass 2', lambda: call('cy_conv3x3_winograd_wgrad_bn', x.data_ptr(), z2.data_ptr(), dz.data_ptr(), dzo.data_ptr(),
                                                   sc.data_ptr(), sh.data_ptr(), mu.data_ptr(), isd.data_ptr(), 1.0, 1, red.data_ptr(), B * H * H,
                                                   dW.data_ptr(), ws.data_ptr(), B, H, H, 128, 256, st), fl)
    ws4 = torch.empty(query('cy_wino4_wgrad_ws_floats', B, H, H, 128, 256), device=dev)
    timeit('conv2 wgrad F(3x3,4x4) + bn pass 2', lambda: call('cy_conv3x3_winograd4_wgrad_bn', x.data_ptr(), z2.data_ptr(), dz.data_ptr(), dzo.data_ptr(),
                                                              sc.data_ptr(), mu.data_ptr(), isd.data_ptr(), red.data_ptr(), B * H * H,
                                                              dW.data_ptr(), ws4.data_ptr(), B, H, H, 128, 256, st), fl)
    del z2, dzo, ws4
    del x, dz
if what in ('conv3', 'all'):
    H = 416
    x = torch.randn(B, H, H, 256, device=dev)
    w = torch.randn(64, 256, 4, 4, device=dev) * 0.03
    dz = torch.randn(B, H // 2, H // 2, 64, device=dev)
    fl = 2.0 * B * (H // 2) ** 2 * 64 * 4096
    timeit('conv3 fwd', lambda: ops.conv_forward(x, w, None, 4, 2, 1), fl)
    timeit('conv3 dgrad', lambda: ops.conv_dgrad(dz, w, (B, H, H, 256), 4, 2, 1), fl)
    # as in the training step: with conv_2's BatchNorm-backward sums (z read, premasked gradient stored), and the forward with
    # its own BatchNorm statistics
    z3 = torch.randn(B, H, H, 256, device=dev)
    sc3, sh3 = torch.rand(256, device=dev) + 0.5, torch.randn(256, device=dev) * 0.1
    mu3, isd3 = torch.randn(256, device=dev) * 0.1, torch.rand(256, device=dev) + 0.5
    red3 = torch.zeros(ops.STATS_COPIES, 256, 2, dtype=torch.float64, device=dev)
    st3 = torch.zeros(ops.STATS_COPIES, 64, 2, dtype=torch.float64, device=dev)
    timeit('conv3 dgrad + bn sums', lambda: ops.conv_dgrad(dz, w, (B, H, H, 256), 4, 2, 1, 'c', (z3, sc3, sh3, mu3, isd3, 0.1, red3), {}), fl)
    timeit('conv3 fwd + stats', lambda: ops.conv_forward(x, w, None, 4, 2, 1, False, st3, False), fl)
    del z3
    timeit('conv3 wgrad', lambda: ops.conv_wgrad(x, dz, 4, 2, 1), fl)
    del x, dz
if what in ('bn', 'all'):
    from capsyolo_amd._lib import call
    H, N = 416, 256
    P = B * H * H
    z = torch.randn(P, N, device=dev)
    da = torch.randn(P, N, device=dev)
    out = torch.empty_like(z)
    sc, sh = torch.rand(N, device=dev) + 0.5, torch.randn(N, device=dev)
    mu, isd = torch.randn(N, device=dev) * 0.1, torch.rand(N, device=dev) + 0.5
    red = torch.zeros(N, 2, dtype=torch.float64, device=dev)
    st = torch.cuda.current_stream().cuda_stream
    nb = z.numel() * 4.0
    timeit('affine_act (BN apply + LeakyReLU)', lambda: call('cy_affine_act', z.data_ptr(), out.data_ptr(), sc.data_ptr(), sh.data_ptr(), 0.1, P, N, st), None, 2 * nb)
    timeit('bn_bwd_reduce', lambda: call('cy_bn_bwd_reduce', z.data_ptr(), da.data_ptr(), sc.data_ptr(), sh.data_ptr(), mu.data_ptr(), isd.data_ptr(), 0.1, red.data_ptr(), P, N, st), None, 2 * nb)
    timeit('bn_bwd_apply', lambda: call('cy_bn_bwd_apply', z.data_ptr(), da.data_ptr(), out.data_ptr(), sc.data_ptr(), sh.data_ptr(), mu.data_ptr(), isd.data_ptr(), None, 0.1, red.data_ptr(), None, None, P, N, st), None, 3 * nb)
    del z, da, out
if what in ('routing', 'all'):
    g = 13
    feat = torch.randn(B, 4 * g, 4 * g, 256, device=dev, requires_grad=True)
    W = (0.1 * torch.randn(1, 512, 1, 8, 5, device=dev)).requires_grad_(True)
    R = g * g * B
    nb = 4.0 * (R * 4096 + 4096 * 5 + R * 5)
    timeit('routing fwd C=1 (gather)', lambda: ops.routing(feat.detach(), W.detach(), 3, g, B), None, nb)
    v = ops.routing(feat, W, 3, g, B)
    gv = torch.randn_like(v)
    timeit('routing fwd+bwd C=1', lambda: torch.autograd.grad(ops.routing(feat, W, 3, g, B), (feat, W), gv), None, None)
    # general C (CapsuleNet head, B rows) and DarkCapsuleNet3-like head
    u = torch.randn(B, 1296, 8, device=dev, requires_grad=True)
    W2 = (0.1 * torch.randn(1, 1296, 43, 8, 16, device=dev)).requires_grad_(True)
    nb2 = 4.0 * (B * 1296 * 8 + 1296 * 43 * 128 + B * 43 * 16)
    timeit('routing fwd C=43 N=1296 R=%d' % B, lambda: ops.routing(u.detach(), W2.detach(), 3), None, nb2)
    v2 = ops.routing(u, W2, 3)
    gv2 = torch.randn_like(v2)
    timeit('routing fwd+bwd C=43', lambda: torch.autograd.grad(ops.routing(u, W2, 3), (u, W2), gv2))
    # DarkCapsuleNet3 head (C=43, 8->21, cell gather): the many-row, all-iterations-in-one-launch kernels
    W3 = (0.1 * torch.randn(1, 512, 43, 8, 21, device=dev)).requires_grad_(True)
    nb3 = 4.0 * (R * 4096 + 512 * 43 * 168 + R * 43 * 21)
    timeit('routing fwd C=43 N=512 R=%d (DarkCapsuleNet3 head)' % R, lambda: ops.routing(feat.detach(), W3.detach(), 3, g, B), None, nb3)
    v3 = ops.routing(feat, W3, 3, g, B)
    gv3 = torch.randn_like(v3)
    timeit('routing fwd+bwd DarkCapsuleNet3 head', lambda: torch.autograd.grad(ops.routing(feat, W3, 3, g, B), (feat, W3), gv3))
if what == 'bf16':
    # the bf16 path's conv_2 / conv_3 launches at the 608 x 608 shape of BASELINE configs[4] (not part of `all`: 19 GB of operands)
    H = int(os.environ.get('CY_BF16_H', '608'))
    BF = torch.bfloat16
    x = torch.randn(B, H, H, 128, device=dev).to(BF)
    w = torch.randn(256, 128, 3, 3, device=dev) * 0.03
    b = torch.zeros(256, device=dev)
    dz = torch.randn(B, H, H, 256, device=dev).to(BF)
    fl = 2.0 * B * H * H * 256 * 1152
    stats = torch.zeros(ops.STATS_COPIES, 256, 2, dtype=torch.float64, device=dev)
    timeit('bf16 conv2 fwd (+stats)', lambda: ops.conv_forward_bf16(x, w, b, 3, 1, 1, stats), fl)
    timeit('bf16 conv2 dgrad (fp32 out)', lambda: ops.conv_dgrad_bf16(dz, w, (B, H, H, 128), 3, 1, 1, True), fl)
    timeit('bf16 conv2 wgrad', lambda: ops.conv_wgrad_bf16(x, dz, 3, 1, 1), fl)
    # ... and with the BatchNorm-backward apply on the way in (cy_conv_wgrad_bf16_bn: what the training step launches for blocks 2-4)
    from capsyolo_amd._lib import call, query
    nws = query('cy_conv_wgrad_bf16_bn_ws_floats', B, H, H, 128, 256, 3, 1)
    wsb = torch.empty(nws, device=dev); dzo = torch.empty_like(dz); dWb = torch.empty(256, 128, 3, 3, device=dev)
    scb, mub, isb = torch.rand(256, device=dev) + 0.5, torch.randn(256, device=dev) * 0.1, torch.rand(256, device=dev) + 0.5
    redb = torch.randn(256, 2, dtype=torch.float64, device=dev)
    zb = torch.randn(B, H, H, 256, device=dev).to(BF)
    timeit('bf16 conv2 wgrad + bn apply', lambda: call('cy_conv_wgrad_bf16_bn', x.data_ptr(), dz.data_ptr(), zb.data_ptr(), dzo.data_ptr(), dWb.data_ptr(),
                                                        wsb.data_ptr(), scb.data_ptr(), mub.data_ptr(), isb.data_ptr(), redb.data_ptr(), None, None,
                                                        B, H, H, 128, H, H, 256, 3, 1, torch.cuda.current_stream().cuda_stream), fl)
    del x, zb, dzo, wsb
    w3 = torch.randn(64, 256, 4, 4, device=dev) * 0.03
    dz3 = torch.randn(B, H // 2, H // 2, 64, device=dev).to(BF)
    z2 = dz                                       # (any bf16 tensor of conv_2's output shape plays z)
    sc, sh = torch.rand(256, device=dev) + 0.5, torch.randn(256, device=dev) * 0.1
    mu, isd = torch.randn(256, device=dev) * 0.1, torch.rand(256, device=dev) + 0.5
    red = torch.zeros(ops.STATS_COPIES, 256, 2, dtype=torch.float64, device=dev)
    fl3 = 2.0 * B * (H // 2) ** 2 * 64 * 4096
    timeit('bf16 conv3 dgrad, 4 classes in one launch + bn sums',
           lambda: ops.conv_dgrad_bf16(dz3, w3, (B, H, H, 256), 4, 2, 1, False, 'c', (z2, sc, sh, mu, isd, 0.1, red)), fl3)
    timeit('bf16 conv3 dgrad, one launch', lambda: ops.conv_dgrad_bf16(dz3, w3, (B, H, H, 256), 4, 2, 1), fl3)
