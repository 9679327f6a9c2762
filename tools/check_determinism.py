"""Bit-identical repetition of the convolution kernels on the same inputs (fixed-order partial sums, no float atomics in what
they return): the bf16 GEMM / weight-gradient kernels, the fp32 Winograd kernels and the weight gradient with the fused
BatchNorm backward.  A kernel whose VALU work lands in a hardware hazard shows up here as run-to-run differences in a few
lanes long before a tolerance test notices.   usage: python3 tools/check_determinism.py [B]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from capsyolo_amd import ops
from capsyolo_amd._lib import call, query

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
BF = torch.bfloat16
dev = torch.device('cuda:0')
torch.manual_seed(0)
bad = 0


def rep(name, fn, n=4):
    global bad
    outs = [fn() for _ in range(n)]
    torch.cuda.synchronize()
    nd = [int((outs[0] != o).sum()) for o in outs[1:]]
    bad += sum(nd)
    print('%-44s differing elements over %d repetitions: %s' % (name, n - 1, nd), flush=True)


for (Cin, H, Cout, k, s) in ((128, 208, 256, 3, 1), (256, 208, 64, 4, 2), (64, 104, 128, 4, 2), (128, 52, 256, 4, 2)):
    Ho = (H + 2 - k) // s + 1
    x = torch.randn(B, H, H, Cin, device=dev)
    w = torch.randn(Cout, Cin, k, k, device=dev) * 0.03
    b = torch.zeros(Cout, device=dev)
    dz = torch.randn(B, Ho, Ho, Cout, device=dev)
    xb, dzb = x.to(BF), dz.to(BF)
    tag = '%d->%d k%d s%d %dx%d ' % (Cin, Cout, k, s, H, H)
    rep(tag + 'bf16 fwd', lambda: ops.conv_forward_bf16(xb, w, b, k, s, 1))
    rep(tag + 'bf16 dgrad', lambda: ops.conv_dgrad_bf16(dzb, w, (B, H, H, Cin), k, s, 1, False))
    rep(tag + 'bf16 wgrad', lambda: ops.conv_wgrad_bf16(xb, dzb, k, s, 1))
    rep(tag + 'fp32 fwd', lambda: ops.conv_forward(x, w, b, k, s, 1))
    rep(tag + 'fp32 dgrad', lambda: ops.conv_dgrad(dz, w, (B, H, H, Cin), k, s, 1))
    rep(tag + 'fp32 wgrad', lambda: ops.conv_wgrad(x, dz, k, s, 1))
    if k == 3:
        z = torch.randn(B, Ho, Ho, Cout, device=dev) * 1.5 + 0.7
        sc, sh = torch.rand(Cout, device=dev) + 0.5, torch.randn(Cout, device=dev) * 0.3
        mu, isd = torch.randn(Cout, device=dev) * 0.2 + 0.7, torch.rand(Cout, device=dev) * 0.3 + 0.5
        red = torch.randn(Cout, 2, device=dev).double()
        st = torch.cuda.current_stream().cuda_stream
        ws = torch.empty(query('cy_wino_wgrad_ws_floats', B, Cin, Cout), device=dev)

        def fused():
            o, dw = torch.empty_like(z), torch.empty(Cout, Cin, 3, 3, device=dev)
            call('cy_conv3x3_winograd_wgrad_bn', x.data_ptr(), z.data_ptr(), dz.data_ptr(), o.data_ptr(), sc.data_ptr(), sh.data_ptr(),
                 mu.data_ptr(), isd.data_ptr(), 0.1, 0, red.data_ptr(), B * Ho * Ho, dw.data_ptr(), ws.data_ptr(), B, H, H, Cin, Cout, st)
            return torch.cat([o.flatten(), dw.flatten()])
        rep(tag + 'fp32 wgrad + BatchNorm pass 2 (dz, dW)', fused)
        if query('cy_wino4_wgrad_ok', B, H, H, Cin, Cout):
            ws4 = torch.empty(query('cy_wino4_wgrad_ws_floats', B, H, H, Cin, Cout), device=dev)

            def fused4():
                o, dw = torch.empty_like(z), torch.empty(Cout, Cin, 3, 3, device=dev)
                call('cy_conv3x3_winograd4_wgrad_bn', x.data_ptr(), z.data_ptr(), dz.data_ptr(), o.data_ptr(), sc.data_ptr(), mu.data_ptr(),
                     isd.data_ptr(), red.data_ptr(), B * Ho * Ho, dw.data_ptr(), ws4.data_ptr(), B, H, H, Cin, Cout, st)
                return torch.cat([o.flatten(), dw.flatten()])
            rep(tag + 'fp32 F(3x3,4x4) wgrad + BatchNorm pass 2 (dz, dW)', fused4)
            old = ops.WINOGRAD4_MIN_PIXELS
            ops.WINOGRAD4_MIN_PIXELS = 0
            rep(tag + 'fp32 F(3x3,4x4) wgrad', lambda: ops.conv_wgrad(x, dz, k, s, 1))
            rep(tag + 'fp32 F(4x4,3x3) fwd', lambda: ops.conv_forward(x, w, b, k, s, 1))
            rep(tag + 'fp32 F(4x4,3x3) dgrad', lambda: ops.conv_dgrad(dz, w, (B, H, H, Cin), k, s, 1))
            ops.WINOGRAD4_MIN_PIXELS = old
    if k == 4 and query('cy_wino4s2_ok', B, H, H, Cin, Cout) and query('cy_wino4s2_dgrad_ok', B, H, H, Cin, Cout):
        old = ops.WINOGRAD4_S2_MIN_PIXELS
        ops.WINOGRAD4_S2_MIN_PIXELS = 0
        zz = torch.randn(B, H, H, Cin, device=dev)
        sc, sh = torch.rand(Cin, device=dev) + 0.5, torch.randn(Cin, device=dev) * 0.3
        mu, isd = torch.randn(Cin, device=dev) * 0.2, torch.rand(Cin, device=dev) * 0.3 + 0.5
        red = torch.zeros(ops.STATS_COPIES, Cin, 2, dtype=torch.float64, device=dev)
        rep(tag + 'fp32 F(4x4,2x2) fwd', lambda: ops.conv_forward(x, w, b, k, s, 1))
        rep(tag + 'fp32 F(4x4,2x2) fwd, input affine', lambda: ops.conv_forward(x, w, None, k, s, 1, False, None, False, 'c', (sc, sh, 0.1)))
        rep(tag + 'fp32 F(4x4,2x2) dgrad', lambda: ops.conv_dgrad(dz, w, (B, H, H, Cin), k, s, 1))
        # (the sums themselves are double atomics: the stored premasked gradient is what must repeat)
        rep(tag + 'fp32 F(4x4,2x2) dgrad + BatchNorm sums (dx)', lambda: ops.conv_dgrad(dz, w, (B, H, H, Cin), k, s, 1, 'c', (zz, sc, sh, mu, isd, 0.1, red), {}))
        ops.WINOGRAD4_S2_MIN_PIXELS = old
print('check_determinism:', 'ok' if bad == 0 else 'FAILED')
sys.exit(0 if bad == 0 else 1)
