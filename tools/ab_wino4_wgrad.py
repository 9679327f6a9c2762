"""conv_2 weight gradient on F(3x3,4x4) (winograd4_wgrad.hip) against F(3x3,2x2) (winograd.hip): launch times (HIP events, median)
and agreement, plain and with the fused BatchNorm backward.   usage: python3 tools/ab_wino4_wgrad.py [B] [reps] [H]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import capsyolo_amd
from capsyolo_amd import ops
from capsyolo_amd._lib import call, query

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
H = int(sys.argv[3]) if len(sys.argv) > 3 else 416
dev = torch.device('cuda:0')
torch.manual_seed(0)
x = torch.randn(B, H, H, 128, device=dev)
x = torch.where(x > 0, x, 0.1 * x)
dz = torch.randn(B, H, H, 256, device=dev)
fl = 2.0 * B * H * H * 256 * 1152


def med(fn):
    fn(); torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record(); fn(); e.record(); torch.cuda.synchronize()
        ts.append(s.elapsed_time(e))
    ts.sort()
    return ts[len(ts) // 2]


out = {}
for f4 in (False, True):
    ops.USE_WINOGRAD4_WGRAD = f4
    out[f4] = ops.conv_wgrad(x, dz, 3, 1, 1)
    t = med(lambda: ops.conv_wgrad(x, dz, 3, 1, 1))
    print('F(3x3,%s) weight gradient %.3f ms (%.1f TFLOP/s direct-equivalent)' % ('4x4' if f4 else '2x2', t, fl / t / 1e9), flush=True)
d0, d1 = out[False], out[True]
print('rel L2 diff %.3e  max/max %.3e' % (float((d1 - d0).norm() / d0.norm()), float((d1 - d0).abs().max() / d0.abs().max())))
# fused BatchNorm backward (premasked)
z = torch.randn(B, H, H, 256, device=dev)
dzo = torch.empty_like(z)
sc, mu, isd = torch.rand(256, device=dev) + 0.5, torch.randn(256, device=dev) * 0.1, torch.rand(256, device=dev) + 0.5
sh = torch.randn(256, device=dev) * 0.1
red = torch.randn(256, 2, device=dev).double()
dW = torch.empty(256, 128, 3, 3, device=dev)
st = torch.cuda.current_stream().cuda_stream
ws2 = torch.empty(query('cy_wino_wgrad_ws_floats', B, 128, 256), device=dev)
ws4 = torch.empty(query('cy_wino4_wgrad_ws_floats', B, H, H, 128, 256), device=dev)
t2 = med(lambda: call('cy_conv3x3_winograd_wgrad_bn', x.data_ptr(), z.data_ptr(), dz.data_ptr(), dzo.data_ptr(), sc.data_ptr(), sh.data_ptr(),
                      mu.data_ptr(), isd.data_ptr(), 1.0, 1, red.data_ptr(), B * H * H, dW.data_ptr(), ws2.data_ptr(), B, H, H, 128, 256, st))
a2, w2 = dzo.clone(), dW.clone()
t4 = med(lambda: call('cy_conv3x3_winograd4_wgrad_bn', x.data_ptr(), z.data_ptr(), dz.data_ptr(), dzo.data_ptr(), sc.data_ptr(), mu.data_ptr(),
                      isd.data_ptr(), red.data_ptr(), B * H * H, dW.data_ptr(), ws4.data_ptr(), B, H, H, 128, 256, st))
print('with the fused BatchNorm backward: F(3x3,2x2) %.3f ms, F(3x3,4x4) %.3f ms; dz max diff %.2e, dW rel L2 %.2e'
      % (t2, t4, float((dzo - a2).abs().max()), float((dW - w2).norm() / w2.norm())))
