"""conv_2 forward (+ statistics) and input gradient on the F(4x4,3x3) kernel (winograd4.hip) against F(2x2,3x3) (winograd.hip):
launch times (HIP events, median) and agreement.   usage: python3 tools/ab_wino4.py [B] [reps] [H]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import capsyolo_amd
from capsyolo_amd import ops

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
H = int(sys.argv[3]) if len(sys.argv) > 3 else 416
dev = torch.device('cuda:0')
torch.manual_seed(0)
x = torch.randn(B, H, H, 128, device=dev)
x = torch.where(x > 0, x, 0.1 * x)
w = torch.randn(256, 128, 3, 3, device=dev) * 0.03
b = torch.zeros(256, device=dev)
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
    ops.USE_WINOGRAD4 = f4
    stats = torch.zeros(ops.STATS_COPIES, 256, 2, dtype=torch.float64, device=dev)
    z = ops.conv_forward(x, w, b, 3, 1, 1, False, stats, False)
    dx = ops.conv_dgrad(dz, w, (B, H, H, 128), 3, 1, 1)
    t_f = med(lambda: ops.conv_forward(x, w, b, 3, 1, 1, False, stats, False))
    t_d = med(lambda: ops.conv_dgrad(dz, w, (B, H, H, 128), 3, 1, 1))
    out[f4] = (z, dx, stats.sum(0) / (reps + 2))
    print('F(%s): fwd+stats %.3f ms (%.1f TFLOP/s direct-equivalent)   dgrad %.3f ms (%.1f)' % ('4x4,3x3' if f4 else '2x2,3x3', t_f, fl / t_f / 1e9, t_d, fl / t_d / 1e9), flush=True)
z0, dx0, s0 = out[False]; z1, dx1, s1 = out[True]
print('fwd rel L2 diff %.3e  max %.3e | dgrad rel L2 %.3e | stats rel %.3e' % (
    float((z1 - z0).norm() / z0.norm()), float((z1 - z0).abs().max() / z0.abs().max()),
    float((dx1 - dx0).norm() / dx0.norm()), float(((s1 - s0).abs() / (s0.abs() + 1e-9)).max())))
