"""conv_3 / conv_4 / conv_5 forward (4x4 / stride 2) on F(4x4,2x2) (winograd4_s2.hip) against F(2x2,2x2) (winograd_s2.hip): launch times
(HIP events, median) with BatchNorm statistics, plain and with the fused input affine, and agreement.   usage: python3 tools/ab_wino42.py [B]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import capsyolo_amd
from capsyolo_amd import ops

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
reps = 5
dev = torch.device('cuda:0')
torch.manual_seed(0)


def med(fn):
    fn(); torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record(); fn(); e.record(); torch.cuda.synchronize()
        ts.append(s.elapsed_time(e))
    ts.sort()
    return ts[len(ts) // 2]


for name, H, Cin, Cout in (('conv_3', 416, 256, 64), ('conv_4', 208, 64, 128), ('conv_5', 104, 128, 256)):
    x = torch.randn(B, H, H, Cin, device=dev)
    w = torch.randn(Cout, Cin, 4, 4, device=dev) * 0.03
    b = torch.zeros(Cout, device=dev)
    sc, sh = torch.rand(Cin, device=dev) + 0.5, torch.randn(Cin, device=dev) * 0.1
    fl = 2.0 * B * (H // 2) ** 2 * Cout * 16 * Cin
    res = {}
    for f42 in (False, True):
        ops.USE_WINOGRAD4_S2 = f42
        stats = torch.zeros(ops.STATS_COPIES, Cout, 2, dtype=torch.float64, device=dev)
        z = ops.conv_forward(x, w, b, 4, 2, 1, False, stats, False)
        za = ops.conv_forward(x, w, None, 4, 2, 1, False, stats, False, 'c', (sc, sh, 0.1))
        t = med(lambda: ops.conv_forward(x, w, b, 4, 2, 1, False, stats, False))
        ta = med(lambda: ops.conv_forward(x, w, None, 4, 2, 1, False, stats, False, 'c', (sc, sh, 0.1)))
        res[f42] = (z, za, t, ta)
    z0, za0, t0, ta0 = res[False]; z1, za1, t1, ta1 = res[True]
    print('%s forward + stats: F(2x2,2x2) %.3f ms, F(4x4,2x2) %.3f ms (%.0f TFLOP/s direct-equivalent) | with the input affine %.3f -> %.3f ms | rel L2 diff %.2e / %.2e'
          % (name, t0, t1, fl / t1 / 1e9, ta0, ta1, float((z1 - z0).norm() / z0.norm()), float((za1 - za0).norm() / za0.norm())), flush=True)
    # input gradient, plain and with the producer block's fused BatchNorm-backward sums
    dz = torch.randn(B, H // 2, H // 2, Cout, device=dev)
    zin = torch.randn(B, H, H, Cin, device=dev)
    mu, isd = torch.randn(Cin, device=dev) * 0.1, torch.rand(Cin, device=dev) + 0.5
    res = {}
    for f42 in (False, True):
        ops.USE_WINOGRAD4_S2_DGRAD = f42
        red = torch.zeros(ops.STATS_COPIES, Cin, 2, dtype=torch.float64, device=dev)
        dx = ops.conv_dgrad(dz, w, (B, H, H, Cin), 4, 2, 1)
        dxb = ops.conv_dgrad(dz, w, (B, H, H, Cin), 4, 2, 1, 'c', (zin, sc, sh, mu, isd, 0.1, red), {})
        redc = red.sum(0).clone()
        t = med(lambda: ops.conv_dgrad(dz, w, (B, H, H, Cin), 4, 2, 1))
        tb = med(lambda: ops.conv_dgrad(dz, w, (B, H, H, Cin), 4, 2, 1, 'c', (zin, sc, sh, mu, isd, 0.1, red), {}))
        res[f42] = (dx, dxb, redc, t, tb)
    d0, db0, r0, t0, tb0 = res[False]; d1, db1, r1, t1, tb1 = res[True]
    print('%s input gradient: F(2x2,2x2) %.3f ms, F(4x4,2x2) %.3f ms (%.0f TFLOP/s direct-equivalent) | with the BatchNorm sums %.3f -> %.3f ms | rel L2 diff %.2e / %.2e, sums %.2e'
          % (name, t0, t1, fl / t1 / 1e9, tb0, tb1, float((d1 - d0).norm() / d0.norm()), float((db1 - db0).norm() / db0.norm()),
             float((r1 - r0).norm() / r0.norm())), flush=True)
    del x, zin, dz, res
