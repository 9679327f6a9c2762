"""temporary: per-phase cycle counts of the stride-2 Winograd input-gradient kernel (library built with -DW2_PROF)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from capsyolo_amd import ops
dev = torch.device('cuda:0')
B, H = 32, 416
fuse = len(sys.argv) > 1 and sys.argv[1] == 'bn'
w = torch.randn(64, 256, 4, 4, device=dev) * 0.03
dz = torch.randn(B, H // 2, H // 2, 64, device=dev)
prof = torch.zeros(256, 4, dtype=torch.int64, device=dev)
os.environ['CY_W2_PROF'] = hex(prof.data_ptr())
bn = None
if fuse:
    z = torch.randn(B, H, H, 256, device=dev)
    sc, sh, mu, istd = [torch.rand(256, device=dev) + 0.5 for _ in range(4)]
    red = torch.zeros(ops.STATS_COPIES, 256, 2, dtype=torch.float64, device=dev)
    bn = (z, sc, sh, mu, istd, 0.1, red)
for _ in range(2):
    ops.conv_dgrad(dz, w, (B, H, H, 256), 4, 2, 1, bn_fuse=bn) if fuse else ops.conv_dgrad(dz, w, (B, H, H, 256), 4, 2, 1)
torch.cuda.synchronize()
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
s.record()
ops.conv_dgrad(dz, w, (B, H, H, 256), 4, 2, 1, bn_fuse=bn) if fuse else ops.conv_dgrad(dz, w, (B, H, H, 256), 4, 2, 1)
e.record(); torch.cuda.synchronize()
p = prof.cpu().double()
p[:, 3] = p[:, 3] % 4294967296
print("ms", s.elapsed_time(e), "tiles/block", p[:, 3].mean().item())
for i, n in enumerate(('loop', 'ep1 (acc->stores)', 'ep2 (sums+barrier)')):
    print('%-20s per tile: mean %.0f  min %.0f  max %.0f cycles' % (n, (p[:, i] / p[:, 3]).mean(), (p[:, i] / p[:, 3]).min(), (p[:, i] / p[:, 3]).max()))
print('total per block (cycles): mean %.0f' % p[:, :3].sum(1).mean())
