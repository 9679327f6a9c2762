"""temporary: per-phase cycle counts of the stride-2 Winograd forward kernel (library built with -DW2_PROF)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from capsyolo_amd import ops
dev = torch.device('cuda:0')
B, H = 32, 416
x = torch.randn(B, H, H, 256, device=dev)
w = torch.randn(64, 256, 4, 4, device=dev) * 0.03
prof = torch.zeros(256, 4, dtype=torch.int64, device=dev)
os.environ['CY_W2_PROF'] = hex(prof.data_ptr())
aff = None
if len(sys.argv) > 1 and sys.argv[1] == 'affine':
    aff = (torch.rand(256, device=dev) + 0.5, torch.randn(256, device=dev), 0.1)
f = lambda: ops.conv_forward(x, w, None, 4, 2, 1, in_affine=aff) if aff else ops.conv_forward(x, w, None, 4, 2, 1)
for _ in range(2): f()
torch.cuda.synchronize()
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
s.record(); f(); e.record(); torch.cuda.synchronize()
p = prof.cpu().double()
print('ms', s.elapsed_time(e), 'tiles/block', (p[:, 3] % 4294967296).mean().item(), 'min', (p[:, 3] % 4294967296).min().item(), 'max', (p[:, 3] % 4294967296).max().item())
print('loop per chunk: mean %.0f cycles (128 chunks per tile); epilogue per tile: %.0f' % ((p[:, 0] / (p[:, 3] % 4294967296)).mean() / 128, (p[:, 1] / (p[:, 3] % 4294967296)).mean()))
print('total per block (cycles): mean %.0f max %.0f' % (p[:, :3].sum(1).mean(), p[:, :3].sum(1).max()))
pi = prof.cpu()
nt = (pi[:, 3] & 0xffffffff).double(); nf = (pi[:, 3] >> 32).double()
lf = pi[:, 2].double(); la = pi[:, 0].double()
print('tiles %d inside %d | loop cycles per chunk: inside tiles %.0f, border tiles %.0f' % (nt.sum(), nf.sum(), lf.sum() / nf.sum() / 128, (la.sum() - lf.sum()) / (nt.sum() - nf.sum()) / 128))
if os.environ.get('STAMPS'):
    st = torch.zeros(256, 16, 8, dtype=torch.int64, device=dev)
    os.environ['CY_W2_STAMPS'] = hex(st.data_ptr())
    f(); torch.cuda.synchronize()
    d = st.cpu().double()[:, :11, :]          # tiles 0..10 of every block
    names = ['slots 0-23', 'slots 24-47', 'slots 48-71', 'barrier', 'cursor bookkeeping']
    for i, n in enumerate(names):
        x = d[:, :, i + 1] - d[:, :, i]
        print('%-20s mean %.0f  median %.0f  max %.0f' % (n, x.mean(), x.median(), x.max()))
