"""s_memtime stamps of the F(4x4,3x3) chunk loop (winograd4.hip built with -DCY_F4_PROF into libdbg_f4prof.so, loaded through
CAPSYOLO_LIB): cycles per third of a chunk, barrier wait, tail, drain -- median over the blocks.
usage: CAPSYOLO_LIB=.../libdbg_f4prof.so python3 tools/f4prof.py [fwd|dgrad]"""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import capsyolo_amd
from capsyolo_amd import ops, _lib

what = sys.argv[1] if len(sys.argv) > 1 else 'fwd'
dev = torch.device('cuda:0')
torch.manual_seed(0)
B, H = 32, 416
if what == 'fwd':
    x = torch.randn(B, H, H, 128, device=dev); w = torch.randn(256, 128, 3, 3, device=dev) * 0.03
    stats = torch.zeros(ops.STATS_COPIES, 256, 2, dtype=torch.float64, device=dev)
    fn = lambda: ops.conv_forward(x, w, torch.zeros(256, device=dev), 3, 1, 1, False, stats, False)
else:
    dz = torch.randn(B, H, H, 256, device=dev); w = torch.randn(256, 128, 3, 3, device=dev) * 0.03
    fn = lambda: ops.conv_dgrad(dz, w, (B, H, H, 128), 3, 1, 1)
for _ in range(3):
    fn()
torch.cuda.synchronize()
lib = _lib.load()
buf = (ctypes.c_ulonglong * (256 * 16))()
rc = lib.cy_wino4_read_prof(buf)
a = np.array(buf, dtype=np.uint64).reshape(256, 16).astype(np.int64)
d = lambda i, j: float(np.median(a[:, j] - a[:, i]))
print('%s: chunk %d cycles = slots 0-47 %d | 48-95 %d | 96-135 %d | barrier %d | 137-143 %d | tail (SALU) %d ;  drain %d + barrier %d'
      % (what, d(0, 6), d(0, 1), d(1, 2), d(2, 3), d(3, 4), d(4, 5), d(5, 6), d(8, 9), d(9, 10)))
