"""s_memtime stamps of the F(3x3,4x4) weight-gradient chunk loop (winograd4_wgrad.hip built with -DCY_G4_PROF into libdbg_g4prof.so,
loaded through CAPSYOLO_LIB): cycles per segment of a chunk, for the input-transform waves and the dz-transform waves.
usage: CAPSYOLO_LIB=.../libdbg_g4prof.so python3 tools/g4prof.py [bn]"""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import capsyolo_amd
from capsyolo_amd import ops, _lib
from capsyolo_amd._lib import call, query

bn = len(sys.argv) > 1 and sys.argv[1] == 'bn'
dev = torch.device('cuda:0')
torch.manual_seed(0)
B, H = 32, 416
x = torch.randn(B, H, H, 128, device=dev); dz = torch.randn(B, H, H, 256, device=dev)
if bn:
    z = torch.randn(B, H, H, 256, device=dev); dzo = torch.empty_like(z)
    sc, mu, isd = torch.rand(256, device=dev) + 0.5, torch.randn(256, device=dev) * 0.1, torch.rand(256, device=dev) + 0.5
    red = torch.randn(256, 2, device=dev).double(); dW = torch.empty(256, 128, 3, 3, device=dev)
    ws4 = torch.empty(query('cy_wino4_wgrad_ws_floats', B, H, H, 128, 256), device=dev)
    st = torch.cuda.current_stream().cuda_stream
    fn = lambda: call('cy_conv3x3_winograd4_wgrad_bn', x.data_ptr(), z.data_ptr(), dz.data_ptr(), dzo.data_ptr(), sc.data_ptr(), mu.data_ptr(),
                      isd.data_ptr(), red.data_ptr(), B * H * H, dW.data_ptr(), ws4.data_ptr(), B, H, H, 128, 256, st)
else:
    fn = lambda: ops.conv_wgrad(x, dz, 3, 1, 1)
for _ in range(3):
    fn()
torch.cuda.synchronize()
lib = _lib.load()
buf = (ctypes.c_ulonglong * (256 * 2 * 16))()
lib.cy_wino4_wgrad_read_prof(buf)
a = np.array(buf, dtype=np.uint64).reshape(256, 2, 16).astype(np.int64)
for role, name in ((0, 'input-transform waves'), (1, 'dz-transform waves')):
    d = lambda i, j: float(np.median(a[:, role, j] - a[:, role, i]))
    print('%s %s: chunk %d cycles = slots 0-15 %d | 16-30 %d | mid barrier %d | 32-47 %d | 48-63 %d | end barrier %d | 65-71 %d'
          % ('bn' if bn else 'plain', name, d(0, 7), d(0, 1), d(1, 2), d(2, 3), d(3, 4), d(4, 5), d(5, 6), d(6, 7)))
